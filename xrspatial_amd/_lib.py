"""ctypes binding of libxrs_hip.so (C ABI declared in include/xrs_hip.h).

There is deliberately NO CPU fallback: if the HIP library cannot be loaded, or no
MI355X is visible, every compute entry point raises.  (The CPU oracle under
`oracle/` is test infrastructure and is never imported from this package.)
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XRS_LIB") or os.path.join(_HERE, "libxrs_hip.so")   # XRS_LIB: A/B builds of the same ABI

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int64 = ctypes.c_int64
c_double = ctypes.c_double
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t
c_uint = ctypes.c_uint


class XrsError(RuntimeError):
    """A libxrs_hip.so entry point returned non-zero."""


# name -> argtypes (every function returns int unless listed in _RESTYPES)
_PROTOTYPES = {
    "xrs_version": [],
    "xrs_last_error": [ctypes.c_char_p, c_size_t],
    "xrs_build_id": [ctypes.c_char_p, c_size_t],
    "xrs_device_count": [ctypes.POINTER(c_int)],
    "xrs_set_device": [c_int],
    "xrs_get_device": [ctypes.POINTER(c_int)],
    "xrs_device_name": [c_int, ctypes.c_char_p, c_size_t],
    "xrs_mem_info": [ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)],
    "xrs_malloc": [ctypes.POINTER(c_void_p), c_size_t],
    "xrs_free": [c_void_p],
    "xrs_memcpy_h2d": [c_void_p, c_void_p, c_size_t, c_void_p],
    "xrs_memcpy_d2h": [c_void_p, c_void_p, c_size_t, c_void_p],
    "xrs_memcpy_d2d": [c_void_p, c_void_p, c_size_t, c_void_p],
    "xrs_memset": [c_void_p, c_int, c_size_t, c_void_p],
    "xrs_copy_f32": [c_void_p, c_void_p, c_int64, c_void_p],
    "xrs_stream_mix_f32": [c_void_p, c_void_p, c_int, c_int64, c_void_p],
    "xrs_copy2d": [c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_int64, c_void_p],
    "xrs_match_bbox": [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "xrs_nan_minmax_f32": [c_void_p, c_int64, c_void_p, c_void_p],
    "xrs_true_color_u8": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_double, c_double, c_double,
                          c_void_p, c_void_p],
    "xrs_stream_create": [ctypes.POINTER(c_void_p)],
    "xrs_stream_destroy": [c_void_p],
    "xrs_stream_sync": [c_void_p],
    "xrs_device_sync": [],
    "xrs_event_create": [ctypes.POINTER(c_void_p)],
    "xrs_event_destroy": [c_void_p],
    "xrs_event_record": [c_void_p, c_void_p],
    "xrs_stream_wait_event": [c_void_p, c_void_p],
    "xrs_cast_f32": [c_void_p, c_int, c_void_p, c_int64, c_void_p],
    "xrs_host_alloc": [ctypes.POINTER(c_void_p), c_size_t],
    "xrs_host_free": [c_void_p],
    "xrs_event_sync": [c_void_p],
    "xrs_event_elapsed_ms": [c_void_p, c_void_p, ctypes.POINTER(c_float)],
    "xrs_slope_f32": [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_double, c_double,
                      c_int, c_int, c_void_p],
    "xrs_aspect_f32": [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_void_p],
    "xrs_curvature_f32": [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_double,
                          c_int, c_int, c_void_p],
    "xrs_hillshade_f32": [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_double, c_double,
                          c_int, c_int, c_void_p],
    "xrs_terrain_fused_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                              c_int64, c_double, c_double, c_double, c_double, c_int, c_int, c_void_p],
    "xrs_raster_pass_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                            c_void_p, c_int64, c_int64, c_int64, c_int64, c_double, c_double, c_double, c_double,
                            c_int, c_int, c_void_p],
    "xrs_raster_pass_edges_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_void_p, c_int64, c_int64, c_int64, c_int64, c_double, c_double, c_double, c_double,
                                  c_int, c_int, c_int64, c_void_p],
    "xrs_geodesic_workspace_bytes": [c_int64, c_int64],
    "xrs_geodesic_f32": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int64, c_int64,
                         c_int64, c_double, c_double, c_double, c_int, c_void_p, c_int, c_int, c_void_p],
    "xrs_normalized_ratio_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "xrs_evi_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double, c_double, c_double,
                    c_void_p],
    "xrs_savi_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_double, c_void_p],
    "xrs_arvi_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "xrs_gci_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "xrs_sipi_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "xrs_ebbi_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "xrs_kxk_workspace_bytes": [c_int, c_int],
    "xrs_convolve2d_f32": [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int, c_int,
                           c_void_p, c_int, c_int, c_void_p],
    "xrs_focal_stats_f32": [c_void_p, ctypes.POINTER(c_void_p), c_uint, c_int64, c_int64, c_int64, c_int64,
                            c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p],
    "xrs_focal_stats_f32_ex": [c_void_p, ctypes.POINTER(c_void_p), c_uint, c_int64, c_int64, c_int64, c_int64,
                               c_void_p, c_int, c_int, c_void_p, c_size_t, c_int, c_int, c_uint, c_void_p],
    "xrs_focal_workspace_bytes": [c_int64, c_int64, c_int, c_int],
    "xrs_focal_mean3x3_passes": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_void_p],
    "xrs_focal_mean3x3": [c_void_p, c_int, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int,
                          c_int, c_int, c_void_p],
    "xrs_nan_moments_f32": [c_void_p, c_int64, c_void_p, c_void_p],
    "xrs_hotspots_classify_f32": [c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p],
    "xrs_zonal_init": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "xrs_zonal_partials_f32": [c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_double, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p],
    "xrs_zonal_partials_lut_f32": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_double, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "xrs_zonal_partials_lut_f64": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_double, c_int, c_double, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "xrs_zonal_init_f64": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "xrs_zonal_partials_f64": [c_void_p, c_void_p, c_int64, c_int, c_double, c_int, c_double, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p],
    "xrs_zonal_scan": [c_void_p, c_int, c_int64, c_void_p, c_void_p],
    "xrs_zonal_partials_window_f32": [c_void_p, c_int, c_int, c_void_p, c_int64, c_float, c_int, c_double, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "xrs_zonal_partials_window_f64": [c_void_p, c_int, c_int, c_void_p, c_int64, c_double, c_int, c_double, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "xrs_zonal_sample_f32": [c_void_p, c_void_p, c_int64, c_int64, c_float, c_int, c_void_p, c_void_p],
    "xrs_zonal_sample_f64": [c_void_p, c_void_p, c_int64, c_int64, c_double, c_int, c_void_p, c_void_p],
    "xrs_zonal_scan_presence_i32": [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p],
    "xrs_zonal_presence": [c_void_p, c_int, c_int64, c_double, c_int64, c_void_p, c_void_p],
    "xrs_zonal_index": [c_void_p, c_int, c_int64, c_double, c_int64, c_void_p, c_void_p, c_void_p],
    "xrs_crosstab_counts": [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p],
    "xrs_zonal_majority_workspace_bytes": [c_int64, c_int, c_int],
    "xrs_zonal_majority_f32": [c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p, c_size_t, c_void_p,
                               c_void_p],
    "xrs_zonal_majority_f64": [c_void_p, c_void_p, c_int64, c_int, c_double, c_int, c_void_p, c_size_t, c_void_p,
                               c_void_p],
    "xrs_zonal_mode_workspace_bytes": [c_int64, c_int, c_int],
    "xrs_zonal_mode_max_zones": [],
    "xrs_zonal_mode_f32": [c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p],
    "xrs_zonal_mode_f64": [c_void_p, c_void_p, c_int64, c_int, c_double, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p],
    "xrs_focal_windows_f32": [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_void_p],
    "xrs_zonal_group_f32": [c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p, c_size_t, c_void_p, c_void_p],
    "xrs_zonal_group_f64": [c_void_p, c_void_p, c_int64, c_int, c_double, c_int, c_void_p, c_size_t, c_void_p, c_void_p],
    "xrs_zonal_backproject_f64": [c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "xrs_comm_unique_id": [c_void_p],
    "xrs_comm_init_rank": [ctypes.POINTER(c_void_p), c_void_p, c_int, c_int],
    "xrs_comm_destroy": [c_void_p],
    "xrs_comm_info": [c_void_p, c_void_p],
    "xrs_comm_selftest_f32": [c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "xrs_halo_exchange_f32": [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p],
    "xrs_zonal_allreduce": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "xrs_allreduce_f64": [c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "xrs_allreduce_u8": [c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "xrs_allreduce_u64": [c_void_p, c_void_p, c_int64, c_int, c_void_p],
}
_RESTYPES = {"xrs_kxk_workspace_bytes": c_size_t, "xrs_focal_workspace_bytes": c_size_t, "xrs_zonal_majority_workspace_bytes": c_size_t,
             "xrs_zonal_mode_workspace_bytes": c_size_t,
             "xrs_geodesic_workspace_bytes": c_size_t}

EXPORTED = tuple(_PROTOTYPES)

_lib = None
_lock = threading.Lock()


def load():
    """Load libxrs_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise XrsError(
                    f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                    "(or `make -C xrspatial_amd/csrc`).  There is no CPU fallback.")
            lib = ctypes.CDLL(LIB_PATH)
            for name, argtypes in _PROTOTYPES.items():
                fn = getattr(lib, name)
                fn.argtypes = argtypes
                fn.restype = _RESTYPES.get(name, c_int)
            _lib = lib
    return _lib


def last_error() -> str:
    buf = ctypes.create_string_buffer(512)
    load().xrs_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def build_id() -> str:
    """Identity of the sources the loaded library was built from (xrs_build_id)."""
    buf = ctypes.create_string_buffer(64)
    load().xrs_build_id(buf, 64)
    return buf.value.decode(errors="replace")


def check(rc: int, what: str = ""):
    if rc != 0:
        raise XrsError(f"{what}: {last_error()}" if what else last_error())


_device_ready = False
_device_index = 0
_tls = threading.local()


def call(name: str, *args):
    """Call an int-returning entry point and raise XrsError on failure."""
    lib = load()
    if _device_ready and not getattr(_tls, "bound", False):
        # the HIP "current device" is per host thread: bind every new thread to this process's GPU once
        _tls.bound = True
        check(lib.xrs_set_device(_device_index), "xrs_set_device")
    check(getattr(lib, name)(*args), name)


def require_device():
    """Fail loudly unless a HIP device is usable (selects LOCAL_RANK's GPU once)."""
    global _device_ready, _device_index
    if _device_ready:
        return
    lib = load()
    n = c_int(0)
    rc = lib.xrs_device_count(ctypes.byref(n))
    if rc != 0 or n.value < 1:
        raise XrsError("no MI355X / HIP device visible (%s); xrspatial_amd has no CPU fallback"
                       % (last_error() or "device count is 0"))
    dev = int(os.environ.get("XRS_DEVICE", os.environ.get("LOCAL_RANK", "0"))) % n.value
    check(lib.xrs_set_device(dev), "xrs_set_device")
    _device_index = dev
    _tls.bound = True
    _device_ready = True


def device_available() -> bool:
    try:
        require_device()
        return True
    except (XrsError, OSError):
        return False
