"""Device-resident arrays: the fifth backend slot next to numpy / cupy / dask / dask+cupy.

A `DeviceArray` owns (or views) a C-order buffer in MI355X HBM obtained from
xrs_malloc.  A DataArray whose `.data` is a DeviceArray stays on the device across
calls (the reference's analogue is a cupy-backed DataArray, utils.py:124-143);
a numpy-backed DataArray is copied in and out around each call.

Freed buffers go to a small size-keyed free list so steady-state pipelines
(bench.py, repeated calls) do not pay hipMalloc/hipFree per call.

Host side of the numpy-in / numpy-out path (measured on the MI355X box, tools/hostcopy_probe.py): copies run
at PCIe rate (~56 GB/s) in both directions EXCEPT into a freshly allocated result array, where the first touch
of every page (kernel zero-fill, ~12 GB/s) dominates the whole call.  Results therefore land in recycled host
blocks (`host_empty`): the block goes back to a free list when the last NumPy view of it is garbage-collected
and the next result of that size reuses the already-faulted pages.  Inputs that are not float32 are sent in
their own dtype and converted in HBM (`xrs_cast_f32`) instead of `.astype(np.float32)` on one CPU core.
"""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np

from . import _lib

_pool = {}                               # nbytes -> [(device pointer, event recorded when the block was freed)]
_pool_lock = threading.RLock()          # re-entrant: DeviceArray.__del__ can run (GC) while a thread holds it
_POOL_MAX_BYTES = 64 << 30
_pool_bytes = 0
_spare_events = []


def _active_stream():
    from ._launch import get_stream       # (imported late: _launch imports this module)
    return get_stream()


def _raw_alloc(nbytes: int) -> int:
    global _pool_bytes
    nbytes = max(int(nbytes), 16)
    entry = None
    with _pool_lock:
        lst = _pool.get(nbytes)
        if lst:
            _pool_bytes -= nbytes
            entry = lst.pop()
    if entry is not None:
        ptr, ev = entry
        if ev is not None:
            # the block was freed while kernels launched on the then-active stream could still be reading it: it may be
            # handed to ANY stream now (the banded pipeline runs on its own three), so wait for that point of the freeing
            # stream first -- long past in practice, a few microseconds when it is not
            _lib.call("xrs_event_sync", ev)
            with _pool_lock:
                _spare_events.append(ev)
        return ptr
    _lib.require_device()
    p = ctypes.c_void_p()
    try:
        _lib.call("xrs_malloc", ctypes.byref(p), nbytes)
    except _lib.XrsError:
        empty_cache()
        _lib.call("xrs_malloc", ctypes.byref(p), nbytes)
    return p.value


def _raw_free(ptr: int, nbytes: int):
    global _pool_bytes
    nbytes = max(int(nbytes), 16)
    with _pool_lock:
        room = _pool_bytes + nbytes <= _POOL_MAX_BYTES
        ev = _spare_events.pop() if (room and _spare_events) else None
    if room:
        try:
            if ev is None:
                ev = ctypes.c_void_p()
                _lib.call("xrs_event_create", ctypes.byref(ev))
            _lib.call("xrs_event_record", ev, _active_stream())
        except Exception:                  # (no device / interpreter shutdown: pool the block without a fence)
            ev = None
        entry = (ptr, ev)                  # (built outside the lock: an allocation can start a GC pass)
        with _pool_lock:
            _pool.setdefault(nbytes, []).append(entry)
            _pool_bytes += nbytes
        return
    _lib.load().xrs_free(ptr)


def empty_cache():
    """Return every cached device buffer to the driver and drop the recycled host blocks."""
    global _pool_bytes, _host_pool_bytes, _pinned_pool_bytes, _pinned_total_bytes
    with _host_lock:
        _host_pool.clear()
        _host_pool_bytes = 0
        for nb, lst in _pinned_pool.items():
            for ptr in lst:
                _lib.load().xrs_host_free(ptr)
                _pinned_total_bytes -= nb
        _pinned_pool.clear()
        _pinned_pool_bytes = 0
    with _pool_lock:
        blocks = [entry for lst in _pool.values() for entry in lst]
        _pool.clear()
        _pool_bytes = 0
    for p, ev in blocks:
        _lib.load().xrs_free(p)
        if ev is not None:                 # the fence of the block: recycled for the next pooled block
            with _pool_lock:
                _spare_events.append(ev)


# ------------------------------------------------------------------ recycled host blocks for results
_host_pool = {}
_host_lock = threading.RLock()          # re-entrant: __del__ of a block can run (GC) while the lock is held
_host_pool_bytes = 0
_HOST_POOL_MAX_BYTES = int(os.environ.get("XRS_HOST_POOL_MAX_BYTES", 16 << 30))
_HOST_POOL_MIN_BLOCK = 1 << 20          # smaller results: plain np.empty


_pinned_pool = {}                        # nbytes -> [host pointers from xrs_host_alloc]
_pinned_pool_bytes = 0
_PINNED_POOL_MAX_BYTES = int(os.environ.get("XRS_PINNED_POOL_MAX_BYTES", 8 << 30))
_pinned_total_bytes = 0                  # every page-locked byte this process holds (results in use + free list)
_PINNED_TOTAL_MAX_BYTES = int(os.environ.get("XRS_PINNED_TOTAL_MAX_BYTES", 32 << 30))   # beyond: pageable blocks


class _HostBlock:
    """Owner of one recycled block.  NumPy arrays created from it keep it alive through `.base`; when the last
    of them (including any view the caller sliced off) is collected, the memory returns to the free list.
    Two kinds: pageable (`raw`, a uint8 ndarray that owns already-touched pages) and page-locked (`ptr` from
    xrs_host_alloc: device-to-host copies into it are asynchronous, which the banded pipeline of
    `_launch.stencil` needs to overlap them with the uploads)."""

    __slots__ = ("raw", "ptr", "nbytes", "__weakref__")

    def __init__(self, raw=None, ptr=0, nbytes=0):
        self.raw, self.ptr, self.nbytes = raw, ptr, nbytes

    @property
    def __array_interface__(self):
        if self.raw is not None:
            return self.raw.__array_interface__
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        global _host_pool_bytes, _pinned_pool_bytes, _pinned_total_bytes
        try:
            if self.raw is not None:
                raw = self.raw
                with _host_lock:
                    if _host_pool_bytes + raw.nbytes <= _HOST_POOL_MAX_BYTES:
                        _host_pool.setdefault(raw.nbytes, []).append(raw)
                        _host_pool_bytes += raw.nbytes
                return
            with _host_lock:
                if _pinned_pool_bytes + self.nbytes <= _PINNED_POOL_MAX_BYTES:
                    _pinned_pool.setdefault(self.nbytes, []).append(self.ptr)
                    _pinned_pool_bytes += self.nbytes
                    return
                _pinned_total_bytes -= self.nbytes
            _lib.load().xrs_host_free(self.ptr)
        except Exception:                  # interpreter shutdown
            pass


def host_empty(shape, dtype, pinned: bool = False) -> np.ndarray:
    """Like np.empty, but large arrays come from the recycled-block pools (contents undefined).
    `pinned`: page-locked memory (falls back to pageable if the driver refuses the allocation)."""
    global _host_pool_bytes, _pinned_pool_bytes, _pinned_total_bytes
    dtype = np.dtype(dtype)
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes < _HOST_POOL_MIN_BLOCK:
        return np.empty(shape, dtype)
    if pinned:
        ptr = 0
        with _host_lock:
            lst = _pinned_pool.get(nbytes)
            if lst:
                ptr = lst.pop()
                _pinned_pool_bytes -= nbytes
        if not ptr and _pinned_total_bytes + nbytes <= _PINNED_TOTAL_MAX_BYTES:
            p = ctypes.c_void_p()
            try:
                _lib.call("xrs_host_alloc", ctypes.byref(p), nbytes)
                ptr = p.value
                with _host_lock:
                    _pinned_total_bytes += nbytes
            except _lib.XrsError:
                ptr = 0
        if ptr:
            return np.asarray(_HostBlock(ptr=ptr, nbytes=nbytes)).view(dtype).reshape(shape)
    raw = None
    with _host_lock:
        lst = _host_pool.get(nbytes)
        if lst:
            raw = lst.pop()
            _host_pool_bytes -= nbytes
    if raw is None:
        raw = np.empty(nbytes, np.uint8)
    return np.asarray(_HostBlock(raw=raw)).view(dtype).reshape(shape)


def is_pinned(arr: np.ndarray) -> bool:
    """True if `arr` lives in a page-locked block handed out by host_empty(..., pinned=True)."""
    base = arr
    while isinstance(base, np.ndarray) and base.base is not None:
        base = base.base
    return isinstance(base, _HostBlock) and base.raw is None


_CAST_CODE = {np.dtype(t): c for c, t in enumerate(
    (np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64, np.float64))}


# XRS_DT_* codes of include/xrs_hip.h for every dtype a kernel can read in place (the cast codes + float32)
DTYPE_CODE = dict(_CAST_CODE)
DTYPE_CODE[np.dtype(np.float32)] = 9


def _cast_f32_on_device(src: "DeviceArray", stream=None, src_is_temporary=True) -> "DeviceArray":
    out = DeviceArray(src.shape, np.float32)
    _lib.call("xrs_cast_f32", src.ptr, _CAST_CODE[src.dtype], out.ptr, src.size, stream)
    if src_is_temporary:
        _lib.call("xrs_stream_sync", stream)      # a temporary goes back to the pool when the caller drops it
    return out


class DeviceArray:
    """C-contiguous n-d array in HBM.  Minimal duck array: shape / dtype / ndim / size / get()."""

    __array_priority__ = 1000

    def __init__(self, shape, dtype=np.float32, _ptr=None, _base=None):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._base = _base                  # a view keeps its owner alive
        if _ptr is None:
            self.ptr = _raw_alloc(self.nbytes)
            self._owns = True
        else:
            self.ptr = int(_ptr)
            self._owns = False

    # -- construction ---------------------------------------------------------
    @classmethod
    def from_numpy(cls, arr, dtype=None, stream=None):
        arr = np.ascontiguousarray(arr if dtype is None else np.asarray(arr).astype(dtype, copy=False))
        out = cls(arr.shape, arr.dtype)
        if arr.nbytes:
            _lib.call("xrs_memcpy_h2d", out.ptr, arr.ctypes.data, arr.nbytes, stream)
            _lib.call("xrs_stream_sync", stream)      # `arr` may be a temporary
        return out

    @classmethod
    def empty(cls, shape, dtype=np.float32):
        return cls(shape, dtype)

    # -- duck-array surface ---------------------------------------------------
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def __len__(self):
        return self.shape[0]

    def get(self, stream=None) -> np.ndarray:
        """Copy to a new NumPy array (cupy's spelling)."""
        out = host_empty(self.shape, self.dtype)
        if out.nbytes:
            _lib.call("xrs_memcpy_d2h", out.ctypes.data, self.ptr, out.nbytes, stream)
            _lib.call("xrs_stream_sync", stream)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype)

    def rows(self, start, stop):
        """View of rows [start, stop) (first axis), sharing this buffer."""
        row_bytes = self.nbytes // self.shape[0] if self.shape[0] else 0
        return DeviceArray((stop - start,) + self.shape[1:], self.dtype,
                           _ptr=self.ptr + start * row_bytes, _base=self)

    def astype(self, dtype):
        if np.dtype(dtype) == self.dtype:
            return self
        if np.dtype(dtype) == np.float32 and self.dtype in _CAST_CODE:
            # on the ACTIVE stream: the wrappers launch their kernels on _launch.get_stream(), and a non-blocking stream
            # does not order against the NULL stream the cast would otherwise run on
            from ._launch import get_stream
            return _cast_f32_on_device(self, stream=get_stream(), src_is_temporary=False)   # (the caller's array outlives the kernel)
        return DeviceArray.from_numpy(self.get().astype(dtype))

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, ptr=0x{self.ptr:x})"

    # xarray treats objects with these hooks as duck arrays and leaves them wrapped
    def __array_function__(self, func, types, args, kwargs):
        return NotImplemented

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        return NotImplemented

    def __del__(self):
        try:
            if getattr(self, "_owns", False) and self.ptr:
                _raw_free(self.ptr, self.nbytes)
                self.ptr = 0
        except Exception:
            pass


def is_device_array(x) -> bool:
    return isinstance(x, DeviceArray)


def to_device_f32(data) -> DeviceArray:
    """`data.astype(np.float32)` of the reference runners, landing in HBM."""
    if isinstance(data, DeviceArray):
        return data.astype(np.float32)
    host = np.asarray(data)
    if host.dtype != np.float32 and host.dtype in _CAST_CODE and host.size:
        from ._launch import get_stream
        return _cast_f32_on_device(DeviceArray.from_numpy(host), stream=get_stream())      # native dtype over PCIe, converted in HBM
    return DeviceArray.from_numpy(host, dtype=np.float32)


def synchronize():
    _lib.call("xrs_device_sync")
