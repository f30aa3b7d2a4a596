"""Focal statistics.  Reference: xrspatial/focal.py (mean :162-265, apply :343-473,
focal_stats :800-878, hotspots :881-1125)."""
from __future__ import annotations

import copy
import ctypes
import os

import numpy as np

from . import _lib, fused
from ._launch import finish, get_stream, pipeline_ok, pipelined_rows, plane_args, sharded_f32
from ._xr import DataArray
from .convolution import _kernel_f64, custom_kernel
from .dataset_support import supports_dataset
from .device import DeviceArray, to_device_f32
from .sharded import ShardedArray, ShardedStack
from .utils import ArrayTypeFunctionMapping, da, dask_overlap, is_dask

# order of the XRS_STAT_* enum in include/xrs_hip.h
_STAT_INDEX = {'mean': 0, 'max': 1, 'min': 2, 'range': 3, 'std': 4, 'var': 5, 'sum': 6}


# Accuracy options of the large-window statistics (include/xrs_hip.h: xrs_focal_stats_f32_ex).  Windows of 9x9 cells and
# more run float32 walkers whose mean / var / std / sum agree with the reference's float64 accumulators to <= 2e-6 relative
# (a per-output guard sends ill-conditioned tiles to the exact kernels).  Set
#   options['moments'] = 'exact'      (or XRS_FOCAL_MOMENTS=exact)      float64 running sums for mean / var / std: ~1 ulp;
#   options['sum'] = 'sequential'     (or XRS_FOCAL_SUM=sequential)     `sum` bit-identical to numba's float32 nansum
# to keep whole launches on the exact kernels (about 2x the time).  A value set in `options` wins over the environment;
# None (the initial state) = the environment variable if set, else 'fast' / 'rounded'.
options = {'moments': None, 'sum': None}
_FLAG_EXACT_MOMENTS, _FLAG_SEQUENTIAL_SUM = 1, 2


def _focal_flags():
    import os
    moments = options.get('moments') or os.environ.get('XRS_FOCAL_MOMENTS') or 'fast'
    sums = options.get('sum') or os.environ.get('XRS_FOCAL_SUM') or 'rounded'
    if moments not in ('fast', 'exact'):
        raise ValueError(f"focal moments option must be 'fast' or 'exact', got {moments!r}")
    if sums not in ('rounded', 'sequential'):
        raise ValueError(f"focal sum option must be 'rounded' or 'sequential', got {sums!r}")
    return (_FLAG_EXACT_MOMENTS if moments == 'exact' else 0) | (_FLAG_SEQUENTIAL_SUM if sums == 'sequential' else 0)


def _stats_call(in_ptr, ptrs, mask, rows, cols, ld_in, ld_out, k, work, ht, hb, stream):
    _lib.call("xrs_focal_stats_f32_ex", in_ptr, ptrs, mask, rows, cols, ld_in, ld_out, k.ctypes.data, k.shape[0], k.shape[1],
              work.ptr if work is not None else None, work.nbytes if work is not None else 0, ht, hb, _focal_flags(), stream)


class _BuiltinReducer:
    """Stands in for the reference's `@ngjit _calc_*` functions (focal.py:268-302).

    `apply(raster, kernel, func=_calc_sum)` upstream takes a Numba-compiled callable; the built-in
    reducers are exported under the same names as tokens that `apply` recognises and runs entirely
    on the MI355X.  Any other callable runs on the host on windows the device gathers
    (`_apply_callable`; also for row-sharded rasters, whose windows reach into the halo rows)."""

    def __init__(self, stat):
        self.stat = stat

    def __repr__(self):
        return f"<built-in focal reducer '{self.stat}'>"


_calc_mean = _BuiltinReducer('mean')
_calc_sum = _BuiltinReducer('sum')
_calc_min = _BuiltinReducer('min')
_calc_max = _BuiltinReducer('max')
_calc_std = _BuiltinReducer('std')
_calc_range = _BuiltinReducer('range')
_calc_var = _BuiltinReducer('var')


def _focal_stats_hip(data, kernel, stats, stacked=None):
    """One pass over `data`, all requested statistics; returns {stat: array}.

    `stacked`: optional (len(stats), rows, cols) DeviceArray whose planes receive the results."""
    _lib.require_device()
    like_numpy = not isinstance(data, DeviceArray)
    k = _kernel_f64(kernel)
    src = to_device_f32(data)
    rows, cols, ld = plane_args(src)
    if stacked is not None:
        outs = {s: DeviceArray((rows, cols), np.float32, _ptr=stacked.ptr + i * rows * cols * 4, _base=stacked)
                for i, s in enumerate(stats)}
    else:
        outs = {s: DeviceArray((rows, cols), np.float32) for s in dict.fromkeys(stats)}
    ptrs = (ctypes.c_void_p * 7)()
    mask = 0
    for s, arr in outs.items():
        ptrs[_STAT_INDEX[s]] = arr.ptr
        mask |= 1 << _STAT_INDEX[s]
    stream = get_stream()
    work = _window_workspace(k, rows, cols)
    _stats_call(src.ptr, ptrs, mask, rows, cols, ld, ld, k, work, 0, 0, stream)
    if like_numpy:
        return {s: arr.get(stream) for s, arr in outs.items()}
    return outs


def _focal_stats_banded(host, kernel, stats):
    """Large numpy-backed rasters: row bands upload / compute / download concurrently (_launch.pipelined_rows);
    returns the (len(stats), rows, cols) float32 stack."""
    _lib.require_device()
    k = _kernel_f64(kernel)
    cols = host.shape[1]
    mask = 0
    for s in stats:
        mask |= 1 << _STAT_INDEX[s]

    def launch(in_ptr, out_ptrs, n_rows, ht, hb, stream):
        ptrs = (ctypes.c_void_p * 7)()
        for s, ptr in zip(stats, out_ptrs):
            ptrs[_STAT_INDEX[s]] = ptr
        # one workspace per band launch (the bands run on several streams: no shared tile map) -- np.ones boxes then take the
        # separable walk here as they do for device-resident rasters, so both backends give the same var / std
        work = _window_workspace(k, n_rows, cols)
        _stats_call(in_ptr, ptrs, mask, n_rows, cols, cols, cols, k, work, ht, hb, stream)
        keep.append(work)

    # (the workspaces live until pipelined_rows has synchronised its streams: a block freed earlier is fenced on the GLOBAL
    # stream by device.py, not on the band's, and could be handed out again while the band's kernels still use it)
    keep = []
    return pipelined_rows(host, [np.float32] * len(stats), launch, k.shape[0] // 2)


def _window_workspace(k, rows=0, cols=0):
    """Device scratch of a launch, or None: windows beyond the tiled kernels' 63 x 63 read the mask from a device copy of the
    kernel (csrc/kxk_big.hip); np.ones((k, k)) masks -- the reference's benchmark kernels -- get the tile map of the
    separable box walk (csrc/boxsep.hip; xrs_focal_workspace_bytes).  The block goes back to the pool behind the launch on
    the launch's stream (device.py fences recycled blocks with an event), so nobody has to wait for it."""
    big = max(k.shape) > 63
    # 7x7 .. 25x25: the large-window kernels (circles, boxes, annuli; mean / sum and the moments) note the tiles their fast
    # walks hand on -- the rim of a nodata region, dense nodata -- in a work-list inside this block (csrc/mom_impl.h:
    # focal_mom_rescue_kernel, csrc/wide_impl.h: focal_wide_rescue_kernel)
    walked = k.shape[0] == k.shape[1] and 7 <= k.shape[0] <= 25 and k.shape[0] % 2 == 1
    if os.environ.get("XRS_MOM_RESCUE") == "0":            # (A/B: no work-list, slow tiles are walked in place as before round 6)
        walked = bool((k == 1.0).all())
    if not (big or walked):
        return None
    return DeviceArray((int(_lib.load().xrs_focal_workspace_bytes(int(rows), int(cols), k.shape[0], k.shape[1])),), np.uint8)


def _apply_sharded(data, kernel, stat):
    # the reference's dask path: map_overlap(depth=k//2, boundary=nan) (focal.py:165-176, 343-356)
    _lib.require_device()
    k = _kernel_f64(kernel)
    src = sharded_f32(data)
    stream = get_stream()
    ht, hb = src.halos(k.shape[0] // 2, stream)
    rows, cols = src.shape
    out = src.like(np.float32)
    ptrs = (ctypes.c_void_p * 7)()
    ptrs[_STAT_INDEX[stat]] = out.ptr
    work = _window_workspace(k, rows, cols)
    _stats_call(src.ptr, ptrs, 1 << _STAT_INDEX[stat], rows, cols, cols, cols, k, work, ht, hb, stream)
    return out


def _focal_stats_sharded(data, kernel, stats):
    """All requested statistics of a sharded raster in one pass; {stat: ShardedArray}."""
    _lib.require_device()
    k = _kernel_f64(kernel)
    src = sharded_f32(data)
    stream = get_stream()
    ht, hb = src.halos(k.shape[0] // 2, stream)
    rows, cols = src.shape
    outs = {s: src.like(np.float32) for s in dict.fromkeys(stats)}
    ptrs = (ctypes.c_void_p * 7)()
    mask = 0
    for s, arr in outs.items():
        ptrs[_STAT_INDEX[s]] = arr.ptr
        mask |= 1 << _STAT_INDEX[s]
    work = _window_workspace(k, rows, cols)
    _stats_call(src.ptr, ptrs, mask, rows, cols, cols, cols, k, work, ht, hb, stream)
    return outs


def _mean_sharded(data, excludes, passes):
    # every pass reads one row from either neighbour: exchange, launch, repeat (the reference: map_overlap per pass)
    _lib.require_device()
    cur = data if data.dtype in (np.float32, np.float64) else data.astype(np.float64)
    ex = np.asarray(list(excludes), dtype=np.float64)
    stream = get_stream()
    rows, cols = cur.shape
    if int(passes) < 1:
        return cur.astype(np.float64)
    for _ in range(int(passes)):
        ht, hb = cur.halos(1, stream)
        out = cur.like(np.float64)
        _lib.call("xrs_focal_mean3x3", cur.ptr, int(cur.dtype == np.float64), out.ptr, rows, cols, cols, cols,
                  ex.ctypes.data, len(ex), ht, hb, stream)
        _lib.call("xrs_stream_sync", stream)        # (`cur` of the previous pass is released below)
        cur = out
    return cur


def _mean_hip(data, excludes, passes):
    # replaces the passes loop over _mean_numpy (focal.py:44-67, 257-259); float64 result
    _lib.require_device()
    like_numpy = not isinstance(data, DeviceArray)
    if isinstance(data, DeviceArray):
        cur = data if data.dtype in (np.float32, np.float64) else data.astype(np.float64)
    else:
        host = np.asarray(data)
        # float32 rasters are widened on the fly by the first pass; everything else is cast like
        # the reference's `.astype(float)`
        cur = DeviceArray.from_numpy(host if host.dtype == np.float32 else host.astype(np.float64, copy=False))
    rows, cols = cur.shape
    ex = np.asarray(list(excludes), dtype=np.float64)
    stream = get_stream()
    passes = int(passes)
    if passes < 1:
        out = cur if cur.dtype == np.float64 else cur.astype(np.float64)  # `agg.data.astype(float)` only
        return finish(out, like_numpy)
    out = DeviceArray((rows, cols), np.float64)
    scratch = DeviceArray((rows, cols), np.float64) if passes > 1 else None
    _lib.call("xrs_focal_mean3x3_passes", cur.ptr, int(cur.dtype == np.float64), out.ptr,
              scratch.ptr if scratch is not None else None, passes, rows, cols, ex.ctypes.data, len(ex), stream)
    if scratch is not None or like_numpy:
        _lib.call("xrs_stream_sync", stream)        # `scratch` / `ex` must outlive the launches
    return finish(out, like_numpy)


def _mean_dask(data, excludes, passes):
    # focal.py:70-75, 257-259: `.astype(float)`, then one map_overlap(depth=(1, 1), boundary=nan) per pass
    out = data.astype(float)
    one_pass = dask_overlap(_mean_hip, (1, 1))
    for _ in range(int(passes)):
        out = one_pass(out, excludes, 1)
    return out


@supports_dataset
def mean(agg, passes=1, excludes=[np.nan], name='mean'):
    """3x3 NaN-skipping moving average, `passes` times; cells equal to a value in `excludes`
    are passed through.  Same signature and results (float64) as `xrspatial.focal.mean`."""
    if len(agg.shape) != 2:
        raise ValueError("`agg` must be 2D")
    if len(excludes) > 8:
        raise ValueError("at most 8 exclude values are supported by the MI355X backend")
    mapper = ArrayTypeFunctionMapping(numpy_func=_mean_hip, hip_func=_mean_hip, sharded_func=_mean_sharded,
                                      dask_func=_mean_dask)
    out = mapper(agg)(agg.data, tuple(excludes), passes)
    return DataArray(out, name=name, dims=agg.dims, coords=agg.coords, attrs=agg.attrs)


def _reducer_name(func):
    """The built-in statistic `func` stands for, or None for a user callable."""
    if isinstance(func, _BuiltinReducer):
        return func.stat
    if isinstance(func, str) and func in _STAT_INDEX:
        return func
    if callable(func):
        return None
    raise TypeError(
        "apply(): `func` must be one of the built-in reducers (_calc_mean/_calc_sum/_calc_min/_calc_max/_calc_std/"
        f"_calc_var/_calc_range) or a callable taking the kernel-shaped window; got {func!r}")


_WINDOW_BAND_BYTES = 256 << 20          # gathered windows held on the device / crossing PCIe per band


def _apply_callable(data, kernel, func):
    """focal.apply with a user callable (focal.py:305-326): the MI355X gathers, for a band of rows at a time, the
    kernel-shaped float32 window of every cell (NaN outside the raster and where the kernel is not 1 -- exactly the
    array _apply_numpy fills); `func` runs on the host on each window.  float32 result, a numpy array."""
    _lib.require_device()
    stream = get_stream()
    k = _kernel_f64(kernel)
    kr, kc = k.shape
    ht = hb = 0
    if isinstance(data, ShardedArray):
        # the dask slot of the reference (focal.py:329-340: map_overlap(depth=k//2, boundary=nan) around _apply_numpy):
        # the gather reads the neighbours' rows in the shard's halo, so a window is cut only at the raster's true edge
        src = sharded_f32(data)
        ht, hb = src.halos(kr // 2, stream)
    else:
        src = to_device_f32(data)
    rows, cols = src.shape
    out = np.zeros((rows, cols), np.float32)
    if rows == 0 or cols == 0:
        return out
    first = src.ptr - ht * cols * 4                           # the plane the gather sees: halo rows + owned rows
    seen = rows + ht + hb
    per_row = cols * kr * kc * 4
    band = int(max(1, min(rows, _WINDOW_BAND_BYTES // per_row)))
    wdev = DeviceArray((band, cols, kr, kc), np.float32)
    for y0 in range(0, rows, band):
        nb = min(band, rows - y0)
        _lib.call("xrs_focal_windows_f32", first, wdev.ptr, seen, cols, cols, y0 + ht, nb, k.ctypes.data, kr, kc, stream)
        win = wdev.get(stream)[:nb]
        for y in range(nb):
            row_w, row_o = win[y], out[y0 + y]
            for x in range(cols):
                row_o[x] = func(row_w[x])
    return out


def apply(raster, kernel, func=_calc_mean, name='focal_apply'):
    """Reduce the cells under `kernel == 1` around every cell with `func` (default: mean).

    Same signature as `xrspatial.focal.apply`; window clipped at the raster edge, NaN
    cells skipped, float32 result.  The built-in reducers run entirely on the MI355X; any other callable gets the
    kernel-shaped float32 window of each cell (gathered on the device, `_apply_callable`) and runs on the host.
    Row-sharded rasters (`ShardedArray`) take both forms: the reference's dask slot, map_overlap(depth = k // 2,
    boundary = nan) (focal.py:329-340)."""
    if not isinstance(raster, DataArray):
        raise TypeError("`raster` must be instance of DataArray")
    if raster.ndim != 2:
        raise ValueError("`raster` must be 2D")
    kernel = custom_kernel(kernel)
    stat = _reducer_name(func)
    scope = fused.current()
    if scope is not None and stat == 'mean':
        return scope.defer('focal_mean', raster, name, {'kernel': _kernel_f64(kernel)})

    if stat is None and is_dask(raster.data):
        # focal.py:329-340 (`_apply_dask_numpy`): the user's callable per chunk, lazily -- the windows of a block are
        # gathered on the device and reduced on the host when the block is computed, never the whole raster at once
        out = dask_overlap(_apply_callable, (kernel.shape[0] // 2, kernel.shape[1] // 2))(raster.data, kernel, func)
        return DataArray(out, name=name, coords=raster.coords, dims=raster.dims, attrs=raster.attrs)
    if stat is None:
        out = _apply_callable(raster.data, kernel, func)
        if isinstance(raster.data, ShardedArray):             # the result is a shard again, like every other operator's
            out = ShardedArray.from_numpy(out, raster.data.comm, raster.data.halo_cap)
        return DataArray(out, name=name, coords=raster.coords, dims=raster.dims, attrs=raster.attrs)

    def run(data, kernel, stat):
        if pipeline_ok(data) and max(np.asarray(kernel).shape) // 2 < 128:
            return _focal_stats_banded(data, kernel, [stat])[0]
        return _focal_stats_hip(data, kernel, [stat])[stat]

    # (dask: focal.py:329-340 -- map_overlap(depth = k // 2, boundary = nan) around the numpy runner)
    mapper = ArrayTypeFunctionMapping(numpy_func=run, hip_func=run, sharded_func=_apply_sharded,
                                      dask_func=dask_overlap(run, (kernel.shape[0] // 2, kernel.shape[1] // 2)))
    out = mapper(raster)(raster.data, kernel, stat)
    return DataArray(out, name=name, coords=raster.coords, dims=raster.dims, attrs=raster.attrs)


def focal_stats(agg, kernel, stats_funcs=['mean', 'max', 'min', 'range', 'std', 'var', 'sum']):
    """All requested focal statistics as a 3-D (stats, y, x) float32 DataArray.

    Same signature and results as `xrspatial.focal.focal_stats`; the reference makes one
    full pass per statistic, this backend computes them in a single pass over the raster."""
    if not isinstance(agg, DataArray):
        raise TypeError("`agg` must be instance of DataArray")
    if agg.ndim != 2:
        raise ValueError("`agg` must be 2D")
    kernel = custom_kernel(kernel)
    stats_funcs = list(stats_funcs)
    for s in stats_funcs:
        if s not in _STAT_INDEX:
            raise KeyError(s)
    if isinstance(agg.data, ShardedArray):
        planes = _focal_stats_sharded(agg.data, kernel, stats_funcs)
        coords = dict(agg.coords.items())
        coords['stats'] = np.array(stats_funcs, dtype=object)
        return DataArray(ShardedStack([planes[s] for s in stats_funcs]), dims=('stats',) + tuple(agg.dims), coords=coords,
                         attrs=agg.attrs)
    if is_dask(agg.data):
        # the reference runs one apply() per statistic and concatenates them (focal.py:782-797); per block the numpy runner
        # computes the requested statistic (one launch), map_overlap(depth = k // 2, boundary = nan) as in apply()
        def one(block, stat):
            return _focal_stats_hip(block, kernel, [stat])[stat]
        depth = (kernel.shape[0] // 2, kernel.shape[1] // 2)
        stacked = da.stack([dask_overlap(one, depth)(agg.data, s) for s in stats_funcs])
        coords = dict(agg.coords.items())
        coords['stats'] = np.array(stats_funcs, dtype=object)
        return DataArray(stacked, dims=('stats',) + tuple(agg.dims), coords=coords, attrs=agg.attrs)
    if not isinstance(agg.data, (np.ndarray, DeviceArray)):
        raise TypeError("Unsupported Array Type: {}".format(type(agg)))
    if pipeline_ok(agg.data) and len(set(stats_funcs)) == len(stats_funcs) and max(kernel.shape) // 2 < 128:
        stacked = _focal_stats_banded(agg.data, kernel, stats_funcs)
    elif isinstance(agg.data, np.ndarray) and len(set(stats_funcs)) == len(stats_funcs):
        # the planes are produced side by side in one device buffer and come back in ONE copy
        dev = DeviceArray((len(stats_funcs),) + tuple(agg.shape), np.float32)
        _focal_stats_hip(to_device_f32(agg.data), kernel, stats_funcs, stacked=dev)
        stacked = dev.get(get_stream())
    elif isinstance(agg.data, np.ndarray):
        planes = _focal_stats_hip(agg.data, kernel, stats_funcs)
        stacked = np.stack([planes[s] for s in stats_funcs])
    else:
        if len(set(stats_funcs)) != len(stats_funcs):
            raise ValueError("duplicate statistics requested")
        stacked = DeviceArray((len(stats_funcs),) + tuple(agg.shape), np.float32)
        _focal_stats_hip(agg.data, kernel, stats_funcs, stacked=stacked)
    coords = dict(agg.coords.items())
    coords['stats'] = np.array(stats_funcs, dtype=object)
    return DataArray(stacked, dims=('stats',) + tuple(agg.dims), coords=coords, attrs=agg.attrs)


def _hotspots_hip(data, kernel):
    # replaces _hotspots_numpy (focal.py:914-934)
    from .convolution import _convolve_2d_hip
    like_numpy = not isinstance(data, DeviceArray)
    if not (issubclass(data.dtype.type, np.integer) or issubclass(data.dtype.type, np.floating)):
        raise ValueError("data type must be integer or float")
    _lib.require_device()
    src = to_device_f32(data)
    k = np.asarray(kernel, dtype=np.float64)
    mean_array = _convolve_2d_hip(src, k / k.sum())
    stream = get_stream()
    mom = DeviceArray((4,), np.float64)
    _lib.call("xrs_nan_moments_f32", src.ptr, src.size, mom.ptr, stream)
    raw = mom.get(stream)
    count = int(raw[0:1].view(np.uint64)[0])
    with np.errstate(all="ignore"):
        global_mean = np.float32(raw[3])
        global_std = np.float32(np.sqrt(raw[2] / count)) if count else np.float32(np.nan)
    if global_std == 0:
        raise ZeroDivisionError("Standard deviation of the input raster values is 0.")
    out = DeviceArray(src.shape, np.int8)
    _lib.call("xrs_hotspots_classify_f32", mean_array.ptr, out.ptr, out.size, float(global_mean), float(global_std),
              stream)
    return finish(out, like_numpy)


def _hotspots_sharded(data, kernel):
    # the reference's dask path (focal.py:940-984): block-wise convolution, GLOBAL mean / std, block-wise classification.
    # Here every rank reduces its rows to (count, mean, sum of squared deviations); the triples are combined pairwise
    # (Chan et al.) after ONE small all-reduce, so every rank classifies against the same global moments.
    from .convolution import _convolve_2d_sharded
    if not (issubclass(data.dtype.type, np.integer) or issubclass(data.dtype.type, np.floating)):
        raise ValueError("data type must be integer or float")
    _lib.require_device()
    src = sharded_f32(data)
    k = np.asarray(kernel, dtype=np.float64)
    mean_array = _convolve_2d_sharded(src, k / k.sum())
    stream = get_stream()
    mom = DeviceArray((4,), np.float64)
    _lib.call("xrs_nan_moments_f32", src.ptr, src.size, mom.ptr, stream)
    raw = mom.get(stream)
    count = float(raw[0:1].view(np.uint64)[0])
    mine = np.array([count, raw[3] if count else 0.0, raw[2] if count else 0.0])
    with np.errstate(all="ignore"):
        if src.world > 1:
            slots = np.zeros((src.world, 3))
            slots[src.rank] = mine
            slots = src.comm.allreduce(slots, 'sum')
        else:
            slots = mine[None, :]
        n = slots[:, 0].sum()
        if n:
            mean = (slots[:, 0] * slots[:, 1]).sum() / n
            ssd = slots[:, 2].sum() + (slots[:, 0] * (slots[:, 1] - mean) ** 2).sum()
            global_mean, global_std = np.float32(mean), np.float32(np.sqrt(ssd / n))
        else:
            global_mean = global_std = np.float32(np.nan)
    if global_std == 0:
        raise ZeroDivisionError("Standard deviation of the input raster values is 0.")
    out = src.like(np.int8)
    _lib.call("xrs_hotspots_classify_f32", mean_array.ptr, out.ptr, out.size, float(global_mean), float(global_std),
              stream)
    return out


def _hotspots_dask(data, kernel):
    # focal.py:940-976 -- pass 1: the two global scalars, computed eagerly (all chunks read once, 8 bytes kept); pass 2:
    # convolution + z-score + classification fused per chunk under map_overlap(depth = k // 2, boundary = nan), int8 out
    from .convolution import _convolve_2d_hip
    if not np.issubdtype(data.dtype, np.floating):
        data = data.astype(np.float32)
    global_mean, global_std = da.compute(da.nanmean(data), da.nanstd(data))
    global_mean, global_std = np.float32(global_mean), np.float32(global_std)
    if global_std == 0:
        raise ZeroDivisionError("Standard deviation of the input raster values is 0.")
    k = np.asarray(kernel, dtype=np.float64)
    norm = k / k.sum()

    def chunk(block):
        src = to_device_f32(block)
        mean_array = _convolve_2d_hip(src, norm)
        out = DeviceArray(src.shape, np.int8)
        _lib.call("xrs_hotspots_classify_f32", mean_array.ptr, out.ptr, out.size, float(global_mean), float(global_std),
                  get_stream())
        return finish(out, True)

    return dask_overlap(chunk, (k.shape[0] // 2, k.shape[1] // 2), meta=np.array((), dtype=np.int8))(data)


def hotspots(raster, kernel):
    """Getis-Ord Gi* hot / cold spots: int8 raster of {0, +-90, +-95, +-99} confidence levels.

    Same signature and behaviour as `xrspatial.focal.hotspots`: neighbourhood mean by `convolve_2d` with the
    normalised kernel, z-score against the raster's global nanmean / nanstd, ZeroDivisionError for a constant
    raster, `attrs['unit'] = '%'`."""
    if not isinstance(raster, DataArray):
        raise TypeError("`raster` must be instance of DataArray")
    if raster.ndim != 2:
        raise ValueError("`raster` must be 2D")
    mapper = ArrayTypeFunctionMapping(numpy_func=_hotspots_hip, hip_func=_hotspots_hip, sharded_func=_hotspots_sharded,
                                      dask_func=_hotspots_dask)
    out = mapper(raster)(raster.data, kernel)
    attrs = copy.deepcopy(raster.attrs)
    attrs['unit'] = '%'
    return DataArray(out, coords=raster.coords, dims=raster.dims, attrs=attrs)
