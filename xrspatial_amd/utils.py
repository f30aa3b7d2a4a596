"""Backend dispatch and resolution helpers (host side).

Mirrors the parts of the reference's xrspatial/utils.py that sit on the hot
path: ArrayTypeFunctionMapping (:117-143) -- here with an extra `hip_func` slot
for device-resident data --, validate_arrays (:146-165), calc_res (:204-230),
get_dataarray_resolution (:233-277), not_implemented_func (:113-114).
"""
from __future__ import annotations

import functools
import threading

import numpy as np

from . import _lib
from .device import DeviceArray
from .sharded import ShardedArray

try:                                     # optional, like the reference
    import dask.array as da              # pragma: no cover
except ImportError:
    da = None


def has_hip() -> bool:
    """True when libxrs_hip.so is built and an MI355X is visible."""
    return _lib.device_available()


def has_dask_array() -> bool:
    return da is not None


def is_dask(data) -> bool:
    return da is not None and isinstance(data, da.Array)


# dask's threaded scheduler calls a block function from several threads at once; the numpy runners share one stream and
# reuse staging buffers between calls, and the device serialises the launches anyway: one block at a time
_DASK_BLOCK_LOCK = threading.Lock()


def _run_block(block_func, args, kwargs, *blocks):
    """One dask block (or one block of each raster) through a numpy runner.  Module level on purpose: dask pickles the
    block function for its process / distributed schedulers, and a closure defined inside dask_overlap would drag the
    lock below along by value ("cannot pickle '_thread.lock'"); here it is looked up when the block runs."""
    with _DASK_BLOCK_LOCK:
        return np.asarray(block_func(*[np.ascontiguousarray(b) for b in blocks], *args, **kwargs))


def dask_overlap(block_func, depth, meta=None):
    """The dask slot of a stencil runner.  The reference wraps its numpy runner in
    `data.map_overlap(func, depth=depth, boundary=np.nan, meta=np.array(()))` (slope.py:86-97, aspect.py:151-160,
    curvature.py:56-59, hillshade.py:42-45, focal.py:70-75 and 329-340, convolution.py:316-327); so does this: every
    block (with `depth` cells of its neighbours, NaN beyond the raster) goes through this package's numpy runner --
    staged through HBM, computed by the HIP kernels, brought back -- and dask trims the overlap.  Lazy like upstream:
    nothing runs before `.compute()`.  (Rasters that fit one node's GPUs are better served as a `ShardedArray`; this slot
    exists so that a dask-backed DataArray that worked upstream works here.)  `meta` is the reference's, float64 `np.array(())`
    over float32 blocks included: what `.dtype` says before `.compute()` is upstream's answer too."""
    def run(data, *args, **kwargs):
        if not np.issubdtype(data.dtype, np.floating):
            data = data.astype(np.float32)      # (a NaN boundary needs a float raster; the runners cast to float32 anyway)
        return data.map_overlap(functools.partial(_run_block, block_func, args, kwargs), depth=depth, boundary=np.nan,
                                meta=np.array(()) if meta is None else meta)
    return run


def dask_blocks(block_func):
    """The dask slot of a per-cell runner over one or more equally chunked rasters: `da.map_blocks(func, *arrays,
    meta=np.array(()))` around the numpy runner (multispectral.py:60-63, 205-208, 845-848 ...)."""
    def run(*arrays):
        return da.map_blocks(functools.partial(_run_block, block_func, (), {}), *arrays, meta=np.array(()))
    return run


def not_implemented_func(agg, *args, messages='Not yet implemented.'):
    raise NotImplementedError(messages)


class ArrayTypeFunctionMapping(object):
    """Pick the runner for `type(agg.data)` (reference: utils.py:117-143).

    numpy-backed data is served by `numpy_func` (which, in this package, stages
    through HBM and runs the HIP kernels -- there is no Numba path);
    DeviceArray-backed data by `hip_func` (stays in HBM).  dask-backed data goes
    to `dask_func` when dask is installed.  Anything else: TypeError, as upstream.
    """

    def __init__(self, numpy_func, hip_func=None, dask_func=None, cupy_func=None, dask_cupy_func=None,
                 sharded_func=None):
        self.numpy_func = numpy_func
        self.hip_func = hip_func
        self.sharded_func = sharded_func        # row-sharded multi-GPU rasters (xrspatial_amd.sharded), dask's role upstream
        self.dask_func = dask_func
        self.cupy_func = cupy_func
        self.dask_cupy_func = dask_cupy_func

    def __call__(self, arr):
        if isinstance(arr.data, np.ndarray):
            return self.numpy_func
        if isinstance(arr.data, DeviceArray):
            if self.hip_func is None:
                raise NotImplementedError("not implemented for device-resident arrays")
            return self.hip_func
        if isinstance(arr.data, ShardedArray):
            if self.sharded_func is None:
                raise NotImplementedError("not implemented for row-sharded (multi-GPU) arrays")
            return self.sharded_func
        if da is not None and isinstance(arr.data, da.Array):   # pragma: no cover
            if self.dask_func is None:
                raise NotImplementedError("not implemented for dask-backed arrays")
            return self.dask_func
        raise TypeError("Unsupported Array Type: {}".format(type(arr)))


def validate_arrays(*arrays):
    """All rasters must share one shape and one array backend (same errors as upstream, utils.py:146-165)."""
    if len(arrays) < 2:
        raise ValueError("validate_arrays() input must contain 2 or more arrays")
    head = arrays[0].data
    for other in arrays[1:]:
        if tuple(other.data.shape) != tuple(head.shape):
            raise ValueError("input arrays must have equal shapes")
        if not isinstance(head, type(other.data)):
            raise ValueError("input arrays must have same type")


def _coordinate_extent(raster, dim):
    coord = raster[dim]
    return coord.min().item(), coord.max().item()


def get_xy_range(raster, xdim=None, ydim=None):
    """((xmin, xmax), (ymin, ymax)) of the raster's last two coordinate axes (upstream: utils.py:168-201)."""
    xdim = raster.dims[-1] if xdim is None else xdim
    ydim = raster.dims[-2] if ydim is None else ydim
    return _coordinate_extent(raster, xdim), _coordinate_extent(raster, ydim)


def calc_res(raster, xdim=None, ydim=None):
    """(xres, yres): coordinate extent divided by (cells - 1) per axis (upstream: utils.py:204-230)."""
    (x_lo, x_hi), (y_lo, y_hi) = get_xy_range(raster, xdim, ydim)
    n_rows, n_cols = raster.shape[-2:]
    return (x_hi - x_lo) / (n_cols - 1), (y_hi - y_lo) / (n_rows - 1)


def get_dataarray_resolution(agg, xdim=None, ydim=None):
    """Cell size: attrs['res'] when it is a number or a pair of numbers, otherwise derived from the coordinates
    (upstream: utils.py:233-277)."""
    plain_number = (int, float)
    res = getattr(agg, "attrs", {}).get("res") if isinstance(getattr(agg, "attrs", None), dict) else None
    if isinstance(res, plain_number):
        return res, res
    if isinstance(res, (tuple, list, np.ndarray)) and len(res) == 2 and all(isinstance(r, plain_number) for r in res):
        return res[0], res[1]
    return calc_res(agg, xdim, ydim)
