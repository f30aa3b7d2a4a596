"""Backend dispatch and resolution helpers (host side).

Mirrors the parts of the reference's xrspatial/utils.py that sit on the hot
path: ArrayTypeFunctionMapping (:117-143) -- here with an extra `hip_func` slot
for device-resident data --, validate_arrays (:146-165), calc_res (:204-230),
get_dataarray_resolution (:233-277), not_implemented_func (:113-114).
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .device import DeviceArray

try:                                     # optional, like the reference
    import dask.array as da              # pragma: no cover
except ImportError:
    da = None


def has_hip() -> bool:
    """True when libxrs_hip.so is built and an MI355X is visible."""
    return _lib.device_available()


def has_dask_array() -> bool:
    return da is not None


def not_implemented_func(agg, *args, messages='Not yet implemented.'):
    raise NotImplementedError(messages)


class ArrayTypeFunctionMapping(object):
    """Pick the runner for `type(agg.data)` (reference: utils.py:117-143).

    numpy-backed data is served by `numpy_func` (which, in this package, stages
    through HBM and runs the HIP kernels -- there is no Numba path);
    DeviceArray-backed data by `hip_func` (stays in HBM).  dask-backed data goes
    to `dask_func` when dask is installed.  Anything else: TypeError, as upstream.
    """

    def __init__(self, numpy_func, hip_func=None, dask_func=None, cupy_func=None, dask_cupy_func=None):
        self.numpy_func = numpy_func
        self.hip_func = hip_func
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
        if da is not None and isinstance(arr.data, da.Array):   # pragma: no cover
            if self.dask_func is None:
                raise NotImplementedError("not implemented for dask-backed arrays")
            return self.dask_func
        raise TypeError("Unsupported Array Type: {}".format(type(arr)))


def validate_arrays(*arrays):
    """Equal shapes and equal array types (reference: utils.py:146-165)."""
    if len(arrays) < 2:
        raise ValueError("validate_arrays() input must contain 2 or more arrays")
    first = arrays[0]
    for other in arrays[1:]:
        if not first.data.shape == other.data.shape:
            raise ValueError("input arrays must have equal shapes")
        if not isinstance(first.data, type(other.data)):
            raise ValueError("input arrays must have same type")


def get_xy_range(raster, xdim=None, ydim=None):
    if ydim is None:
        ydim = raster.dims[-2]
    if xdim is None:
        xdim = raster.dims[-1]
    xmin = raster[xdim].min().item()
    xmax = raster[xdim].max().item()
    ymin = raster[ydim].min().item()
    ymax = raster[ydim].max().item()
    return (xmin, xmax), (ymin, ymax)


def calc_res(raster, xdim=None, ydim=None):
    """(xres, yres) from the coordinate extents (reference: utils.py:204-230)."""
    h, w = raster.shape[-2:]
    xrange, yrange = get_xy_range(raster, xdim, ydim)
    xres = (xrange[-1] - xrange[0]) / (w - 1)
    yres = (yrange[-1] - yrange[0]) / (h - 1)
    return xres, yres


def get_dataarray_resolution(agg, xdim=None, ydim=None):
    """attrs['res'] (pair or scalar) else calc_res (reference: utils.py:233-277)."""
    try:
        cellsize = agg.attrs.get("res")
        if (isinstance(cellsize, (tuple, np.ndarray, list)) and len(cellsize) == 2
                and isinstance(cellsize[0], (int, float)) and isinstance(cellsize[1], (int, float))):
            cellsize_x, cellsize_y = cellsize
        elif isinstance(cellsize, (int, float)):
            cellsize_x = cellsize
            cellsize_y = cellsize
        else:
            cellsize_x, cellsize_y = calc_res(agg, xdim, ydim)
    except Exception:
        cellsize_x, cellsize_y = calc_res(agg, xdim, ydim)
    return cellsize_x, cellsize_y
