"""Kernels and 2-D convolution.  Reference: xrspatial/convolution.py.

Host-side kernel builders (`circle_kernel`, `annulus_kernel`, `custom_kernel`,
`calc_cellsize`, :30-282) behave as upstream; `convolve_2d` (:389-397, raw arrays)
and `convolution_2d` (:400-521, DataArray wrapper) run on the MI355X.
"""
from __future__ import annotations

import re

import numpy as np

from . import _lib
from ._launch import finish, get_stream, pipeline_ok, pipelined_rows, plane_args, sharded_f32
from ._xr import DataArray
from .device import DeviceArray, to_device_f32
from .sharded import ShardedArray
from .utils import dask_overlap, get_dataarray_resolution, is_dask

DEFAULT_UNIT = 'meter'
_UNIT_IN_METERS = {
    'meter': 1, 'meters': 1, 'm': 1,
    'feet': 0.3048, 'foot': 0.3048, 'ft': 0.3048,
    'miles': 1609.344, 'mls': 1609.344, 'ml': 1609.344,
    'kilometer': 1000, 'kilometers': 1000, 'km': 1000,
}


def _get_distance(distance_str):
    """'3 km' / '250' / '2.5ft' -> metres (reference: convolution.py:42-75)."""
    parts = [x for x in re.split(r'(-?\d*\.?\d+)', distance_str) if x != '']
    if len(parts) not in (1, 2):
        raise ValueError("Invalid distance.")
    number = parts[0]
    unit = parts[1] if len(parts) == 2 else DEFAULT_UNIT
    try:
        distance = float(number)
    except ValueError:
        raise ValueError("Distance should be a positive numeric value.\n")
    if distance <= 0:
        raise ValueError("Distance should be a positive.\n")
    unit = unit.lower().replace(' ', '')
    if unit not in _UNIT_IN_METERS:
        raise ValueError(
            "Distance unit should be one of the following: \n"
            "meter (meter, meters, m),\n"
            "kilometer (kilometer, kilometers, km),\n"
            "foot (foot, feet, ft),\n"
            "mile (mile, miles, ml, mls)")
    return distance * _UNIT_IN_METERS[unit]


def calc_cellsize(raster):
    """(cellsize_x, |cellsize_y|) in metres from attrs['res'] / coords and attrs['unit'] (:78-134)."""
    unit = raster.attrs.get('unit', DEFAULT_UNIT)
    cellsize_x, cellsize_y = get_dataarray_resolution(raster)
    scale = _UNIT_IN_METERS[unit]
    return cellsize_x * scale, np.abs(cellsize_y * scale)


def _ellipse_kernel(half_w, half_h):
    xs = np.linspace(-half_w, half_w, 2 * half_w + 1)
    ys = np.linspace(-half_h, half_h, 2 * half_h + 1)[:, None]
    inside = (xs * half_h) ** 2 + (ys * half_w) ** 2 <= (half_w * half_h) ** 2   # division-free test (:144)
    return inside.astype(float)


def circle_kernel(cellsize_x, cellsize_y, radius):
    """0/1 float64 mask of the cells within `radius` (number or '<n> <unit>' string) (:149-196)."""
    r = _get_distance(str(radius))
    return _ellipse_kernel(int(r / cellsize_x), int(r / cellsize_y))


def annulus_kernel(cellsize_x, cellsize_y, outer_radius, inner_radius):
    """Ring mask: outer circle minus the centred inner circle (:199-259)."""
    outer = circle_kernel(cellsize_x, cellsize_y, outer_radius)
    inner = circle_kernel(cellsize_x, cellsize_y, inner_radius)
    pad = np.array(outer.shape) - np.array(inner.shape)
    inner = np.pad(inner, ((pad[0] // 2, pad[0] // 2), (pad[1] // 2, pad[1] // 2)),
                   mode='constant', constant_values=0)
    return outer - inner


def custom_kernel(kernel):
    """Validate a user kernel: ndarray with odd rows and columns (:262-282)."""
    if not isinstance(kernel, np.ndarray):
        raise ValueError(
            "Received a custom kernel that is not a Numpy array.",
            "The kernel received was of type {} and needs to be of type `ndarray`".format(type(kernel)))
    rows, cols = kernel.shape
    if rows % 2 == 0 or cols % 2 == 0:
        raise ValueError(
            "Received custom kernel with improper dimensions.",
            "A custom kernel needs to have an odd shape, the supplied kernel "
            "has {} rows and {} columns.".format(rows, cols))
    return kernel


def _kernel_f64(kernel):
    k = np.ascontiguousarray(np.asarray(kernel), dtype=np.float64)
    if k.ndim != 2:
        raise ValueError("kernel must be 2D")
    return k


def _convolve_2d_hip(data, kernel):
    # replaces _convolve_2d_numpy (convolution.py:285-313)
    _lib.require_device()
    like_numpy = not isinstance(data, DeviceArray)
    if len(data.shape) != 2:
        raise ValueError("expected a 2D raster")
    k = _kernel_f64(kernel)
    if pipeline_ok(data) and max(k.shape) // 2 < 128:
        # large numpy raster: row bands upload / compute / download concurrently
        cols = data.shape[1]
        work = DeviceArray((max(int(_lib.load().xrs_kxk_workspace_bytes(k.shape[0], k.shape[1])), 16),), np.uint8)

        def launch(in_ptr, out_ptrs, n_rows, ht, hb, stream):
            _lib.call("xrs_convolve2d_f32", in_ptr, out_ptrs[0], n_rows, cols, cols, cols, k.ctypes.data, k.shape[0],
                      k.shape[1], work.ptr, ht, hb, stream)

        return pipelined_rows(data, [np.float32], launch, k.shape[0] // 2)[0]
    src = to_device_f32(data)
    rows, cols, ld = plane_args(src)
    out = DeviceArray((rows, cols), np.float32)
    work = DeviceArray((max(int(_lib.load().xrs_kxk_workspace_bytes(k.shape[0], k.shape[1])), 16),), np.uint8)
    stream = get_stream()
    _lib.call("xrs_convolve2d_f32", src.ptr, out.ptr, rows, cols, ld, ld, k.ctypes.data, k.shape[0],
              k.shape[1], work.ptr, 0, 0, stream)
    _lib.call("xrs_stream_sync", stream)      # `k` and `work` must outlive the launch
    return finish(out, like_numpy)


def _convolve_2d_sharded(data, kernel):
    # the reference's dask path: map_overlap(depth=k//2, boundary=nan) (convolution.py:316-327)
    _lib.require_device()
    k = _kernel_f64(kernel)
    src = sharded_f32(data)
    stream = get_stream()
    ht, hb = src.halos(k.shape[0] // 2, stream)
    rows, cols = src.shape
    out = src.like(np.float32)
    work = DeviceArray((max(int(_lib.load().xrs_kxk_workspace_bytes(k.shape[0], k.shape[1])), 16),), np.uint8)
    _lib.call("xrs_convolve2d_f32", src.ptr, out.ptr, rows, cols, cols, cols, k.ctypes.data, k.shape[0],
              k.shape[1], work.ptr, ht, hb, stream)
    _lib.call("xrs_stream_sync", stream)      # `k` and `work` must outlive the launch
    return out


def convolve_2d(data, kernel):
    """Correlate a raw 2-D array with `kernel` (NaN border of k//2, NaNs propagate).  Raw arrays in/out."""
    if isinstance(data, (np.ndarray, DeviceArray)):
        return _convolve_2d_hip(data, kernel)
    if isinstance(data, ShardedArray):
        return _convolve_2d_sharded(data, kernel)
    if is_dask(data):                       # convolution.py:316-327: map_overlap(depth = k // 2, boundary = nan)
        k = _kernel_f64(kernel)
        return dask_overlap(_convolve_2d_hip, (k.shape[0] // 2, k.shape[1] // 2))(data, kernel)
    raise TypeError("Unsupported Array Type: {}".format(type(data)))


def convolution_2d(agg, kernel, name='convolution_2d'):
    """DataArray wrapper of `convolve_2d` (:400-521)."""
    out = convolve_2d(agg.data, kernel)
    return DataArray(out, name=name, coords=agg.coords, dims=agg.dims, attrs=agg.attrs)
