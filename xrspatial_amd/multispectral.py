"""Per-cell multispectral indices: ndvi, evi, savi, the other normalized-ratio indices that share
ndvi's kernel (nbr, nbr2, ndmi) and arvi, gci, sipi, ebbi.  Reference: xrspatial/multispectral.py.
`true_color` (the RGBA composite with a sigmoid contrast stretch) runs on the device too.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._launch import finish, get_stream, percell_pipelined, sharded_f32
from ._xr import DataArray
from .dataset_support import supports_dataset_bands
from .device import DTYPE_CODE, DeviceArray, to_device_f32
from .sharded import ShardedArray, same_layout
from .utils import ArrayTypeFunctionMapping, dask_blocks, validate_arrays


def _percell(fn_name, bands, extra):
    """bands: tuple of same-shape arrays -> float32 result of the same shape."""
    _lib.require_device()
    if isinstance(bands[0], ShardedArray):            # per-cell: every rank works on its own rows, nothing is exchanged
        same_layout(*bands)
        dev = [sharded_f32(b) for b in bands]
        out = dev[0].like(np.float32)
        _lib.call(fn_name, *[d.ptr for d in dev], out.ptr, out.size, *extra, get_stream())
        return out
    like_numpy = not isinstance(bands[0], DeviceArray)
    if like_numpy:                                    # large numpy rasters: overlapped upload / compute / download
        out = percell_pipelined(fn_name, [np.asarray(b) for b in bands], extra)
        if out is not None:
            return out
    dev = [to_device_f32(b) for b in bands]           # `.astype('f4')` of the reference wrappers
    out = DeviceArray(dev[0].shape, np.float32)
    _lib.call(fn_name, *[d.ptr for d in dev], out.ptr, out.size, *extra, get_stream())
    return finish(out, like_numpy)


def _normalized_ratio(arr1, arr2):
    # replaces _normalized_ratio_cpu (multispectral.py:825-841)
    return _percell("xrs_normalized_ratio_f32", (arr1, arr2), ())


def _wrap(out, name, like):
    return DataArray(out, name=name, coords=like.coords, dims=like.dims, attrs=like.attrs)


def _nr_index(band1, band2, name):
    validate_arrays(band1, band2)
    mapper = ArrayTypeFunctionMapping(numpy_func=_normalized_ratio, hip_func=_normalized_ratio, sharded_func=_normalized_ratio,
                                      dask_func=dask_blocks(_normalized_ratio))
    return _wrap(mapper(band1)(band1.data, band2.data), name, band1)


@supports_dataset_bands(nir='nir_agg', red='red_agg')
def ndvi(nir_agg, red_agg, name='ndvi'):
    """Normalized Difference Vegetation Index (nir - red) / (nir + red); NaN where nir + red == 0.
    Same signature and float32 results as `xrspatial.multispectral.ndvi` (:653-733)."""
    return _nr_index(nir_agg, red_agg, name)


@supports_dataset_bands(nir='nir_agg', swir2='swir2_agg')
def nbr(nir_agg, swir2_agg, name='nbr'):
    """Normalized Burn Ratio (nir - swir2) / (nir + swir2)  (multispectral.py:476-560)."""
    return _nr_index(nir_agg, swir2_agg, name)


@supports_dataset_bands(swir1='swir1_agg', swir2='swir2_agg')
def nbr2(swir1_agg, swir2_agg, name='nbr2'):
    """Normalized Burn Ratio 2 (swir1 - swir2) / (swir1 + swir2)  (multispectral.py:563-650)."""
    return _nr_index(swir1_agg, swir2_agg, name)


@supports_dataset_bands(nir='nir_agg', swir1='swir1_agg')
def ndmi(nir_agg, swir1_agg, name='ndmi'):
    """Normalized Difference Moisture Index (nir - swir1) / (nir + swir1)  (multispectral.py:737-823)."""
    return _nr_index(nir_agg, swir1_agg, name)


@supports_dataset_bands(nir='nir_agg', red='red_agg', blue='blue_agg')
def evi(nir_agg, red_agg, blue_agg, c1=6.0, c2=7.5, soil_factor=1.0, gain=2.5, name='evi'):
    """Enhanced Vegetation Index gain * (nir - red) / (nir + c1*red - c2*blue + soil_factor).
    Same signature, validation and float32 results as `xrspatial.multispectral.evi` (:226-346)."""
    if not red_agg.shape == nir_agg.shape == blue_agg.shape:
        raise ValueError("input layers expected to have equal shapes")
    if not isinstance(c1, (float, int)):
        raise ValueError("c1 must be numeric")
    if not isinstance(c2, (float, int)):
        raise ValueError("c2 must be numeric")
    if soil_factor > 1.0 or soil_factor < -1.0:
        raise ValueError("soil factor must be between [-1.0, 1.0]")
    if gain < 0:
        raise ValueError("gain must be greater than 0")
    validate_arrays(nir_agg, red_agg, blue_agg)

    def run(nir, red, blue):   # replaces _evi_cpu (multispectral.py:175-188)
        return _percell("xrs_evi_f32", (nir, red, blue),
                        (float(c1), float(c2), float(soil_factor), float(gain)))

    mapper = ArrayTypeFunctionMapping(numpy_func=run, hip_func=run, sharded_func=run, dask_func=dask_blocks(run))
    return _wrap(mapper(red_agg)(nir_agg.data, red_agg.data, blue_agg.data), name, nir_agg)


@supports_dataset_bands(nir='nir_agg', red='red_agg')
def savi(nir_agg, red_agg, soil_factor=1.0, name='savi'):
    """Soil Adjusted Vegetation Index (nir - red) / ((nir + red + L) * (1 + L)).
    Same signature, validation and float32 results as `xrspatial.multispectral.savi` (:927-1013)."""
    validate_arrays(red_agg, nir_agg)
    if not -1.0 <= soil_factor <= 1.0:
        raise ValueError("soil factor must be between [-1.0, 1.0]")

    def run(nir, red):         # replaces _savi_cpu (multispectral.py:876-890)
        return _percell("xrs_savi_f32", (nir, red), (float(soil_factor),))

    mapper = ArrayTypeFunctionMapping(numpy_func=run, hip_func=run, sharded_func=run, dask_func=dask_blocks(run))
    return _wrap(mapper(red_agg)(nir_agg.data, red_agg.data), name, nir_agg)


def _simple_index(fn_name, like, bands, name, order=None):
    """validate, run a parameter-free per-cell kernel on `bands`, wrap with `like`'s metadata."""
    validate_arrays(*(order or bands))

    def run(*arrays):
        return _percell(fn_name, arrays, ())

    mapper = ArrayTypeFunctionMapping(numpy_func=run, hip_func=run, sharded_func=run, dask_func=dask_blocks(run))
    return _wrap(mapper(like)(*[b.data for b in bands]), name, like)


@supports_dataset_bands(nir='nir_agg', red='red_agg', blue='blue_agg')
def arvi(nir_agg, red_agg, blue_agg, name='arvi'):
    """Atmospherically Resistant Vegetation Index (nir - 2*red + blue) / (nir + 2*red + blue).
    Same signature and float32 results as `xrspatial.multispectral.arvi` (:79-171)."""
    return _simple_index("xrs_arvi_f32", nir_agg, (nir_agg, red_agg, blue_agg), name,
                         order=(red_agg, nir_agg, blue_agg))


@supports_dataset_bands(nir='nir_agg', green='green_agg')
def gci(nir_agg, green_agg, name='gci'):
    """Green Chlorophyll Index nir / green - 1; NaN where green == 0  (multispectral.py:392-472)."""
    return _simple_index("xrs_gci_f32", nir_agg, (nir_agg, green_agg), name)


@supports_dataset_bands(nir='nir_agg', red='red_agg', blue='blue_agg')
def sipi(nir_agg, red_agg, blue_agg, name='sipi'):
    """Structure Insensitive Pigment Index (nir - blue) / (nir - red)  (multispectral.py:1066-1156)."""
    return _simple_index("xrs_sipi_f32", nir_agg, (nir_agg, red_agg, blue_agg), name,
                         order=(red_agg, nir_agg, blue_agg))


@supports_dataset_bands(red='red_agg', swir='swir_agg', tir='tir_agg')
def ebbi(red_agg, swir_agg, tir_agg, name='ebbi'):
    """Enhanced Built-Up and Bareness Index (swir - red) / (10 * sqrt(swir + tir))  (multispectral.py:1209-1332)."""
    return _simple_index("xrs_ebbi_f32", red_agg, (red_agg, swir_agg, tir_agg), name)


def _true_color_hip(r, g, b, nodata, c, th):
    # replaces _true_color_numpy / _normalize_data_cpu (multispectral.py:1334-1361, 1387-1399): three min / max
    # reductions, then one pass that stretches the bands and packs RGBA bytes
    _lib.require_device()
    like_numpy = not isinstance(r, DeviceArray)
    stream = get_stream()
    raw = r if isinstance(r, DeviceArray) else np.ascontiguousarray(r)
    if raw.dtype not in DTYPE_CODE:                       # bool, float16 ...: compared as float64
        raw = np.asarray(raw.get() if isinstance(raw, DeviceArray) else raw).astype(np.float64)
    raw_dev = raw if isinstance(raw, DeviceArray) else DeviceArray.from_numpy(raw)
    red = raw_dev.astype(np.float32)                      # (device-side cast; a float32 band is used as is)
    bands = [red, to_device_f32(g), to_device_f32(b)]
    minmax = DeviceArray((6,), np.float32)
    for i, band in enumerate(bands):
        _lib.call("xrs_nan_minmax_f32", band.ptr, band.size, minmax.ptr + 8 * i, stream)
    out = DeviceArray(tuple(red.shape) + (4,), np.uint8)
    _lib.call("xrs_true_color_u8", bands[0].ptr, bands[1].ptr, bands[2].ptr, raw_dev.ptr, DTYPE_CODE[raw_dev.dtype],
              red.size, minmax.ptr, float(nodata), float(c), float(th), out.ptr, stream)
    _lib.call("xrs_stream_sync", stream)                  # the float32 casts above are temporaries
    return finish(out, like_numpy)


def true_color(r, g, b, nodata=1, c=10.0, th=0.125, name='true_color'):
    """RGBA composite (uint8, dims [y, x, band]) of three bands: each band is stretched to its own min .. max, passed
    through the sigmoid `1 / (1 + exp(c * (th - v)))` and scaled to 0 .. 255; alpha is 0 where the red band is NaN or
    <= `nodata`.  Same signature and results as `xrspatial.multispectral.true_color` (:1419-1495); device-resident bands
    are supported as well (the reference has no GPU path for it)."""
    validate_arrays(r, g, b)
    if len(r.shape) != 2:
        raise ValueError("true_color() takes 2D bands")
    mapper = ArrayTypeFunctionMapping(numpy_func=_true_color_hip, hip_func=_true_color_hip)
    out = mapper(r)(r.data, g.data, b.data, nodata, c, th)
    y_dim, x_dim = r.dims
    coords = {'band': [0, 1, 2, 3]}
    for k in (y_dim, x_dim):
        if k in r.coords:
            coords[k] = r.coords[k]
    return DataArray(out, name=name, dims=[y_dim, x_dim, 'band'], coords=coords, attrs=r.attrs)
