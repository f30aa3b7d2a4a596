"""zonal.stats drop-in.  Reference: xrspatial/zonal.py:422-667 (`stats`), NumPy backend :280-332.

The reference sorts the whole raster twice (np.unique + np.argsort) and then loops over zones
in Python.  Here the host only maps zone ids to dense indices; ONE streaming pass on the
MI355X produces per-zone count / sum / sum-of-squares / min / max (the per-block partials of the
reference's own dask path, zonal.py:83-102), from which mean / std / var follow.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Union

import numpy as np
import pandas as pd

from . import _lib
from ._launch import get_stream
from ._xr import DataArray, Dataset
from .device import DeviceArray
from .utils import validate_arrays

_DEVICE_STATS = ('mean', 'max', 'min', 'sum', 'std', 'var', 'count')
_DEFAULT_STATS = _DEVICE_STATS + ('majority',)
_DENSE_RANGE_LIMIT = 1 << 26


def _dense_zone_index(zones: np.ndarray):
    """unique finite zone ids (ascending, zones dtype) and an int32 dense index per cell (-1: no zone).

    Same set the reference obtains with np.unique(zones[np.isfinite(zones)]) (zonal.py:290),
    but integral ids in a bounded range are mapped with O(N) table lookups instead of a sort.
    """
    flat = zones.ravel()
    if np.issubdtype(flat.dtype, np.floating):
        finite = np.isfinite(flat)
        vals = flat[finite]
    else:
        finite = None
        vals = flat
    idx = np.full(flat.shape, -1, dtype=np.int32)
    if vals.size == 0:
        return flat[:0].copy(), idx.reshape(zones.shape)
    lo, hi = vals.min(), vals.max()
    integral = np.issubdtype(vals.dtype, np.integer) or bool(np.all(vals == np.floor(vals)))
    if integral and float(hi) - float(lo) < _DENSE_RANGE_LIMIT:
        off = vals.astype(np.int64) - np.int64(lo)
        present = np.zeros(int(np.int64(hi) - np.int64(lo)) + 1, dtype=bool)
        present[off] = True
        lut = np.cumsum(present, dtype=np.int64).astype(np.int32) - 1
        uniq = (np.flatnonzero(present) + np.int64(lo)).astype(flat.dtype)
        dense = lut[off]
    else:
        uniq, inv = np.unique(vals, return_inverse=True)
        dense = inv.astype(np.int32)
    if finite is None:
        idx = dense.astype(np.int32, copy=False)
    else:
        idx[finite] = dense
    return uniq, idx.reshape(zones.shape)


_ZONE_DTYPE_CODE = {np.dtype(np.int32): 0, np.dtype(np.int64): 1, np.dtype(np.float32): 2, np.dtype(np.float64): 3}


def _dense_zone_index_device(zones_dev: DeviceArray):
    """Device-side counterpart of `_dense_zone_index` for zone rasters already in HBM: returns
    (unique ids as a host array of the zones dtype, int32 DeviceArray of dense indices), or None when
    the ids are not integral / span too wide a range (the caller then takes the host path)."""
    code = _ZONE_DTYPE_CODE.get(zones_dev.dtype)
    if code is None:
        return None
    stream = get_stream()
    n = zones_dev.size
    res = DeviceArray((4,), np.float64)
    _lib.call("xrs_zonal_scan", zones_dev.ptr, code, n, res.ptr, stream)
    raw = res.get(stream)
    zmin, zmax = raw[0], raw[1]
    n_finite = int(raw[2:3].view(np.uint64)[0])
    all_integral = int(raw[3:4].view(np.int32)[0])
    if n_finite == 0:
        return zones_dev.get()[:0].ravel(), DeviceArray.from_numpy(np.full(zones_dev.shape, -1, np.int32))
    if not all_integral or zmax - zmin >= _DENSE_RANGE_LIMIT:
        return None
    rng = int(zmax - zmin) + 1
    present = DeviceArray((rng,), np.uint8)
    _lib.call("xrs_zonal_presence", zones_dev.ptr, code, n, float(zmin), rng, present.ptr, stream)
    mask = present.get(stream).astype(bool)
    lut = (np.cumsum(mask, dtype=np.int64) - 1).astype(np.int32)
    uniq = (np.flatnonzero(mask).astype(np.float64) + zmin).astype(zones_dev.dtype)
    lut_dev = DeviceArray.from_numpy(lut)
    idx = DeviceArray(zones_dev.shape, np.int32)
    _lib.call("xrs_zonal_index", zones_dev.ptr, code, n, float(zmin), rng, lut_dev.ptr, idx.ptr, stream)
    _lib.call("xrs_stream_sync", stream)
    return uniq, idx


def _stage(zone_idx, values):
    zdev = zone_idx if isinstance(zone_idx, DeviceArray) else DeviceArray.from_numpy(
        np.ascontiguousarray(zone_idx, dtype=np.int32))
    if isinstance(values, DeviceArray):
        vdev = values if values.dtype in (np.float32, np.float64) else values.astype(np.float64)
    else:
        host = np.asarray(values)
        vdev = DeviceArray.from_numpy(host if host.dtype == np.float32 else host.astype(np.float64))
    return zdev, vdev


def zonal_partials(zone_idx, values, n_zones, nodata_values=None, comm=None):
    """Per-zone (count, sum, sumsq, min, max) NumPy arrays for dense `zone_idx` (device or host arrays).

    `comm`: optional multi-GPU communicator (xrspatial_amd.distributed.Comm); the partials are
    all-reduced over it so every rank returns the global result."""
    _lib.require_device()
    stream = get_stream()
    zdev, vdev = _stage(zone_idx, values)
    f64 = vdev.dtype == np.float64
    vt = np.float64 if f64 else np.float32
    cnt = DeviceArray((n_zones,), np.uint64)
    s1 = DeviceArray((n_zones,), np.float64)
    s2 = DeviceArray((n_zones,), np.float64)
    mn = DeviceArray((n_zones,), vt)
    mx = DeviceArray((n_zones,), vt)
    sfx = "_f64" if f64 else ""
    _lib.call("xrs_zonal_init" + sfx, cnt.ptr, s1.ptr, s2.ptr, mn.ptr, mx.ptr, n_zones, stream)
    has_nodata = nodata_values is not None
    nodata = float(nodata_values) if has_nodata else 0.0
    _lib.call("xrs_zonal_partials_f64" if f64 else "xrs_zonal_partials_f32", zdev.ptr, vdev.ptr, vdev.size,
              n_zones, nodata, int(has_nodata), cnt.ptr, s1.ptr, s2.ptr, mn.ptr, mx.ptr, stream)
    if comm is not None:
        _lib.call("xrs_zonal_allreduce", comm.handle, cnt.ptr, s1.ptr, s2.ptr, mn.ptr, mx.ptr, int(f64),
                  n_zones, stream)
    return cnt.get(stream), s1.get(stream), s2.get(stream), mn.get(stream), mx.get(stream)


def zonal_majority(zone_idx, values, n_zones, nodata_values=None):
    """Per-zone most frequent valid value (ties -> smallest), float64, NaN for empty zones."""
    _lib.require_device()
    stream = get_stream()
    zdev, vdev = _stage(zone_idx, values)
    f64 = vdev.dtype == np.float64
    out = DeviceArray((n_zones,), np.float64)
    nbytes = int(_lib.load().xrs_zonal_majority_workspace_bytes(vdev.size, n_zones, int(f64)))
    work = DeviceArray((nbytes,), np.uint8)
    has_nodata = nodata_values is not None
    nodata = float(nodata_values) if has_nodata else 0.0
    _lib.call("xrs_zonal_majority_f64" if f64 else "xrs_zonal_majority_f32", zdev.ptr, vdev.ptr, vdev.size,
              n_zones, nodata, int(has_nodata), work.ptr, nbytes, out.ptr, stream)
    return out.get(stream)


def finalize_stats(stat_names, count, s1, s2, mn, mx, majority=None):
    """Per-zone statistics from the partials (formulas of zonal.py:100-102); zones without a valid
    cell are NaN in every column, count included (zonal.py:153-161 pre-fills NaN)."""
    n = count.astype(np.float64)
    empty = count == 0
    with np.errstate(all="ignore"):
        mean = s1 / n
        var = (s2 - s1 * s1 / n) / n
        var = np.where(var < 0, 0.0, var)         # rounding guard; the exact value is >= 0
    table = {'mean': mean, 'max': mx.astype(np.float64), 'min': mn.astype(np.float64), 'sum': s1,
             'std': np.sqrt(var), 'var': var, 'count': n, 'majority': majority}
    out = {}
    for name in stat_names:
        col = np.array(table[name], dtype=np.float64)
        col[empty] = np.nan
        out[name] = col
    return out


def _stats_hip(zones_data, values_data, zone_ids, stat_names, nodata_values, return_type, comm=None):
    like_numpy = not isinstance(values_data, DeviceArray)
    mapped = None
    if isinstance(zones_data, DeviceArray):
        _lib.require_device()
        mapped = _dense_zone_index_device(zones_data)           # stays in HBM when ids are integral
    if mapped is None:
        zones_host = zones_data.get() if isinstance(zones_data, DeviceArray) else np.asarray(zones_data)
        unique_zones, idx = _dense_zone_index(zones_host)
        idx_dev = None
    else:
        unique_zones, idx_dev = mapped
        idx = idx_dev                                           # only .shape / .size are used below
    if zone_ids is None:
        selected = unique_zones
    else:
        wanted = np.unique(zone_ids)
        selected = [z for z in wanted if z in unique_zones]
    nz = len(unique_zones)
    if idx_dev is None:
        idx_dev = DeviceArray.from_numpy(idx)
    _, vdev = _stage(idx_dev, values_data)
    count, s1, s2, mn, mx = zonal_partials(idx_dev, vdev, nz, nodata_values, comm)
    majority = zonal_majority(idx_dev, vdev, nz, nodata_values) if 'majority' in stat_names else None
    cols = finalize_stats(stat_names, count, s1, s2, mn, mx, majority)
    keep = [i for i, z in enumerate(unique_zones) if z in selected]
    if return_type == 'pandas.DataFrame':
        frame = {'zone': selected}
        for name in stat_names:
            frame[name] = cols[name][keep]
        return pd.DataFrame(frame)
    # back-projection (zonal.py:313-332): every cell gets its zone's statistic, NaN outside selected zones
    table = np.full((len(stat_names), max(nz, 1)), np.nan)
    for i, name in enumerate(stat_names):
        table[i, keep] = cols[name][keep]
    tdev = DeviceArray.from_numpy(table)
    out = DeviceArray((len(stat_names),) + tuple(idx.shape), np.float64)
    _lib.call("xrs_zonal_backproject_f64", idx_dev.ptr, idx.size, tdev.ptr, len(stat_names), max(nz, 1), out.ptr,
              get_stream())
    return out.get(get_stream()) if like_numpy else out


def stats(
    zones,
    values,
    zone_ids: Optional[List[Union[int, float]]] = None,
    stats_funcs: Union[Dict, List] = [
        "mean",
        "max",
        "min",
        "sum",
        "std",
        "var",
        "count",
        "majority",
    ],
    nodata_values: Union[int, float] = None,
    return_type: str = 'pandas.DataFrame',
):
    """Summary statistics of `values` for every zone of `zones`.

    Same signature as `xrspatial.zonal.stats`.  All eight default statistics (mean / max / min / sum /
    std / var / count from one streaming partial-sum pass, majority from a device sort), `zone_ids`,
    `nodata_values`, Dataset `values` and both return types run on the MI355X.  Custom callables
    (`stats_funcs` as a dict) are arbitrary Python and raise NotImplementedError here."""
    if isinstance(values, Dataset):
        if return_type != 'pandas.DataFrame':
            raise ValueError("return_type must be 'pandas.DataFrame' when values is a Dataset")
        dfs = []
        for var_name in values.data_vars:
            df = stats(zones, values[var_name], zone_ids, stats_funcs, nodata_values, 'pandas.DataFrame')
            df = df.rename(columns={c: f'{var_name}_{c}' for c in df.columns if c != 'zone'})
            dfs.append(df)
        result = dfs[0]
        for df in dfs[1:]:
            result = result.merge(df, on='zone', how='outer')
        return result

    validate_arrays(zones, values)
    if not (issubclass(zones.data.dtype.type, np.integer) or issubclass(zones.data.dtype.type, np.floating)):
        raise ValueError("`zones` must be an array of integers or floats.")
    if not (issubclass(values.data.dtype.type, np.integer) or issubclass(values.data.dtype.type, np.floating)):
        raise ValueError("`values` must be an array of integers or floats.")
    if len(values.shape) != 2:
        raise ValueError("`values` must be 2D (pass a Dataset for several layers)")

    if isinstance(stats_funcs, dict):
        raise NotImplementedError(
            "custom stats callables cannot run on the MI355X backend; pass a list of names from "
            f"{list(_DEVICE_STATS)}")
    names = list(stats_funcs)
    for name in names:
        if name not in _DEFAULT_STATS:
            raise ValueError(f"Invalid stat name. {name} option not supported.")
    if return_type not in ('pandas.DataFrame', 'xarray.DataArray'):
        raise ValueError(f"unknown return_type {return_type!r}")
    if not isinstance(values.data, (np.ndarray, DeviceArray)):
        raise TypeError("Unsupported Array Type: {}".format(type(values)))
    result = _stats_hip(zones.data, values.data, zone_ids, names, nodata_values, return_type)
    if return_type == 'xarray.DataArray':
        coords = dict(values.coords.items())
        coords['stats'] = names
        return DataArray(result, coords=coords, dims=('stats',) + tuple(values.dims), attrs=values.attrs)
    return result
