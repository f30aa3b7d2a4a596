"""zonal.stats drop-in.  Reference: xrspatial/zonal.py:422-667 (`stats`), NumPy backend :280-332.

The reference sorts the whole raster twice (np.unique + np.argsort) and then loops over zones
in Python.  Here the host only maps zone ids to dense indices; ONE streaming pass on the
MI355X produces per-zone count / sum / sum-of-squares / min / max (the per-block partials of the
reference's own dask path, zonal.py:83-102), from which mean / std / var follow.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Union

import numpy as np
import pandas as pd

from . import _lib
from ._launch import get_stream
from ._xr import DataArray, Dataset
from .device import DTYPE_CODE, DeviceArray
from .sharded import ShardedArray, ShardedStack, same_layout
from .utils import is_dask, validate_arrays

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


_OPTIMISTIC_WINDOW = 1 << 20         # bytes of the presence map filled during the scan of int32 zone ids

_ZONE_DTYPE_CODE = {np.dtype(np.int32): 0, np.dtype(np.int64): 1, np.dtype(np.float32): 2, np.dtype(np.float64): 3}


def _zone_table_device(zones_dev: DeviceArray):
    """(unique ids, zmin, range, int32 DeviceArray table id - zmin -> dense index) for an integral zone raster in
    HBM, without touching the raster on the host; None if the ids are not integral / too spread out / all invalid."""
    code = _ZONE_DTYPE_CODE.get(zones_dev.dtype)
    if code is None:
        return None
    stream = get_stream()
    n = zones_dev.size
    res = DeviceArray((4,), np.float64)
    window_map = None
    if code == 0:
        # int32 ids: the scan marks ids in [0, 2^20) on the way, which usually makes the presence pass unnecessary
        window_map = DeviceArray((_OPTIMISTIC_WINDOW,), np.uint8)
        _lib.call("xrs_zonal_scan_presence_i32", zones_dev.ptr, n, res.ptr, window_map.ptr, _OPTIMISTIC_WINDOW, stream)
    else:
        _lib.call("xrs_zonal_scan", zones_dev.ptr, code, n, res.ptr, stream)
    raw = res.get(stream)
    zmin, zmax = raw[0], raw[1]
    n_finite = int(raw[2:3].view(np.uint64)[0])
    all_integral = int(raw[3:4].view(np.int32)[0])
    if n_finite == 0 or not all_integral or zmax - zmin >= _DENSE_RANGE_LIMIT:
        return None
    rng = int(zmax - zmin) + 1
    if window_map is not None and zmin >= 0 and zmax < _OPTIMISTIC_WINDOW:
        mask = window_map.get(stream)[int(zmin):int(zmax) + 1].astype(bool)
    else:
        present = DeviceArray((rng,), np.uint8)
        _lib.call("xrs_zonal_presence", zones_dev.ptr, code, n, float(zmin), rng, present.ptr, stream)
        mask = present.get(stream).astype(bool)
    lut = np.where(mask, np.cumsum(mask, dtype=np.int64) - 1, -1).astype(np.int32)
    uniq = (np.flatnonzero(mask).astype(np.float64) + zmin).astype(zones_dev.dtype)
    return uniq, zmin, rng, DeviceArray.from_numpy(lut)


_ONE_PASS_SAMPLES = 1 << 16          # cells of the strided sample that picks the id window and the shift
_ONE_PASS_WINDOW_MAX = 4096          # ids per guessed window (29 B of LDS each)


def _one_pass_partials(zones_dev: DeviceArray, vdev: DeviceArray, nodata_values):
    """zonal partials of an int32 zone raster WITHOUT a discovery pass over it: a strided sample of the rasters (one tiny
    launch, 24 bytes back) gives the range of ids to expect and the shift; the window [base, base + W) guessed around that
    range is then reduced in ONE pass (xrs_zonal_partials_window_*: tables indexed by id - base in LDS), which also notes
    ids that occur with invalid values only and whether any cell fell outside the window.  One more read of the tables
    brings everything back.  Returns (unique ids, count, sum, sumsq, min, max, shift) with one entry per id that occurs --
    the set np.unique(zones) of zonal.py:290 -- or None when the guess did not hold (ids wider spread than a window, or a
    cell outside it): the caller then runs discovery and reduction as two passes."""
    stream = get_stream()
    n = int(zones_dev.size)
    f64 = vdev.dtype == np.float64
    sfx, vt = ("f64", np.float64) if f64 else ("f32", np.float32)
    has_nodata = nodata_values is not None
    nodata = float(nodata_values) if has_nodata else 0.0
    res = DeviceArray((3,), np.float64)
    _lib.call("xrs_zonal_sample_" + sfx, zones_dev.ptr, vdev.ptr, n, min(n, _ONE_PASS_SAMPLES), nodata, int(has_nodata), res.ptr,
              stream)
    raw = res.get(stream)
    zmin, zmax = (int(v) for v in raw[:1].view(np.int32)[:2])
    n_valid = int(raw[2:3].view(np.uint64)[0])
    shift = float(np.rint(raw[1] / n_valid)) if n_valid else 0.0                    # (an integer, like _pick_shift's)
    rng = zmax - zmin + 1
    window = 256
    while window < 2 * rng:
        window *= 2
    window = min(window, _ONE_PASS_WINDOW_MAX)
    if rng + rng // 4 > window:
        return None
    base = max(zmin - (window - rng) // 2, -(1 << 31))
    if base + window > (1 << 31):
        base = (1 << 31) - window
    # one buffer for everything that comes back: count u64 | sum f64 | sumsq f64 | min | max | present u8 | overflow i32
    vsz = np.dtype(vt).itemsize
    off_s1, off_s2 = 8 * window, 16 * window
    off_mn = 24 * window
    off_mx = off_mn + vsz * window
    off_pr = off_mx + vsz * window
    off_ov = (off_pr + window + 7) & ~7
    buf = DeviceArray((off_ov + 8,), np.uint8)
    p = buf.ptr
    _lib.call("xrs_zonal_partials_window_" + sfx, zones_dev.ptr, base, window, vdev.ptr, n, nodata, int(has_nodata), shift,
              p, p + off_s1, p + off_s2, p + off_mn, p + off_mx, p + off_pr, p + off_ov, stream)
    host = buf.get(stream)
    if int(host[off_ov:off_ov + 4].view(np.int32)[0]):
        return None
    count = host[:off_s1].view(np.uint64)
    seen = (count > 0) | (host[off_pr:off_pr + window] > 0)
    keep = np.flatnonzero(seen)
    ids = (keep + base).astype(np.int32)
    return (ids, count[keep], host[off_s1:off_s2].view(np.float64)[keep], host[off_s2:off_mn].view(np.float64)[keep],
            host[off_mn:off_mx].view(vt)[keep], host[off_mx:off_pr].view(vt)[keep], shift)


def _dense_zone_index_device(zones_dev: DeviceArray, max_range=None):
    """Device-side counterpart of `_dense_zone_index` for zone rasters already in HBM: returns
    (unique ids as a host array of the zones dtype, int32 DeviceArray of dense indices), or None when
    the ids are not integral / span too wide a range (the caller then takes the host path).
    `max_range`: give up right after the (min, max) scan -- before the presence and index passes over the raster -- when
    the ids span more than that many values."""
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
    if max_range is not None and rng > max_range:
        return None
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
        vdev = DeviceArray.from_numpy(host if host.dtype == np.float32 else host.astype(np.float64, copy=False))
    return zdev, vdev


def _pick_shift(vdev, comm, stream, nodata_values=None):
    """A value near the data for the shifted moments (xrs_zonal_partials_*: sums of x - shift and (x - shift)^2): the
    mean of a few hundred valid cells (finite, not nodata) sampled from three places of the plane, ROUNDED TO AN INTEGER
    -- x - shift and the sums then stay exact for integral rasters (class maps, integer DEMs), as they were before the
    shift existed.  Every rank of a sharded run must use the same one, so they take the smallest candidate; a rank
    without cells still takes part in that collective (with +inf).  0 when nothing valid was sampled anywhere (any
    value is correct)."""
    n = int(vdev.size)
    cand = np.inf
    if n:
        take = min(256, n)
        itemsize = vdev.dtype.itemsize
        sample = np.empty(3 * take, vdev.dtype)
        for k, start in enumerate((0, max(0, n // 2 - take // 2), max(0, (3 * n) // 4 - take // 2))):
            start = min(start, n - take)
            _lib.call("xrs_memcpy_d2h", sample[k * take:(k + 1) * take].ctypes.data, vdev.ptr + start * itemsize, take * itemsize, stream)
        _lib.call("xrs_stream_sync", stream)
        good = sample[np.isfinite(sample)]
        if nodata_values is not None:
            good = good[good != nodata_values]
        if good.size:
            cand = float(np.rint(np.mean(good, dtype=np.float64)))
    if comm is not None:
        cand = float(np.asarray(comm.allreduce(np.array([cand]), 'min')).reshape(-1)[0])
    return cand if np.isfinite(cand) else 0.0


def zonal_partials(zone_idx, values, n_zones, nodata_values=None, comm=None, table=None, shift=None):
    """Per-zone (count, sum, sumsq, min, max, shift) for dense `zone_idx` (device or host arrays): NumPy arrays, with
    sum / sumsq the sums of (x - shift) and (x - shift)^2 (finalize_stats adds the shift back).

    `comm`: optional multi-GPU communicator (xrspatial_amd.distributed.Comm); the partials are
    all-reduced over it so every rank returns the global result.
    `table`: (zmin, range, lut DeviceArray) -- `zone_idx` then holds RAW int32 zone ids that the kernel maps through
    the table itself (no dense index raster is materialised).
    `shift`: the value the moments are taken about; callers that gather per-rank partials themselves (comm=None on every
    rank) must pass the SAME one everywhere or hand the shifts to distributed.combine_zonal_partials."""
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
    if shift is None:
        shift = _pick_shift(vdev, comm, stream, nodata_values)       # (a collective when comm is given: every rank, also an empty one)
    shift = float(shift)
    if table is not None:
        zmin, rng, lut_dev = table
        _lib.call("xrs_zonal_partials_lut_f64" if f64 else "xrs_zonal_partials_lut_f32", zdev.ptr, int(zmin), int(rng),
                  lut_dev.ptr, vdev.ptr, vdev.size, n_zones, nodata, int(has_nodata), shift, cnt.ptr, s1.ptr, s2.ptr, mn.ptr,
                  mx.ptr, stream)
    else:
        _lib.call("xrs_zonal_partials_f64" if f64 else "xrs_zonal_partials_f32", zdev.ptr, vdev.ptr, vdev.size,
                  n_zones, nodata, int(has_nodata), shift, cnt.ptr, s1.ptr, s2.ptr, mn.ptr, mx.ptr, stream)
    if comm is not None:                 # distributed.Comm (RCCL) or any transport with the same surface
        return tuple(comm.allreduce_zonal(cnt, s1, s2, mn, mx, f64, n_zones, stream)) + (shift,)
    return cnt.get(stream), s1.get(stream), s2.get(stream), mn.get(stream), mx.get(stream), shift


# (zones x distinct values) counters of the counting path: the regime in which xrs_crosstab_counts keeps the whole table
# in per-workgroup LDS counters (zonal_index.hip): there counting takes 4.7 - 5.5 ms against the 19 ms of the two radix
# sorts (16384^2, 1000 zones) whatever the data.  Above it the kernel issues one 64-bit device atomic per cell, and
# which path wins depends on the DATA (tools/majority_table_probe.py, profiles/r04/majority_table_probe.log): classes
# drawn uniformly, 64 000 .. 4 M counters, 12.9 - 17.3 ms -- faster than sorting -- but a raster on which 90 % of a zone's
# cells carry one class (land cover) piles its atomics onto 1000 addresses: 29 - 60 ms.  The sort's 19 ms do not depend on
# the data, so larger tables keep the sorting path.
_MAJORITY_TABLE_LIMIT = 36864


def _majority_by_counting(zdev, vdev, n_zones, nodata_values, stream):
    """`majority` for CATEGORICAL values (integral, bounded range -- land-cover classes, integer DEMs): the values get
    dense category indices through the same scan / presence / index kernels as zone ids, (zone, category) pairs are
    counted by the crosstab kernel, and the arg-max of each zone's row (first maximum = smallest value, as
    _stats_majority's np.unique + argmax, zonal.py:56-68) is the answer: four streaming passes instead of two radix
    sorts of the whole raster.  None when the values are not categorical (the caller sorts)."""
    if n_zones == 0:
        return np.full(0, np.nan)
    # (rejected on the value RANGE right after the scan, before the presence and index passes over the raster)
    cat = _dense_zone_index_device(vdev, max_range=_MAJORITY_TABLE_LIMIT // n_zones)
    if cat is None:
        return None
    cats, cidx = cat
    nc = len(cats)
    out = np.full(n_zones, np.nan)
    if nc == 0 or n_zones == 0:
        return out
    if n_zones * nc > _MAJORITY_TABLE_LIMIT:
        return None
    cdev = DeviceArray((n_zones * nc,), np.uint64)
    _lib.call("xrs_memset", cdev.ptr, 0, cdev.nbytes, stream)
    _lib.call("xrs_crosstab_counts", zdev.ptr, cidx.ptr, zdev.size, n_zones, nc, cdev.ptr, stream)
    counts = cdev.get(stream).reshape(n_zones, nc)
    if nodata_values is not None:
        counts[:, np.asarray(cats, dtype=np.float64) == float(nodata_values)] = 0
    best = counts.argmax(axis=1)
    has = counts[np.arange(n_zones), best] > 0
    out[has] = np.asarray(cats, dtype=np.float64)[best[has]]
    return out


def zonal_majority(zone_idx, values, n_zones, nodata_values=None, counts=None):
    """Per-zone most frequent valid value (ties -> smallest), float64, NaN for empty zones.  `counts`: the valid cells per
    zone if the caller has them (the `count` of zonal_partials for the same rasters), which saves the counting pass."""
    _lib.require_device()
    stream = get_stream()
    zdev, vdev = _stage(zone_idx, values)
    how = os.environ.get("XRS_ZONAL_MAJORITY", "")                  # A/B and tests: "sort" / "hash" force one path
    f64 = vdev.dtype == np.float64
    has_nodata = nodata_values is not None
    nodata = float(nodata_values) if has_nodata else 0.0
    if how not in ("sort", "hash"):
        counted = _majority_by_counting(zdev, vdev, n_zones, nodata_values, stream)
        if counted is not None:
            return counted
    if how != "sort" and 0 < n_zones <= int(_lib.load().xrs_zonal_mode_max_zones()):
        # continuous values: route the cells by (zone, hash of the value) into LDS-sized parts and count there
        # (csrc/zonal_mode.hip); the entry behind the zones counts parts that did not fit -- then, and only then, sort
        out = DeviceArray((n_zones + 1,), np.float64)
        nbytes = int(_lib.load().xrs_zonal_mode_workspace_bytes(vdev.size, n_zones, int(f64)))
        work = DeviceArray((nbytes,), np.uint8)
        cdev = None
        if counts is not None and len(counts) == n_zones and int(np.max(counts, initial=0)) < (1 << 32):
            cdev = DeviceArray.from_numpy(np.ascontiguousarray(counts, dtype=np.uint32))
        _lib.call("xrs_zonal_mode_f64" if f64 else "xrs_zonal_mode_f32", zdev.ptr, vdev.ptr, vdev.size, n_zones, nodata,
                  int(has_nodata), cdev.ptr if cdev is not None else None, work.ptr, nbytes, out.ptr, stream)
        res = out.get(stream)
        del work
        if res[n_zones] == 0:
            return res[:n_zones]
    out = DeviceArray((n_zones,), np.float64)
    nbytes = int(_lib.load().xrs_zonal_majority_workspace_bytes(vdev.size, n_zones, int(f64)))
    work = DeviceArray((nbytes,), np.uint8)
    _lib.call("xrs_zonal_majority_f64" if f64 else "xrs_zonal_majority_f32", zdev.ptr, vdev.ptr, vdev.size,
              n_zones, nodata, int(has_nodata), work.ptr, nbytes, out.ptr, stream)
    return out.get(stream)


def zonal_grouped_values(zone_idx, values, n_zones, nodata_values=None):
    """The valid cells of every zone as one host array ordered by (zone index, value ascending): the device-side
    replacement of _sort_and_stride (zonal.py:121-141) for statistics that are host callables.  Only the valid cells
    cross PCIe; zone z's values are out[offsets[z]:offsets[z + 1]] with offsets = cumsum of the valid-cell counts."""
    _lib.require_device()
    stream = get_stream()
    zdev, vdev = _stage(zone_idx, values)
    f64 = vdev.dtype == np.float64
    n = int(vdev.size)
    nbytes = int(_lib.load().xrs_zonal_majority_workspace_bytes(n, max(n_zones, 1), int(f64)))
    work = DeviceArray((nbytes,), np.uint8)
    out = DeviceArray((max(n, 1),), vdev.dtype)
    has_nodata = nodata_values is not None
    nodata = float(nodata_values) if has_nodata else 0.0
    _lib.call("xrs_zonal_group_f64" if f64 else "xrs_zonal_group_f32", zdev.ptr, vdev.ptr, n, n_zones, nodata,
              int(has_nodata), work.ptr, nbytes, out.ptr, stream)
    return out, vdev.dtype


def _custom_columns(funcs, idx_dev, vdev, nz, count, keep, nodata_values, values_dtype):
    """Per-zone results of user callables: every callable gets the 1-D array of the zone's valid values (finite and
    != nodata_values, as _calc_stats zonal.py:157-161 filters them), in ascending order, in the dtype of the caller's
    raster; zones without a valid cell stay NaN and the callable is not called for them."""
    sorted_dev, _ = zonal_grouped_values(idx_dev, vdev, nz, nodata_values)
    offsets = np.concatenate([[0], np.cumsum(count.astype(np.int64))])
    n_valid = int(offsets[-1])
    host = np.empty(n_valid, sorted_dev.dtype)
    if n_valid:
        stream = get_stream()
        _lib.call("xrs_memcpy_d2h", host.ctypes.data, sorted_dev.ptr, n_valid * host.dtype.itemsize, stream)
        _lib.call("xrs_stream_sync", stream)
    if np.issubdtype(values_dtype, np.integer):
        host = host.astype(values_dtype)
    cols = {}
    for name, func in funcs.items():
        col = np.full(nz, np.nan)
        for i in keep:
            if offsets[i + 1] > offsets[i]:
                col[i] = func(host[offsets[i]:offsets[i + 1]])
        cols[name] = col
    return cols


def finalize_stats(stat_names, count, s1, s2, mn, mx, majority=None, shift=0.0):
    """Per-zone statistics from the partials (formulas of zonal.py:100-102, on moments of x - shift: the one-pass
    variance sum(d^2) - sum(d)^2 / n cancels in proportion to (mean - shift)^2 / var, so a shift near the data keeps it
    well conditioned for rasters with a large offset and a small spread); zones without a valid cell are NaN in every
    column, count included (zonal.py:153-161 pre-fills NaN)."""
    n = count.astype(np.float64)
    empty = count == 0
    with np.errstate(all="ignore"):
        mean = shift + s1 / n
        var = (s2 - s1 * s1 / n) / n
        var = np.where(var < 0, 0.0, var)         # rounding guard; the exact value is >= 0
    table = {'mean': mean, 'max': mx.astype(np.float64), 'min': mn.astype(np.float64), 'sum': s1 + n * shift,
             'std': np.sqrt(var), 'var': var, 'count': n, 'majority': majority}
    out = {}
    for name in stat_names:
        col = np.array(table[name], dtype=np.float64)
        col[empty] = np.nan
        out[name] = col
    return out


def _stats_hip(zones_data, values_data, zone_ids, stat_names, nodata_values, return_type, comm=None, custom=None):
    """`stat_names`: the output columns in order; `custom`: {name: callable} for the names that are not built in."""
    like_numpy = not isinstance(values_data, DeviceArray)
    custom = custom or {}
    builtin = [n for n in stat_names if n not in custom]
    mapped = None
    small_int = zones_data.dtype in (np.int32, np.int16, np.int8, np.uint16, np.uint8)
    if (small_int and return_type == 'pandas.DataFrame' and 'majority' not in stat_names and not custom and int(zones_data.size) > 0
            and (isinstance(zones_data, np.ndarray) or zones_data.dtype == np.int32)):
        # integer zones (<= 32 bit), partial-sum statistics only: the raw ids go to HBM as they are and the reduction
        # kernel maps them through a small table itself -- no pass over the raster on the host and no dense index
        # raster (4 B written + 4 B read per cell) on the device
        _lib.require_device()
        if isinstance(zones_data, np.ndarray):
            zones_data = DeviceArray.from_numpy(np.ascontiguousarray(zones_data, dtype=np.int32))
        _, vdev = _stage(zones_data, values_data)
        one = _one_pass_partials(zones_data, vdev, nodata_values) if comm is None else None
        if one is not None:
            # the ids were found by the reduction itself (no discovery pass: 8 B per cell in all)
            unique_zones, count, s1, s2, mn, mx, shift = one
            cols = finalize_stats(stat_names, count, s1, s2, mn, mx, None, shift)
            keep = slice(None) if zone_ids is None else np.flatnonzero(np.isin(unique_zones, np.unique(zone_ids)))
            frame = {'zone': unique_zones[keep]}
            for name in stat_names:
                frame[name] = cols[name][keep]
            return pd.DataFrame(frame)
        tab = _zone_table_device(zones_data)
        if tab is not None:
            unique_zones, zmin, rng, lut_dev = tab
            nz = len(unique_zones)
            # (the values staged for the one-pass attempt above: not uploaded a second time)
            count, s1, s2, mn, mx, shift = zonal_partials(zones_data, vdev, nz, nodata_values, comm, table=(zmin, rng, lut_dev))
            cols = finalize_stats(stat_names, count, s1, s2, mn, mx, None, shift)
            if zone_ids is None:
                keep = np.arange(nz)
            else:
                keep = np.flatnonzero(np.isin(unique_zones, np.unique(zone_ids)))
            frame = {'zone': unique_zones[keep]}
            for name in stat_names:
                frame[name] = cols[name][keep]
            return pd.DataFrame(frame)
    if isinstance(zones_data, np.ndarray) and zones_data.dtype in _ZONE_DTYPE_CODE and zones_data.size:
        # numpy zones: one upload, then the ids are mapped on the device (a host np.unique over the raster costs more
        # than the whole reduction); non-integral / widely spread ids fall back to the host below
        _lib.require_device()
        zdev = DeviceArray.from_numpy(np.ascontiguousarray(zones_data))
        mapped = _dense_zone_index_device(zdev)
        del zdev
    if isinstance(zones_data, DeviceArray):
        _lib.require_device()
        mapped = _dense_zone_index_device(zones_data)           # stays in HBM when ids are integral
    if mapped is None:
        zones_host = zones_data.get() if isinstance(zones_data, DeviceArray) else np.asarray(zones_data)
        unique_zones, idx_host = _dense_zone_index(zones_host)
        idx_dev = DeviceArray.from_numpy(idx_host)
    else:
        unique_zones, idx_dev = mapped
    if zone_ids is None:
        selected = unique_zones
    else:
        wanted = np.unique(zone_ids)
        selected = [z for z in wanted if z in unique_zones]
    nz = len(unique_zones)
    _, vdev = _stage(idx_dev, values_data)
    count, s1, s2, mn, mx, shift = zonal_partials(idx_dev, vdev, nz, nodata_values, comm)
    # (the counts of THIS device's cells: with a communicator they have been added over the ranks and are not passed on)
    majority = (zonal_majority(idx_dev, vdev, nz, nodata_values, counts=count if comm is None else None)
                if 'majority' in builtin else None)
    cols = finalize_stats(builtin, count, s1, s2, mn, mx, majority, shift)
    keep = [i for i, z in enumerate(unique_zones) if z in selected]
    if custom:
        cols.update(_custom_columns(custom, idx_dev, vdev, nz, count, keep, nodata_values, np.dtype(values_data.dtype)))
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
    out = DeviceArray((len(stat_names),) + tuple(idx_dev.shape), np.float64)
    _lib.call("xrs_zonal_backproject_f64", idx_dev.ptr, idx_dev.size, tdev.ptr, len(stat_names), max(nz, 1), out.ptr,
              get_stream())
    return out.get(get_stream()) if like_numpy else out


_SHARDED_RANGE_LIMIT = 1 << 22      # widest span of zone ids whose presence map the ranks exchange


def _stats_sharded(zones, values, zone_ids, stat_names, nodata_values, return_type):
    """zonal.stats of a row-sharded raster: every rank reduces its own rows to per-zone partial sums, the partials
    are all-reduced, and every rank finishes the same table -- the block partials + combine of the reference's dask
    path (zonal.py:181-277) with a collective in place of the task graph.  The set of zone ids is agreed on first
    (global id range, then the union of the ranks' presence maps)."""
    same_layout(zones, values)
    if 'majority' in stat_names:
        raise NotImplementedError("'majority' needs a global sort and is not available for sharded rasters; "
                                  "pass stats_funcs without it")
    if zones.dtype != np.int32:
        raise TypeError("sharded zone rasters must be int32")
    comm = zones.comm
    stream = get_stream()
    zloc = zones.local
    res = DeviceArray((4,), np.float64)
    _lib.call("xrs_zonal_scan", zloc.ptr, _ZONE_DTYPE_CODE[zloc.dtype], zloc.size, res.ptr, stream)
    raw = res.get(stream)
    n_local = int(raw[2:3].view(np.uint64)[0])
    lo, hi = (raw[0], raw[1]) if n_local else (np.inf, -np.inf)
    if comm is not None and zones.world > 1:
        lo, neg_hi = (float(v) for v in comm.allreduce(np.array([lo, -hi]), 'min'))      # one small all-reduce
        hi = -neg_hi
    if not np.isfinite(lo):                                   # no rank holds a zone cell
        if return_type == 'pandas.DataFrame':
            return pd.DataFrame({'zone': np.empty(0, np.int32), **{name: np.empty(0) for name in stat_names}})
        # back-projection of an empty table: every plane is NaN everywhere, sharded like the input
        planes = [values.like(np.float64) for _ in stat_names]
        blank = np.full(zloc.shape, np.nan)
        for plane in planes:
            if blank.size:
                _lib.call("xrs_memcpy_h2d", plane.ptr, blank.ctypes.data, blank.nbytes, stream)
        _lib.call("xrs_stream_sync", stream)
        return ShardedStack(planes)
    rng = int(hi - lo) + 1
    if rng > _SHARDED_RANGE_LIMIT:
        raise NotImplementedError(f"zone ids span {rng} values; sharded zonal.stats handles up to {_SHARDED_RANGE_LIMIT}")
    present = DeviceArray((rng,), np.uint8)
    _lib.call("xrs_zonal_presence", zloc.ptr, _ZONE_DTYPE_CODE[zloc.dtype], zloc.size, float(lo), rng, present.ptr, stream)
    seen = present.get(stream)                                # uint8 flags: the union over the ranks is their maximum
    if comm is not None and zones.world > 1:
        seen = comm.allreduce(seen, 'max')
    mask = np.asarray(seen) > 0
    lut = np.where(mask, np.cumsum(mask, dtype=np.int64) - 1, -1).astype(np.int32)
    unique_zones = (np.flatnonzero(mask).astype(np.float64) + lo).astype(np.int32)
    nz = len(unique_zones)
    vloc = values.local if values.dtype in (np.float32, np.float64) else values.local.astype(np.float64)
    lut_dev = DeviceArray.from_numpy(lut)                     # (named: read by the kernels below until their sync)
    count, s1, s2, mn, mx, shift = zonal_partials(zloc, vloc, nz, nodata_values, comm if zones.world > 1 else None,
                                           table=(lo, rng, lut_dev))
    cols = finalize_stats(stat_names, count, s1, s2, mn, mx, None, shift)
    keep = np.arange(nz) if zone_ids is None else np.flatnonzero(np.isin(unique_zones, np.unique(zone_ids)))
    if return_type == 'pandas.DataFrame':
        frame = {'zone': unique_zones[keep]}
        for name in stat_names:
            frame[name] = cols[name][keep]
        return pd.DataFrame(frame)
    # back-projection (zonal.py:313-332) of the agreed table onto this rank's rows: every plane is a shard again
    table = np.full((len(stat_names), nz), np.nan)
    for i, name in enumerate(stat_names):
        table[i, keep] = cols[name][keep]
    tdev = DeviceArray.from_numpy(table)
    idx = DeviceArray(zloc.shape, np.int32)
    _lib.call("xrs_zonal_index", zloc.ptr, _ZONE_DTYPE_CODE[zloc.dtype], zloc.size, float(lo), rng, lut_dev.ptr, idx.ptr, stream)
    planes = [values.like(np.float64) for _ in stat_names]
    for i, plane in enumerate(planes):
        _lib.call("xrs_zonal_backproject_f64", idx.ptr, idx.size, tdev.ptr + i * nz * 8, 1, nz, plane.ptr, stream)
    _lib.call("xrs_stream_sync", stream)                      # idx / tdev / lut_dev are released on return
    return ShardedStack(planes)


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
    `nodata_values`, Dataset `values` and both return types run on the MI355X (row-sharded rasters: every statistic but
    `majority`, both return types -- the DataArray's planes are shards again).  `stats_funcs` as a dict of callables
    (zonal.py:304-310): the cells are grouped by zone on the device and each callable runs on the host on the 1-D array
    of its zone's valid values (ascending order; the reference hands them over in argsort order)."""
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

    custom = {}
    if isinstance(stats_funcs, dict):
        # {column name: callable}: the reference's numpy / cupy form (zonal.py:304-310, :407-411).  The zone's cells are
        # gathered on the MI355X (one sort by zone), the callable runs on the host on each zone's values.
        names = list(stats_funcs)
        for name, func in stats_funcs.items():
            if not callable(func):
                raise ValueError(name)
            custom[name] = func
        if isinstance(values.data, ShardedArray):
            raise NotImplementedError("custom stats callables need all cells of a zone in one place and are not "
                                      "available for row-sharded rasters")
    else:
        names = list(stats_funcs)
        for name in names:
            if name not in _DEFAULT_STATS:
                raise ValueError(f"Invalid stat name. {name} option not supported.")
    if return_type not in ('pandas.DataFrame', 'xarray.DataArray'):
        raise ValueError(f"unknown return_type {return_type!r}")
    if isinstance(values.data, ShardedArray):
        result = _stats_sharded(zones.data, values.data, zone_ids, names, nodata_values, return_type)
        if return_type == 'xarray.DataArray':
            coords = dict(values.coords.items())
            coords['stats'] = names
            return DataArray(result, coords=coords, dims=('stats',) + tuple(values.dims), attrs=values.attrs)
        return result
    if is_dask(values.data):
        if custom or return_type != 'pandas.DataFrame':
            # (zonal.py:628-633: a dask-backed `values` takes a LIST of the default statistics; its dask runner ignores
            #  return_type, :181-277)
            raise ValueError("Got dask-backed DataArray as `values` aggregate. `stats_funcs` must be a subset of default supported "
                             "stats `['mean', 'max', 'min', 'sum', 'std', 'var', 'count']`; the result is a DataFrame")
        return _stats_dask(zones.data, values.data, zone_ids, names, nodata_values)
    if not isinstance(values.data, (np.ndarray, DeviceArray)):
        raise TypeError("Unsupported Array Type: {}".format(type(values)))
    result = _stats_hip(zones.data, values.data, zone_ids, names, nodata_values, return_type, custom=custom)
    if return_type == 'xarray.DataArray':
        coords = dict(values.coords.items())
        coords['stats'] = names
        return DataArray(result, coords=coords, dims=('stats',) + tuple(values.dims), attrs=values.attrs)
    return result


def _stats_dask(zones, values, zone_ids, stat_names, nodata_values):
    """zonal.stats of dask-backed rasters: the reference's `_stats_dask_numpy` (zonal.py:181-277) -- per-block
    count / sum / sum of squares / min / max, combined by adding and reducing, mean / std / var from the combined sums
    (`_dask_mean / _dask_std / _dask_var`, :100-102) -- with the MI355X reducing every block (the partial-sums kernel the
    numpy and sharded backends use) and `distributed.combine_zonal_partials` as the combine.  One block in HBM at a time.
    Like upstream: `majority` is not among the block statistics and is left out of the frame (:83-99, :262-264); the blocks
    of `zones` and `values` are paired in order, so both rasters must be chunked alike.
    Returns a pandas DataFrame (upstream: a dask DataFrame of one partition -- `dask.dataframe` wraps it when importable)."""
    from .distributed import combine_zonal_partials
    if tuple(zones.chunks) != tuple(values.chunks):
        raise ValueError("zones and values must be chunked alike (zonal.py:198-199 pairs their blocks in order)")
    nby, nbx = zones.numblocks
    zblocks = [[np.asarray(zones.blocks[i, j].compute()) for j in range(nbx)] for i in range(nby)]
    uniq = [np.unique(b[np.isfinite(b)]) if np.issubdtype(b.dtype, np.floating) else np.unique(b) for row in zblocks for b in row]
    unique_zones = np.unique(np.concatenate(uniq)) if uniq else np.empty(0)
    nz = len(unique_zones)
    names = [n for n in stat_names if n != 'majority']
    if nz == 0:
        return pd.DataFrame({'zone': unique_zones, **{n: np.empty(0) for n in names}})
    parts = []
    for i in range(nby):
        for j in range(nbx):
            zb = zblocks[i][j]
            vb = np.ascontiguousarray(values.blocks[i, j].compute())
            ok = np.isfinite(zb) if np.issubdtype(zb.dtype, np.floating) else np.ones(zb.shape, bool)
            idx = np.where(ok, np.searchsorted(unique_zones, np.where(ok, zb, unique_zones[0])), -1).astype(np.int32)
            _, vdev = _stage(idx, vb)
            parts.append(zonal_partials(DeviceArray.from_numpy(np.ascontiguousarray(idx)), vdev, nz, nodata_values))
    count, s1, s2, mn, mx, shift = combine_zonal_partials(parts)
    cols = finalize_stats(names, count, s1, s2, mn, mx, None, shift)
    keep = np.arange(nz) if zone_ids is None else np.flatnonzero(np.isin(unique_zones, np.unique(zone_ids)))
    frame = {'zone': unique_zones[keep]}
    for n in names:
        frame[n] = cols[n][keep]
    df = pd.DataFrame(frame)
    try:                                                      # pragma: no cover (dask is not installable in the build image)
        import dask.dataframe as dd
        return dd.from_pandas(df, npartitions=1)
    except ImportError:
        return df


def _dense_index_any(data):
    """(unique finite values, int32 DeviceArray of dense indices) for a host or device raster."""
    if isinstance(data, DeviceArray):
        _lib.require_device()
        mapped = _dense_zone_index_device(data)
        if mapped is not None:
            return mapped
        data = data.get()
    uniq, idx = _dense_zone_index(np.asarray(data))
    return uniq, DeviceArray.from_numpy(idx)


def _crosstab_2d(zones_data, values_data, zone_ids, cat_ids, nodata_values, agg):
    # replaces _crosstab_numpy / _single_zone_crosstab_2d (zonal.py:699-800) for 2-D values
    _lib.require_device()
    stream = get_stream()
    unique_zones, zidx = _dense_index_any(zones_data)
    all_cats, cidx = _dense_index_any(values_data)
    nz, nc = len(unique_zones), len(all_cats)
    counts = np.zeros((nz, nc), dtype=np.uint64)
    if nz and nc:
        cdev = DeviceArray((nz * nc,), np.uint64)
        _lib.call("xrs_memset", cdev.ptr, 0, cdev.nbytes, stream)
        _lib.call("xrs_crosstab_counts", zidx.ptr, cidx.ptr, zidx.size, nz, nc, cdev.ptr, stream)
        counts = cdev.get(stream).reshape(nz, nc)
    return _crosstab_frame(unique_zones, all_cats, counts, zone_ids, cat_ids, nodata_values, agg)


def _crosstab_frame(unique_zones, all_cats, counts, zone_ids, cat_ids, nodata_values, agg):
    """(zones x categories) cell counts -> the reference's DataFrame (zonal.py:699-800): nodata category dropped, zone /
    category selections, counts or percentages of the zone's valid cells."""
    nc = len(all_cats)
    valid_cat = np.ones(nc, dtype=bool) if nodata_values is None else (all_cats != nodata_values)
    unique_cats = all_cats[valid_cat]
    counts = counts[:, valid_cat].astype(np.int64)
    if zone_ids is None:
        sel_zones = unique_zones
    else:
        sel_zones = [z for z in zone_ids if z in unique_zones]
    if cat_ids is None:
        sel_cats = unique_cats
    else:
        sel_cats = [c for c in cat_ids if c in unique_cats]
    zrows = [i for i, z in enumerate(unique_zones) if z in sel_zones]
    total = counts[zrows].sum(axis=1).astype(np.float32)                 # all valid cells of the zone (zonal.py:708-709)
    frame = {'zone': sel_zones}
    # The reference walks the categories in sorted order and advances its run start only past SELECTED ones
    # (`cat_start`, zonal.py:719-725): with `cat_ids` a strict subset, a selected category's column also holds the cells
    # of the unselected categories that sort between the previous selected category and it.  Reproduced as executed
    # (tests/golden/make_reference_exec.py, cases ct/*); with cat_ids=None the runs are the plain per-category counts.
    runs = np.cumsum(counts, axis=1)
    prev = None
    for c in sel_cats:
        frame[c] = None                   # (columns keep the caller's labels: 0 stays 0 where the raster holds 0.0)
    for j, c in enumerate(unique_cats):
        if c in sel_cats:
            frame[c] = runs[zrows, j] - (0 if prev is None else runs[zrows, prev])
            prev = j
    if agg == 'percentage':
        total[total == 0] = np.nan
        for c in sel_cats:
            frame[c] = frame[c] / total * 100
    return pd.DataFrame(frame)[['zone'] + list(sel_cats)]


def _sharded_dense_index(arr, what):
    """Dense indices of an integral row-sharded raster that every rank agrees on: (global unique values in the raster's
    dtype, int32 DeviceArray of this rank's dense indices; -1 for non-finite cells).  The ranks all-reduce the value range
    and then the union of their presence maps -- the protocol of `_stats_sharded`."""
    comm, stream, loc = arr.comm, get_stream(), arr.local
    code = _ZONE_DTYPE_CODE.get(loc.dtype)
    if code is None:
        raise TypeError(f"sharded {what} rasters must be int32 / int64 / float32 / float64")
    res = DeviceArray((4,), np.float64)
    _lib.call("xrs_zonal_scan", loc.ptr, code, loc.size, res.ptr, stream)
    raw = res.get(stream)
    n_local = int(raw[2:3].view(np.uint64)[0])
    integral = float(int(raw[3:4].view(np.int32)[0]) if n_local else 1)
    lo, hi = (raw[0], raw[1]) if n_local else (np.inf, -np.inf)
    if comm is not None and arr.world > 1:
        lo, neg_hi, integral = (float(v) for v in comm.allreduce(np.array([lo, -hi, integral]), 'min'))
        hi = -neg_hi
    if not integral:
        raise NotImplementedError(f"sharded {what} rasters must hold integral values (categories)")
    if not np.isfinite(lo):
        return np.empty(0, loc.dtype), DeviceArray.from_numpy(np.full(loc.shape, -1, np.int32))
    rng = int(hi - lo) + 1
    if rng > _SHARDED_RANGE_LIMIT:
        raise NotImplementedError(f"{what} values span {rng}; sharded rasters handle up to {_SHARDED_RANGE_LIMIT}")
    present = DeviceArray((rng,), np.uint8)
    _lib.call("xrs_zonal_presence", loc.ptr, code, loc.size, float(lo), rng, present.ptr, stream)
    seen = present.get(stream)
    if comm is not None and arr.world > 1:
        seen = comm.allreduce(seen, 'max')
    mask = np.asarray(seen) > 0
    lut = np.where(mask, np.cumsum(mask, dtype=np.int64) - 1, -1).astype(np.int32)
    uniq = (np.flatnonzero(mask).astype(np.float64) + lo).astype(loc.dtype)
    idx = DeviceArray(loc.shape, np.int32)
    lut_dev = DeviceArray.from_numpy(lut)          # named: the kernel reads it until the sync below (a temporary's block
    _lib.call("xrs_zonal_index", loc.ptr, code, loc.size, float(lo), rng, lut_dev.ptr, idx.ptr, stream)   # would be recycled)
    _lib.call("xrs_stream_sync", stream)
    del lut_dev
    return uniq, idx


def _crosstab_2d_sharded(zones, values, zone_ids, cat_ids, nodata_values, agg):
    """2-D crosstab of row-sharded rasters (the dask slot of the reference: zonal.py:868-916 -- per-block tables combined
    by summing): every rank counts its rows with the same kernel into the same (zones x categories) layout, one
    xrs_allreduce_u64 adds the tables, and every rank returns the whole DataFrame."""
    same_layout(zones, values)
    _lib.require_device()
    stream = get_stream()
    unique_zones, zidx = _sharded_dense_index(zones, "zone")
    all_cats, cidx = _sharded_dense_index(values, "category")
    nz, nc = len(unique_zones), len(all_cats)
    counts = np.zeros((nz, nc), dtype=np.uint64)
    if nz and nc:
        cdev = DeviceArray((nz * nc,), np.uint64)
        _lib.call("xrs_memset", cdev.ptr, 0, cdev.nbytes, stream)
        _lib.call("xrs_crosstab_counts", zidx.ptr, cidx.ptr, zidx.size, nz, nc, cdev.ptr, stream)
        counts = cdev.get(stream)
        if zones.comm is not None and zones.world > 1:
            counts = zones.comm.allreduce(counts, 'sum')
        counts = np.asarray(counts).reshape(nz, nc)
    return _crosstab_frame(unique_zones, all_cats, counts, zone_ids, cat_ids, nodata_values, agg)


def _crosstab_2d_dask(zones, values, zone_ids, cat_ids, nodata_values, agg):
    """2-D crosstab of dask-backed rasters: the reference's `_crosstab_dask_numpy` (zonal.py:813-916) -- one table of
    (zone, category) cell counts per block, the tables added, counts or percentages from the sum -- with the MI355X counting
    every block (the crosstab kernel of the numpy and sharded backends) against ONE agreed list of zones and categories
    (a first pass over the blocks collects them, as upstream's `np.unique(zones[...])` and `_find_cats` do).  One block in
    HBM at a time; returns a pandas DataFrame (upstream: a dask DataFrame of one partition, wrapped when importable)."""
    _lib.require_device()
    stream = get_stream()
    if tuple(zones.chunks) != tuple(values.chunks):
        raise ValueError("zones and values must be chunked alike (zonal.py:906-907 pairs their blocks in order)")
    nby, nbx = zones.numblocks

    def finite_unique(b):
        return np.unique(b[np.isfinite(b)]) if np.issubdtype(b.dtype, np.floating) else np.unique(b)
    zb = [[np.asarray(zones.blocks[i, j].compute()) for j in range(nbx)] for i in range(nby)]
    vb = [[np.asarray(values.blocks[i, j].compute()) for j in range(nbx)] for i in range(nby)]
    unique_zones = np.unique(np.concatenate([finite_unique(b) for row in zb for b in row]))
    all_cats = np.unique(np.concatenate([finite_unique(b) for row in vb for b in row]))
    nz, nc = len(unique_zones), len(all_cats)
    counts = np.zeros((nz, nc), dtype=np.uint64)

    def dense(b, uniq):
        ok = np.isfinite(b) if np.issubdtype(b.dtype, np.floating) else np.ones(b.shape, bool)
        return np.ascontiguousarray(np.where(ok, np.searchsorted(uniq, np.where(ok, b, uniq[0])), -1).astype(np.int32))
    if nz and nc:
        for i in range(nby):
            for j in range(nbx):
                zidx, cidx = DeviceArray.from_numpy(dense(zb[i][j], unique_zones)), DeviceArray.from_numpy(dense(vb[i][j], all_cats))
                cdev = DeviceArray((nz * nc,), np.uint64)
                _lib.call("xrs_memset", cdev.ptr, 0, cdev.nbytes, stream)
                _lib.call("xrs_crosstab_counts", zidx.ptr, cidx.ptr, zidx.size, nz, nc, cdev.ptr, stream)
                counts += cdev.get(stream).reshape(nz, nc)
    df = _crosstab_frame(unique_zones, all_cats, counts, zone_ids, cat_ids, nodata_values, agg)
    try:                                                      # pragma: no cover (dask is not installable in the build image)
        import dask.dataframe as dd
        return dd.from_pandas(df, npartitions=1)
    except ImportError:
        return df


def _crosstab_3d(zones_data, values_data, cat_labels, zone_ids, cat_ids, nodata_values, agg):
    # 3-D values: one layer per category, `agg` of the layer's values per zone (zonal.py:724-739)
    unique_zones, zidx = _dense_index_any(zones_data)
    nz = len(unique_zones)
    sel_zones = unique_zones if zone_ids is None else [z for z in zone_ids if z in unique_zones]
    sel_cats = list(cat_labels) if cat_ids is None else [c for c in cat_ids if c in cat_labels]
    zrows = [i for i, z in enumerate(unique_zones) if z in sel_zones]
    frame = {'zone': sel_zones}
    for j, cat in enumerate(cat_labels):
        if cat not in sel_cats:
            continue
        layer = values_data.rows(j, j + 1) if isinstance(values_data, DeviceArray) else values_data[j]
        if isinstance(layer, DeviceArray):
            layer = DeviceArray(layer.shape[1:], layer.dtype, _ptr=layer.ptr, _base=layer)
        count, s1, s2, mn, mx, shift = zonal_partials(zidx, layer, nz, nodata_values)
        majority = zonal_majority(zidx, layer, nz, nodata_values) if agg == 'majority' else None
        col = finalize_stats([agg], count, s1, s2, mn, mx, majority, shift)[agg]
        if agg == 'count':
            col = count.astype(np.int64)
        frame[cat] = col[zrows]
    return pd.DataFrame(frame)[['zone'] + sel_cats]


def crosstab(zones, values, zone_ids=None, cat_ids=None, layer=None, agg="count", nodata_values=None):
    """Cross-tabulated (categorical) statistics of `values` per zone.

    Same signature as `xrspatial.zonal.crosstab`.  2-D `values`: count / percentage of every category
    (distinct value) per zone, counted on the MI355X in one pass.  3-D `values`: one layer per category
    along dimension `layer`, `agg` of each layer per zone (the zonal.stats partials per layer)."""
    if not isinstance(zones, DataArray):
        raise TypeError("zones must be instance of DataArray")
    if not isinstance(values, DataArray):
        raise TypeError("values must be instance of DataArray")
    sharded = isinstance(zones.data, ShardedArray) or isinstance(values.data, ShardedArray)
    if sharded and not (isinstance(zones.data, ShardedArray) and isinstance(values.data, ShardedArray) and values.ndim == 2):
        raise NotImplementedError("zonal.crosstab of row-sharded (multi-GPU) arrays: both rasters sharded, 2-D values")
    if zones.ndim != 2:
        raise ValueError("zones must be 2D")
    if not (issubclass(zones.data.dtype.type, np.integer) or issubclass(zones.data.dtype.type, np.floating)):
        raise ValueError("`zones` must be an xarray of integers or floats")
    if not (issubclass(values.data.dtype.type, np.integer) or issubclass(values.data.dtype.type, np.floating)):
        raise ValueError("`values` must be an xarray of integers or floats")
    if values.ndim not in [2, 3]:
        raise ValueError("`values` must use either 2D or 3D coordinates.")
    if values.ndim == 2:
        validate_arrays(zones, values)
        if agg not in ("percentage", "count"):
            raise ValueError("`agg` method for 2D data array must be one of following ['percentage', 'count']")
        if sharded:
            return _crosstab_2d_sharded(zones.data, values.data, zone_ids, cat_ids, nodata_values, agg)
        if is_dask(values.data):
            return _crosstab_2d_dask(zones.data, values.data, zone_ids, cat_ids, nodata_values, agg)
        return _crosstab_2d(zones.data, values.data, zone_ids, cat_ids, nodata_values, agg)
    if agg not in _DEFAULT_STATS:
        raise ValueError(f"`agg` method for 3D numpy backed data array must be one of following {list(_DEFAULT_STATS)}")
    if layer is None:
        layer = 0
    try:
        cat_dim = values.dims[layer]
    except IndexError:
        raise ValueError("Invalid `layer`")
    data = values.data
    axis = list(values.dims).index(cat_dim)
    if axis != 0:
        if isinstance(data, DeviceArray):
            data = DeviceArray.from_numpy(np.ascontiguousarray(np.moveaxis(data.get(), axis, 0)))
        else:
            data = np.ascontiguousarray(np.moveaxis(np.asarray(data), axis, 0))
    if tuple(zones.shape) != tuple(data.shape[1:]):
        raise ValueError("Incompatible shapes")
    labels = np.asarray(values[cat_dim].values).tolist()
    return _crosstab_3d(zones.data, data, labels, zone_ids, cat_ids, nodata_values, agg)


# ------------------------------------------------------------------ zonal.trim / zonal.crop
def _match_bounds(data, values, invert):
    """(top, bottom, left, right) of the cells that equal one of `values` (`invert`: that equal none), exactly as
    the reference's four scans leave them (zonal.py:1651-1731 `_trim`, :1845-1940 `_crop`) -- including the case
    where no cell qualifies, in which every scan runs to the far edge.  One pass over the raster on the device."""
    _lib.require_device()
    vals = np.asarray(list(values), dtype=np.float64).reshape(-1)
    if vals.size > 16:
        raise ValueError("at most 16 values are supported by the MI355X backend")
    if len(data.shape) != 2:
        raise ValueError("expected a 2D raster")
    rows, cols = data.shape
    if isinstance(data, DeviceArray):
        dev = data if data.dtype in DTYPE_CODE else data.astype(np.float32)
    else:
        host = np.ascontiguousarray(data)
        dev = DeviceArray.from_numpy(host if host.dtype in DTYPE_CODE else host.astype(np.float64))
    box = DeviceArray((4,), np.int32)
    stream = get_stream()
    _lib.call("xrs_match_bbox", dev.ptr, DTYPE_CODE[dev.dtype], rows, cols, cols, vals.ctypes.data, int(vals.size),
              int(bool(invert)), box.ptr, stream)
    top, bottom, left, right = (int(v) for v in box.get(stream))
    if bottom < 0:                            # nothing qualified
        top, bottom, left, right = max(rows - 1, 0), 0, max(cols - 1, 0), 0
    return top, bottom, left, right


def _window(agg, top, bottom, left, right, name):
    """`agg[top:bottom + 1, left:right + 1]` with its coordinates and attributes; device-resident data stays in HBM."""
    ys, xs = slice(top, bottom + 1), slice(left, right + 1)
    data = agg.data
    if isinstance(data, DeviceArray):
        h, w = max(bottom + 1 - top, 0), max(right + 1 - left, 0)
        out = DeviceArray((h, w), data.dtype)
        if h and w:
            isz, pitch = data.dtype.itemsize, data.shape[1] * data.dtype.itemsize
            _lib.call("xrs_copy2d", out.ptr, w * isz, data.ptr + top * pitch + left * isz, pitch, w * isz, h, get_stream())
            _lib.call("xrs_stream_sync", get_stream())
    else:
        out = np.asarray(data)[ys, xs]
    dim_y, dim_x = agg.dims
    coords = {}
    for key, coord in agg.coords.items():
        cd = np.asarray(coord.data.get() if isinstance(coord.data, DeviceArray) else coord.data)
        index = tuple(ys if d == dim_y else xs if d == dim_x else slice(None) for d in coord.dims)
        coords[key] = DataArray(cd[index], dims=coord.dims, name=key, attrs=coord.attrs)
    return DataArray(out, name=name, dims=agg.dims, coords=coords, attrs=agg.attrs)


def trim(raster, values=(np.nan,), name='trim'):
    """Drop the outer rows and columns that hold nothing but `values`.  Same signature and result as
    `xrspatial.zonal.trim` (:1734-1842) -- including its quirk that NaN, compared with `==`, never matches, so the
    default `values=(nan,)` trims nothing."""
    top, bottom, left, right = _match_bounds(raster.data, values, invert=True)
    return _window(raster, top, bottom, left, right, name)


def crop(zones, values, zones_ids, name='crop'):
    """The window of `values` that bounds the cells of `zones` equal to one of `zones_ids`.  Same signature and result
    as `xrspatial.zonal.crop` (:1943-2061)."""
    top, bottom, left, right = _match_bounds(zones.data, zones_ids, invert=False)
    return _window(values, top, bottom, left, right, name)
