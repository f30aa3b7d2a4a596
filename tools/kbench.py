"""Per-kernel micro-benchmark on one MI355X: HIP-event timing of every hot-path kernel on a
device-resident raster, printed as ms / Mcells/s / algorithmic GB/s (SURVEY.md §8d byte counts).

    python tools/kbench.py [--size 16384] [--reps 20] [--only hillshade,focal5_mean]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import xrspatial_amd as xs  # noqa: E402
from tests import synth  # noqa: E402
from xrspatial_amd import _lib  # noqa: E402
from xrspatial_amd.convolution import circle_kernel  # noqa: E402


FAST_INPUTS = False   # --fast-inputs: stage ONE generated band and repeat it (profiling runs; same bytes moved)


def device_raster(rows, cols, maker, band=2048):
    out = xs.DeviceArray((rows, cols), np.float32)
    cache = None
    for y0 in range(0, rows, band):
        n = min(band, rows - y0)
        if FAST_INPUTS and cache is not None and cache.shape[0] == n:
            host = cache
        else:
            host = cache = maker(n, cols, y0)
        _lib.call("xrs_memcpy_h2d", out.ptr + y0 * cols * 4, host.ctypes.data, host.nbytes, None)
        _lib.call("xrs_stream_sync", None)
    return out


class Timer:
    def __init__(self):
        self.e0, self.e1 = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.call("xrs_event_create", ctypes.byref(self.e0))
        _lib.call("xrs_event_create", ctypes.byref(self.e1))

    def time(self, fn, reps, warmup=3, stream=None):
        for _ in range(warmup):
            fn()
        _lib.call("xrs_stream_sync", stream)
        times = []
        for _ in range(reps):
            _lib.call("xrs_event_record", self.e0, stream)
            fn()
            _lib.call("xrs_event_record", self.e1, stream)
            _lib.call("xrs_event_sync", self.e1)
            ms = ctypes.c_float()
            _lib.call("xrs_event_elapsed_ms", self.e0, self.e1, ctypes.byref(ms))
            times.append(ms.value)
        return float(np.median(times)), float(np.min(times))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--fast-inputs", action="store_true")
    args = ap.parse_args()
    global FAST_INPUTS
    FAST_INPUTS = args.fast_inputs
    _lib.require_device()
    n = args.size
    cells = n * n
    name = ctypes.create_string_buffer(256)
    _lib.call("xrs_device_name", 0, name, 256)
    print("device:", name.value.decode(), " raster:", n, "x", n, flush=True)

    only = [s for s in args.only.split(",") if s]

    def needs(*prefixes):
        """Is any selected case one of these?  (inputs of unselected cases are not staged)"""
        return not only or any(c.startswith(p) for c in only for p in prefixes)

    t0 = time.time()
    dem = device_raster(n, n, lambda r, c, y0: synth.asv_dem(r, c, y0=y0, total_rows=n))
    small = xs.DeviceArray((8, 8), np.int32)
    b2 = device_raster(n, n, lambda r, c, y0: synth.bands((r, c), 100 + y0)) if needs("ndvi", "evi", "savi") else small
    b3 = device_raster(n, n, lambda r, c, y0: synth.bands((r, c), 300 + y0)) if needs("evi") else small
    zones = zones_sc = small
    if needs("zonal_1000", "crosstab", "crop"):
        zones = xs.DeviceArray((n, n), np.int32)
        for y0 in range(0, n, 2048):
            z = synth.block_zones(min(2048, n - y0), n, y0=y0)
            _lib.call("xrs_memcpy_h2d", zones.ptr + y0 * n * 4, z.ctypes.data, z.nbytes, None)
            _lib.call("xrs_stream_sync", None)
    if needs("zonal_1000_scattered"):
        zones_sc = xs.DeviceArray((n, n), np.int32)    # scattered: random zone per 8x8-cell block (stresses the atomics)
        rng = np.random.default_rng(9)
        for y0 in range(0, n, 2048):
            blocks = rng.integers(0, 1000, size=(2048 // 8, n // 8)).astype(np.int32)
            z = np.repeat(np.repeat(blocks, 8, axis=0), 8, axis=1)
            _lib.call("xrs_memcpy_h2d", zones_sc.ptr + y0 * n * 4, z.ctypes.data, z.nbytes, None)
            _lib.call("xrs_stream_sync", None)
    print("inputs staged in %.1f s" % (time.time() - t0), flush=True)

    outs = [xs.DeviceArray((n, n), np.float32) for _ in range(7)]
    out64 = xs.DeviceArray((n, n), np.float64) if needs("hillshade_f64out", "focal_mean3x3_f64") else small
    k5 = np.ascontiguousarray(circle_kernel(1, 1, 2))
    k3 = np.ones((3, 3))
    k25 = np.ascontiguousarray(circle_kernel(1, 1, 12))
    k7 = np.ascontiguousarray(circle_kernel(1, 1, 3))
    kb5, kb11 = np.ones((5, 5)), np.ones((11, 11))
    k13 = np.ascontiguousarray(circle_kernel(1, 1, 6))
    w5 = np.ascontiguousarray(k5 / k5.sum())
    w25 = np.ascontiguousarray(k25 / k25.sum())
    k9 = circle_kernel(1, 1, 4)
    w9 = np.ascontiguousarray(k9 / k9.sum())
    work = xs.DeviceArray((1 << 20,), np.uint8)       # kernel copy + tile map of the separable box walk (xrs_focal_workspace_bytes)
    WB = work.nbytes
    from xrspatial_amd.convolution import annulus_kernel
    kb15, kb25 = np.ones((15, 15)), np.ones((25, 25))
    wb25 = np.ascontiguousarray(kb25 / kb25.sum())
    ka21 = np.ascontiguousarray(annulus_kernel(1, 1, 10, 6))       # 21x21 ring, inner radius 6
    ka25 = np.ascontiguousarray(annulus_kernel(1, 1, 12, 4))       # 25x25 ring, inner radius 4
    wa21, wa25 = np.ascontiguousarray(ka21 / ka21.sum()), np.ascontiguousarray(ka25 / ka25.sum())

    def fstats(k, ptrs, mask):
        K = k.shape[0]
        return lambda: L("xrs_focal_stats_f32_ex", dem.ptr, ptrs, mask, n, n, n, n, k.ctypes.data, K, K, work.ptr, WB, 0, 0, 0, S)
    ptr1 = (ctypes.c_void_p * 7)()
    ptr1[0] = outs[0].ptr
    ptr7 = (ctypes.c_void_p * 7)(*[o.ptr for o in outs])
    nz = 1000
    zc = xs.DeviceArray((nz,), np.uint64)
    zs, zq = xs.DeviceArray((nz,), np.float64), xs.DeviceArray((nz,), np.float64)
    zmn, zmx = xs.DeviceArray((nz,), np.float32), xs.DeviceArray((nz,), np.float32)
    ex = np.array([np.nan])
    L = _lib.call
    S = None

    def zonal():
        L("xrs_zonal_init", zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, nz, S)
        L("xrs_zonal_partials_f32", zones.ptr, dem.ptr, cells, nz, 0.0, 0, 0.0, zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, S)

    xt = xs.DeviceArray((1000 * 32,), np.uint64)
    cats = zones64 = cats8 = small
    if needs("crosstab"):
        cats = xs.DeviceArray((n, n), np.int32)
        zones64 = xs.DeviceArray((n, n), np.int32)
        cats8 = xs.DeviceArray((n, n), np.int32)
    rng2 = np.random.default_rng(10)
    for y0 in range(0, n if needs("crosstab") else 0, 2048):
        c = rng2.integers(0, 32, size=(2048, n)).astype(np.int32)               # categorical: changes every cell
        _lib.call("xrs_memcpy_h2d", cats.ptr + y0 * n * 4, c.ctypes.data, c.nbytes, None)
        c8 = (c & 7).astype(np.int32)
        _lib.call("xrs_memcpy_h2d", cats8.ptr + y0 * n * 4, c8.ctypes.data, c8.nbytes, None)
        z64 = synth.block_zones(2048, n, n_zones=64, block=2048, y0=y0)
        _lib.call("xrs_memcpy_h2d", zones64.ptr + y0 * n * 4, z64.ctypes.data, z64.nbytes, None)
        _lib.call("xrs_stream_sync", None)
    lat1 = xs.DeviceArray.from_numpy(np.linspace(40.0, 41.0, n))
    lon1 = xs.DeviceArray.from_numpy(np.linspace(10.0, 11.0, n))
    geo_work = xs.DeviceArray((int(_lib.load().xrs_geodesic_workspace_bytes(n, n)),), np.uint8)
    A2, B2 = 6378137.0 ** 2, 6356752.314245 ** 2
    mom = xs.DeviceArray((4,), np.float64)
    out8 = xs.DeviceArray((n, n), np.int8)
    mm6 = xs.DeviceArray.from_numpy(np.array([0, 4000, 0, 4000, 0, 4000], np.float32))
    mm2 = xs.DeviceArray((2,), np.float32)
    box4 = xs.DeviceArray((4,), np.int32)
    zero = np.zeros(1)

    zones5k = xs.DeviceArray((n, n), np.int32) if needs("zonal_5000") else small
    for y0 in range(0, n if needs("zonal_5000") else 0, 2048):
        z5 = synth.block_zones(2048, n, n_zones=5000, block=128, y0=y0)
        _lib.call("xrs_memcpy_h2d", zones5k.ptr + y0 * n * 4, z5.ctypes.data, z5.nbytes, None)
        _lib.call("xrs_stream_sync", None)
    z5c = xs.DeviceArray((5000,), np.uint64)
    z5s, z5q = xs.DeviceArray((5000,), np.float64), xs.DeviceArray((5000,), np.float64)
    z5mn, z5mx = xs.DeviceArray((5000,), np.float32), xs.DeviceArray((5000,), np.float32)

    def zonal5k():
        L("xrs_zonal_init", z5c.ptr, z5s.ptr, z5q.ptr, z5mn.ptr, z5mx.ptr, 5000, S)
        L("xrs_zonal_partials_f32", zones5k.ptr, dem.ptr, cells, 5000, 0.0, 0, 0.0, z5c.ptr, z5s.ptr, z5q.ptr, z5mn.ptr, z5mx.ptr, S)

    def zonal_scatter():
        L("xrs_zonal_init", zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, nz, S)
        L("xrs_zonal_partials_f32", zones_sc.ptr, dem.ptr, cells, nz, 0.0, 0, 0.0, zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, S)

    # name -> (callable, algorithmic bytes per cell)
    cases = {
        "copy_d2d": (lambda: L("xrs_memcpy_d2d", outs[0].ptr, dem.ptr, cells * 4, S), 8),
        "copy_kernel": (lambda: L("xrs_copy_f32", dem.ptr, outs[0].ptr, cells, S), 8),
        "stream_1r2w": (lambda: L("xrs_stream_mix_f32", dem.ptr, ptr7, 2, cells, S), 12),
        "stream_1r3w": (lambda: L("xrs_stream_mix_f32", dem.ptr, ptr7, 3, cells, S), 16),
        "stream_1r4w": (lambda: L("xrs_stream_mix_f32", dem.ptr, ptr7, 4, cells, S), 20),
        "stream_1r7w": (lambda: L("xrs_stream_mix_f32", dem.ptr, ptr7, 7, cells, S), 32),
        "hillshade": (lambda: L("xrs_hillshade_f32", dem.ptr, outs[0].ptr, 0, n, n, n, n, 225.0, 25.0, 0, 0, S), 8),
        "hillshade_f64out": (lambda: L("xrs_hillshade_f32", dem.ptr, out64.ptr, 1, n, n, n, n, 225.0, 25.0, 0, 0, S), 12),
        "slope": (lambda: L("xrs_slope_f32", dem.ptr, outs[0].ptr, n, n, n, n, 1.0, 1.0, 0, 0, S), 8),
        "aspect": (lambda: L("xrs_aspect_f32", dem.ptr, outs[0].ptr, n, n, n, n, 0, 0, S), 8),
        "curvature": (lambda: L("xrs_curvature_f32", dem.ptr, outs[0].ptr, n, n, n, n, 1.0, 0, 0, S), 8),
        "terrain_fused4": (lambda: L("xrs_terrain_fused_f32", dem.ptr, outs[0].ptr, outs[1].ptr, outs[2].ptr,
                                     outs[3].ptr, n, n, n, n, 1.0, 1.0, 225.0, 25.0, 0, 0, S), 20),
        "terrain_hill_aspect_curv": (lambda: L("xrs_terrain_fused_f32", dem.ptr, None, outs[1].ptr, outs[2].ptr,
                                               outs[3].ptr, n, n, n, n, 1.0, 1.0, 225.0, 25.0, 0, 0, S), 16),
        "pass_hill_focal5": (lambda: L("xrs_raster_pass_f32", dem.ptr, None, None, None, outs[3].ptr, outs[4].ptr,
                                       k5.ctypes.data, 5, 5, None, n, n, n, n, 1.0, 1.0, 225.0, 25.0, 0, 0, S), 12),
        "pass_hill_slope_focal5": (lambda: L("xrs_raster_pass_f32", dem.ptr, outs[0].ptr, None, None, outs[3].ptr,
                                             outs[4].ptr, k5.ctypes.data, 5, 5, None, n, n, n, n, 1.0, 1.0, 225.0,
                                             25.0, 0, 0, S), 16),
        "pass_all4_focal5": (lambda: L("xrs_raster_pass_f32", dem.ptr, outs[0].ptr, outs[1].ptr, outs[2].ptr,
                                       outs[3].ptr, outs[4].ptr, k5.ctypes.data, 5, 5, None, n, n, n, n, 1.0, 1.0,
                                       225.0, 25.0, 0, 0, S), 24),
        "pass_aspect_focal5": (lambda: L("xrs_raster_pass_f32", dem.ptr, None, outs[1].ptr, None, None,
                                         outs[4].ptr, k5.ctypes.data, 5, 5, None, n, n, n, n, 1.0, 1.0, 225.0,
                                         25.0, 0, 0, S), 12),
        "pass_slope_focal5": (lambda: L("xrs_raster_pass_f32", dem.ptr, outs[0].ptr, None, None, None,
                                        outs[4].ptr, k5.ctypes.data, 5, 5, None, n, n, n, n, 1.0, 1.0, 225.0,
                                        25.0, 0, 0, S), 12),
        "pass_curv_hill_focal5": (lambda: L("xrs_raster_pass_f32", dem.ptr, None, None, outs[2].ptr, outs[3].ptr,
                                            outs[4].ptr, k5.ctypes.data, 5, 5, None, n, n, n, n, 1.0, 1.0, 225.0,
                                            25.0, 0, 0, S), 16),
        "pass_slope_curv_hill_focal5": (lambda: L("xrs_raster_pass_f32", dem.ptr, outs[0].ptr, None, outs[2].ptr,
                                                  outs[3].ptr, outs[4].ptr, k5.ctypes.data, 5, 5, None, n, n, n, n,
                                                  1.0, 1.0, 225.0, 25.0, 0, 0, S), 20),
        "pass_hill_focal3": (lambda: L("xrs_raster_pass_f32", dem.ptr, None, None, None, outs[3].ptr, outs[4].ptr,
                                       k3.ctypes.data, 3, 3, None, n, n, n, n, 1.0, 1.0, 225.0, 25.0, 0, 0, S), 12),
        "ndvi": (lambda: L("xrs_normalized_ratio_f32", dem.ptr, b2.ptr, outs[0].ptr, cells, S), 12),
        "evi": (lambda: L("xrs_evi_f32", dem.ptr, b2.ptr, b3.ptr, outs[0].ptr, cells, 6.0, 7.5, 1.0, 2.5, S), 16),
        "savi": (lambda: L("xrs_savi_f32", dem.ptr, b2.ptr, outs[0].ptr, cells, 1.0, S), 12),
        "focal5_mean": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr1, 1, n, n, n, n, k5.ctypes.data, 5, 5, None, 0, 0, S), 8),
        "focal3_mean": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr1, 1, n, n, n, n, k3.ctypes.data, 3, 3, None, 0, 0, S), 8),
        "focal5_stats7": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr7, 127, n, n, n, n, k5.ctypes.data, 5, 5, None, 0, 0, S), 32),
        "box3_stats7": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr7, 127, n, n, n, n, k3.ctypes.data, 3, 3, None, 0, 0, S), 32),
        "box5_stats7": (fstats(kb5, ptr7, 127), 32),
        "box11_mean": (fstats(kb11, ptr1, 1), 8),
        "box11_stats7": (fstats(kb11, ptr7, 127), 32),
        "box11_mean_general": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr1, 1, n, n, n, n, kb11.ctypes.data, 11, 11, None, 0, 0, S), 8),
        "box15_mean": (fstats(kb15, ptr1, 1), 8),
        "box15_stats7": (fstats(kb15, ptr7, 127), 32),
        "box25_mean": (fstats(kb25, ptr1, 1), 8),
        "box25_meanvarstd": (fstats(kb25, ptr7, 0b110001), 16),
        "box25_minmaxrange": (fstats(kb25, ptr7, 0b1110), 16),
        "box25_stats7": (fstats(kb25, ptr7, 127), 32),
        "convolve25_box": (lambda: L("xrs_convolve2d_f32", dem.ptr, outs[0].ptr, n, n, n, n, wb25.ctypes.data, 25, 25, work.ptr, 0, 0, S), 8),
        "convolve21_annulus": (lambda: L("xrs_convolve2d_f32", dem.ptr, outs[0].ptr, n, n, n, n, wa21.ctypes.data, 21, 21, work.ptr, 0, 0, S), 8),
        "convolve25_annulus": (lambda: L("xrs_convolve2d_f32", dem.ptr, outs[0].ptr, n, n, n, n, wa25.ctypes.data, 25, 25, work.ptr, 0, 0, S), 8),
        "annulus21_mean": (fstats(ka21, ptr1, 1), 8),
        "annulus21_stats7": (fstats(ka21, ptr7, 127), 32),
        "annulus25_mean": (fstats(ka25, ptr1, 1), 8),
        "annulus25_stats7": (fstats(ka25, ptr7, 127), 32),
        "focal21_stats7": (fstats(np.ascontiguousarray(circle_kernel(1, 1, 10)), ptr7, 127), 32),
        "focal7_mean": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr1, 1, n, n, n, n, k7.ctypes.data, 7, 7, None, 0, 0, S), 8),
        "focal7_stats7": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr7, 127, n, n, n, n, k7.ctypes.data, 7, 7, None, 0, 0, S), 32),
        "focal13_mean": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr1, 1, n, n, n, n, k13.ctypes.data, 13, 13, None, 0, 0, S), 8),
        "focal13_stats7": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr7, 127, n, n, n, n, k13.ctypes.data, 13, 13, None, 0, 0, S), 32),
        "focal25_mean": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr1, 1, n, n, n, n, k25.ctypes.data, 25, 25, None, 0, 0, S), 8),
        "focal25_stats7": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr7, 127, n, n, n, n, k25.ctypes.data, 25, 25, None, 0, 0, S), 32),
        "focal25_sum": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr7, 1 << 6, n, n, n, n, k25.ctypes.data, 25, 25, None, 0, 0, S), 8),
        "focal25_minmaxrange": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr7, 0b1110, n, n, n, n, k25.ctypes.data, 25, 25, None, 0, 0, S), 16),
        "focal25_meanvarstd": (lambda: L("xrs_focal_stats_f32", dem.ptr, ptr7, 0b110001, n, n, n, n, k25.ctypes.data, 25, 25, None, 0, 0, S), 16),
        "convolve5": (lambda: L("xrs_convolve2d_f32", dem.ptr, outs[0].ptr, n, n, n, n, w5.ctypes.data, 5, 5, work.ptr, 0, 0, S), 8),
        "convolve25_circle": (lambda: L("xrs_convolve2d_f32", dem.ptr, outs[0].ptr, n, n, n, n, w25.ctypes.data, 25, 25, work.ptr, 0, 0, S), 8),
        "convolve9_circle": (lambda: L("xrs_convolve2d_f32", dem.ptr, outs[0].ptr, n, n, n, n, w9.ctypes.data, 9, 9, work.ptr, 0, 0, S), 8),
        "focal_mean3x3_f64": (lambda: L("xrs_focal_mean3x3", dem.ptr, 0, out64.ptr, n, n, n, n, ex.ctypes.data, 1, 0, 0, S), 12),
        "zonal_1000": (zonal, 8),
        "zonal_1000_scattered": (zonal_scatter, 8),
        "zonal_5000": (zonal5k, 8),
        "crosstab_1000x32": (lambda: (L("xrs_memset", xt.ptr, 0, 1000 * 32 * 8, S),
                                      L("xrs_crosstab_counts", zones.ptr, cats.ptr, cells, 1000, 32, xt.ptr, S)), 8),
        "crosstab_64x8": (lambda: (L("xrs_memset", xt.ptr, 0, 64 * 8 * 8, S),
                                   L("xrs_crosstab_counts", zones64.ptr, cats8.ptr, cells, 64, 8, xt.ptr, S)), 8),
        "geodesic_slope": (lambda: L("xrs_geodesic_f32", dem.ptr, 0, lat1.ptr, lon1.ptr, 0, outs[0].ptr, n, n, n, n, n,
                                     A2, B2, 1.0, 0, geo_work.ptr, 0, 0, S), 8),
        "geodesic_aspect": (lambda: L("xrs_geodesic_f32", dem.ptr, 0, lat1.ptr, lon1.ptr, 0, outs[0].ptr, n, n, n, n, n,
                                      A2, B2, 1.0, 1, geo_work.ptr, 0, 0, S), 8),
        # true_color: dem stands in for the three bands (12 B read) + 4 B RGBA written; float64 exp per channel
        "true_color": (lambda: L("xrs_true_color_u8", dem.ptr, dem.ptr, dem.ptr, dem.ptr, 9, cells, mm6.ptr, 1.0, 10.0, 0.125,
                                 outs[0].ptr, S), 16),
        "nan_minmax": (lambda: L("xrs_nan_minmax_f32", dem.ptr, cells, mm2.ptr, S), 4),
        "trim_bbox_f32": (lambda: L("xrs_match_bbox", dem.ptr, 9, n, n, n, zero.ctypes.data, 1, 1, box4.ptr, S), 4),
        "crop_bbox_i32": (lambda: L("xrs_match_bbox", zones.ptr, 4, n, n, n, zero.ctypes.data, 1, 0, box4.ptr, S), 4),
        "nan_moments": (lambda: L("xrs_nan_moments_f32", dem.ptr, cells, mom.ptr, S), 4),
        "hotspots_classify": (lambda: L("xrs_hotspots_classify_f32", dem.ptr, out8.ptr, cells, 50.0, 20.0, S), 5),
    }
    timer = Timer()
    results = {}
    print(f"{'kernel':28s} {'ms(med)':>9s} {'ms(min)':>9s} {'Mcells/s':>11s} {'GB/s(alg)':>10s}")
    for name_, (fn, bpc) in cases.items():
        if only and name_ not in only:
            continue
        reps = args.reps
        med, mn = timer.time(fn, reps, warmup=2)
        gbs = cells * bpc / (med * 1e-3) / 1e9
        results[name_] = {"ms_median": med, "ms_min": mn, "mcells_s": cells / (med * 1e-3) / 1e6, "gb_s": gbs,
                          "bytes_per_cell": bpc}
        print(f"{name_:28s} {med:9.3f} {mn:9.3f} {cells / (med * 1e-3) / 1e6:11.0f} {gbs:10.0f}", flush=True)
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"size": n, "results": results}, fh, indent=1)


if __name__ == "__main__":
    main()
