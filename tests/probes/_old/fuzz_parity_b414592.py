"""Differential fuzzing of the public API against the CPU oracle (test infrastructure; needs an MI355X; lives under
tests/ because it imports the oracle -- not collected by pytest).

Seeded random cases: raster shape (biased to the awkward ones -- fewer than 4 columns, widths 1..3 mod 4, one off a
256 / 1024 tile edge, single rows), dtype, NaN density, inf cells, backend (numpy / device-resident), operator and its
parameters.  Every result is compared with the oracle to the tolerance the parity tests use.  Prints one line per
failure and a summary; exit code 1 if anything differed.

    python tests/fuzz_parity.py [--cases 400] [--seed 1] [--max-cells 400000]
"""
import argparse
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import xrspatial_amd as xs  # noqa: E402
from oracle import c_oracle as corc  # noqa: E402
from oracle import xrs_oracle as orc  # noqa: E402
from xrspatial_amd import focal, zonal  # noqa: E402
from xrspatial_amd.convolution import annulus_kernel, circle_kernel, convolution_2d  # noqa: E402
from xrspatial_amd.multispectral import true_color  # noqa: E402

RTOL = 1e-5


def host(a):
    return a.get() if hasattr(a, "get") and not isinstance(a, np.ndarray) else np.asarray(a)


BIG = False     # --big: rasters large enough (>= 32 MiB) for the banded upload / compute / download pipelines
WINDOWS = False # --windows: only focal.apply / focal_stats with 9x9 .. 25x25 circles, boxes and annuli on rasters of several wave
                # tiles that carry nodata the way real rasters do (regions with straight and ragged rims, scattered cells at many
                # densities, isolated valid cells inside nodata, +-inf, cliffs and lakes): the large-window walkers' cascade


def pick_shape(rng, max_cells):
    if WINDOWS:
        return int(rng.choice([131, 262, 300, 393, 450, 560])), int(rng.choice([128, 256, 300, 512, 640, 900, 1330]))
    if BIG:
        return int(rng.integers(2100, 4200)), int(rng.choice([2048, 2052, 3000, 3601, 4096, 4100]))
    special = [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 255, 256, 257, 259, 511, 513, 1023, 1025, 1030]
    while True:
        rows = int(rng.choice(special)) if rng.random() < 0.4 else int(rng.integers(1, 700))
        cols = int(rng.choice(special)) if rng.random() < 0.5 else int(rng.integers(1, 1400))
        if rows * cols <= max_cells:
            return rows, cols


def make_raster(rng, shape, dtype, allow_nan=True):
    base = 1000 + 400 * np.sin(np.arange(shape[1])[None, :] / 37.0) * np.cos(np.arange(shape[0])[:, None] / 23.0)
    z = base + rng.normal(0, rng.choice([0.01, 1.0, 50.0]), shape)
    if np.issubdtype(dtype, np.integer):
        return np.clip(z, 0, np.iinfo(dtype).max).astype(dtype)
    z = z.astype(dtype)
    if allow_nan and WINDOWS:
        rows, cols = shape
        for _ in range(int(rng.integers(1, 4))):
            kind = rng.choice(["rows", "cols", "ragged", "block", "scatter", "sparse", "lake", "cliff", "inf", "none"])
            if kind == "rows":
                a = int(rng.integers(0, rows)); z[a:a + int(rng.integers(1, rows)), :] = np.nan
            elif kind == "cols":
                a = int(rng.integers(0, cols)); z[:, a:a + int(rng.integers(1, cols))] = np.nan
            elif kind == "ragged":
                edge = int(rng.integers(0, cols)) + (np.arange(rows) // int(rng.integers(1, 9))) % int(rng.integers(2, 30))
                side = rng.random() < 0.5
                m = np.arange(cols)[None, :] < edge[:, None]
                z[m if side else ~m] = np.nan
            elif kind == "block":
                a, b = int(rng.integers(0, rows)), int(rng.integers(0, cols))
                z[a:a + int(rng.integers(1, 120)), b:b + int(rng.integers(1, 300))] = np.nan
            elif kind == "scatter":
                z[rng.random(shape) < rng.choice([1e-4, 1e-3, 3e-3, 0.01, 0.05, 0.3, 0.9])] = np.nan
            elif kind == "sparse":                       # nodata everywhere but a few isolated cells
                keep = rng.random(shape) < rng.choice([1e-4, 1e-3, 0.01])
                z[~keep] = np.nan
            elif kind == "lake":
                a, b = int(rng.integers(0, rows)), int(rng.integers(0, cols))
                z[a:a + int(rng.integers(5, 90)), b:b + int(rng.integers(5, 200))] = dtype.type(rng.choice([0.0, 777.25, 1234.567, -5.25, 16777217.0, 3.3e-5]))
            elif kind == "cliff":
                a = int(rng.integers(0, cols)); z[:, a:] += dtype.type(rng.choice([50.0, 3000.0, -1e5, 1e7]))
            elif kind == "inf":
                for _i in range(int(rng.integers(1, 4))):
                    z.flat[rng.integers(0, z.size)] = rng.choice([np.inf, -np.inf])
        return z
    if allow_nan:
        frac = rng.choice([0.0, 0.0, 1e-3, 0.05, 0.4, 1.0])
        if frac:
            z[rng.random(shape) < frac] = np.nan
        if rng.random() < 0.1 and z.size:
            z.flat[rng.integers(0, z.size)] = np.inf
    return z


def agg_of(data, backend, res=(30.0, 30.0)):
    d = xs.DeviceArray.from_numpy(data) if backend == "hip" else data
    return xs.DataArray(d, dims=["y", "x"], attrs={"res": res})


def close(got, want, rtol=RTOL, atol=0.0):
    got, want = host(got), np.asarray(want)
    if got.shape != want.shape:
        return f"shape {got.shape} vs {want.shape}"
    with np.errstate(all="ignore"):
        if got.dtype.kind in "iu" or want.dtype.kind in "iu":
            bad = got != want
        else:
            both_nan = np.isnan(got) & np.isnan(want)
            bad = ~both_nan & ~(np.abs(got - want) <= atol + rtol * np.abs(want)) & ~(got == want)
    if bad.any():
        i = tuple(int(v[0]) for v in np.nonzero(bad))
        return f"{int(bad.sum())} cells differ, first at {i}: got {got[i]!r} want {want[i]!r}"
    return None


def random_kernel(rng):
    if WINDOWS:
        kind = rng.choice(["circle", "circle", "box", "annulus"])
        r = int(rng.choice([4, 5, 6, 7, 8, 9, 10, 11, 12, 12]))
        return (circle_kernel(1, 1, r) if kind == "circle" else np.ones((2 * r + 1, 2 * r + 1)) if kind == "box"
                else annulus_kernel(1, 1, r, int(rng.integers(1, r))))
    kind = rng.choice(["circle", "circle", "box", "annulus", "custom"])
    if kind == "circle":
        return circle_kernel(1, 1, int(rng.choice([1, 2, 2, 3, 3, 4, 5, 6, 8, 10, 12])))
    if kind == "box":
        k = int(rng.choice([1, 3, 5, 5, 7, 7, 9, 11, 15, 25]))
        return np.ones((k, k))
    if kind == "annulus":
        outer = int(rng.choice([2, 3, 4, 5, 6, 8, 10, 12]))
        return annulus_kernel(1, 1, outer, int(rng.integers(1, outer)))
    kh, kw = int(rng.choice([1, 3, 5, 7])), int(rng.choice([1, 3, 5, 7]))
    k = (rng.random((kh, kw)) < 0.6).astype(np.float64)
    k[kh // 2, kw // 2] = 1
    return k


def one_case(rng, max_cells):
    shape = pick_shape(rng, max_cells)
    backend = str(rng.choice(["numpy", "hip"]))
    op = str(rng.choice(["slope", "aspect", "curvature", "hillshade", "mean", "apply", "focal_stats", "convolve", "ndvi", "evi",
                         "zonal", "crosstab", "hotspots", "fuse", "trim", "true_color"]))
    dtype = np.dtype(rng.choice([np.float32, np.float32, np.float64, np.int16, np.uint8, np.int32]))
    if WINDOWS:
        op, dtype = str(rng.choice(["apply", "focal_stats", "focal_stats"])), np.dtype(rng.choice([np.float32, np.float32, np.float64]))
    desc = f"{op} {shape} {dtype} {backend}"
    z = make_raster(rng, shape, dtype)
    agg = agg_of(z, backend)
    with np.errstate(all="ignore"):
        if op == "slope":
            return desc, close(xs.slope(agg).data, orc.slope(z, 30.0, 30.0))
        if op == "aspect":
            return desc, close(xs.aspect(agg).data, orc.aspect(z))
        if op == "curvature":
            return desc, close(xs.curvature(agg).data, orc.curvature(z, 30.0), atol=1e-9)
        if op == "hillshade":
            az, alt = float(rng.integers(0, 360)), float(rng.integers(1, 90))
            if min(shape) < 2:            # np.gradient refuses (the reference raises); the device returns the NaN border
                return desc, None if np.isnan(host(xs.hillshade(agg, az, alt).data)).all() else "expected all NaN"
            return desc + f" az={az} alt={alt}", close(xs.hillshade(agg, az, alt).data, orc.hillshade(z, az, alt), atol=1e-6)
        if op == "mean":
            passes = int(rng.integers(1, 4))
            return desc + f" passes={passes}", close(focal.mean(agg, passes=passes).data, orc.focal_mean3x3(z, passes=passes), rtol=1e-12)
        if op in ("apply", "focal_stats", "convolve", "hotspots"):
            k = random_kernel(rng)
            desc += f" k={k.shape} taps={int(k.sum())}"
            if max(k.shape) // 2 >= min(shape) and op == "hotspots":
                return desc, None
            # extrema are bit-exact; the moments and the window sum of LARGE windows are float32 sums behind a guard (mom_impl.h:
            # documented <= 2e-6 for mean / std / sum, <= 5e-6 for var, contract 1e-5) -- at 1e-6 for every plane, 3 000 cases
            # found five windows between 1.0e-6 and 1.1e-6 (profiles/r04/r04z_fuzz_s5*.log)
            def tol(stat):
                if WINDOWS and stat == "var" and k.size >= 49:
                    return 1e-5                                  # (the contract; 2 400 adversarial cases: one window at 5.3e-6)
                return 1e-6 if stat in ("max", "min", "range") or k.size < 49 else 5e-6
            def check(got, stat):
                want = corc.focal_apply(z, k, stat, nthreads=8)
                large = k.size >= 49
                # large windows, mean and sum: float32 sums about a moving shift -- a window of identical cells v (a lake) comes out
                # within 1e-10 of the raster's largest magnitude of v, not bit for bit (var / std / extrema of such a window are exact)
                zf = np.asarray(z, dtype=np.float64)
                amax = float(np.max(np.abs(zf[np.isfinite(zf)]))) if np.isfinite(zf).any() else 0.0
                atol = 1e-10 * amax if large and stat in ("mean", "sum") else 1e-30
                err = close(got, want, rtol=tol(stat), atol=atol)
                if err and stat == "sum" and large:
                    # tests/test_gpu_parity.py, check_window_sum: the reference adds the taps one by one in float32 and carries up to
                    # (n - 1) 2^-24 sum|v| of rounding -- 2.6e-5 of a sum of 441 same-sign taps.  A cell beyond 5e-6 of the reference
                    # passes if BOTH lie where they say they do about the float64 sum: the kernel within 2e-6 of it (+ one rounding of
                    # sum|v|: windows that cancel), the reference within its own bound
                    from scipy import ndimage
                    z64 = z.astype(np.float32).astype(np.float64)
                    zz = np.where(np.isfinite(z64), z64, 0.0)             # (NaN cells are skipped; windows with +-inf are compared as they are)
                    with np.errstate(all="ignore"):
                        exact = ndimage.correlate(zz, np.asarray(k, dtype=np.float64), mode="constant", cval=0.0)
                        sum_abs = ndimage.correlate(np.abs(zz), np.asarray(k, dtype=np.float64), mode="constant", cval=0.0)
                        bound = (k.sum() - 1) * 2.0 ** -24 * sum_abs
                        g, w = host(got).astype(np.float64), want.astype(np.float64)
                        fin = np.isfinite(g) & np.isfinite(w)
                        same = (np.isnan(g) & np.isnan(w)) | (g == w)
                        d = np.abs(g - w)
                        rel_ok = fin & (d <= tol(stat) * np.abs(w) + atol)
                        ours = fin & (np.abs(g - exact) <= 2e-6 * np.abs(exact) + 2.0 ** -24 * sum_abs + atol)
                        theirs = fin & (np.abs(w - exact) <= 1.01 * bound + 1e-30)
                        ok = same | rel_ok | (ours & theirs)
                    if ok.all():
                        return None
                    y, x = np.argwhere(~ok)[0]
                    err += f" [{int((~ok).sum())} cells outside the rounding rule, first ({y}, {x}): exact {exact[y, x]!r} bound {bound[y, x]:.3g}]"
                return err
            if op == "apply":
                stat = str(rng.choice(orc.FOCAL_STATS))
                fn = getattr(focal, "_calc_" + stat)
                return desc + " " + stat, check(focal.apply(agg, k, fn).data, stat)
            if op == "focal_stats":
                got = host(focal.focal_stats(agg, k).data)
                for i, stat in enumerate(orc.FOCAL_STATS):
                    err = check(got[i], stat)
                    if err:
                        return desc + " " + stat, err
                return desc, None
            if op == "convolve":
                w = k / k.sum()
                return desc, close(convolution_2d(agg, w).data, corc.convolve_2d(z, w, nthreads=8), rtol=1e-6, atol=1e-30)
            if not np.isfinite(z.astype(np.float64)).any():
                return desc, None
            try:
                want, zscore = orc.hotspots(z, k)
            except ZeroDivisionError:
                try:
                    focal.hotspots(agg, k)
                except ZeroDivisionError:
                    return desc, None
                return desc, "oracle raised ZeroDivisionError, device path did not"
            # classes may differ where |z| sits within 1e-5 of a threshold (float32 nanmean / nanstd upstream)
            try:
                got = host(focal.hotspots(agg, k).data).copy()
            except ZeroDivisionError:
                # a CONSTANT raster: the device's moments are exact (std == 0 -> the reference's documented error), while
                # numpy's float32 pairwise sums leave the reference a mean one ulp off and a std of rounding noise
                z32 = z.astype(np.float32)
                if float(np.nanmax(z32)) == float(np.nanmin(z32)):
                    return desc + " (constant raster: device raises like the reference's std == 0 branch)", None
                raise
            for t in (1.29, 1.65, 1.96, 2.33, 2.58):
                near = np.abs(np.abs(zscore) - t) < 1e-5
                got[near] = want[near]
            return desc, close(got, want)
        if op in ("ndvi", "evi"):
            b2, b3 = make_raster(rng, shape, dtype), make_raster(rng, shape, dtype)
            if op == "ndvi":
                return desc, close(xs.ndvi(agg, agg_of(b2, backend)).data, orc.normalized_ratio(z, b2), rtol=0)
            return desc, close(xs.evi(agg, agg_of(b2, backend), agg_of(b3, backend)).data, orc.evi(z, b2, b3), rtol=1e-6)
        if op in ("zonal", "crosstab"):
            nz = int(rng.choice([1, 2, 7, 100, 3000]))
            zones = rng.integers(-3, nz, shape).astype(rng.choice([np.int32, np.int64]))
            if rng.random() < 0.5:
                zones = (zones // 1) * 1 + 0
                zones = np.repeat(np.repeat(zones[::8, ::8], 8, 0), 8, 1)[:shape[0], :shape[1]]
            zagg = agg_of(zones, backend)
            if op == "zonal":
                names = ['mean', 'max', 'min', 'sum', 'std', 'var', 'count'] + (['majority'] if rng.random() < 0.3 else [])
                nodata = None if rng.random() < 0.7 else float(z.flat[0]) if z.size and np.isfinite(z.flat[0]) else None
                got = zonal.stats(zagg, agg, stats_funcs=names, nodata_values=nodata)
                want = orc.zonal_stats(zones, z, stats_funcs=names, nodata_values=nodata)
                if list(np.asarray(got['zone'])) != list(np.asarray(want['zone'])):
                    return desc, f"zone ids differ: {len(got)} vs {len(want['zone'])}"
                fin = z[np.isfinite(z)] if z.dtype.kind == 'f' else z
                amax = float(np.abs(fin.astype(np.float64)).max()) if fin.size else 0.0
                for name in names:
                    # (float32 values: the reference reduces in float32, the device in float64 -- a zone of a few nearly
                    #  equal cells has a float32 std / var that is rounding noise of size eps32 * |values|)
                    loose = name in ('std', 'var') or z.dtype == np.float32
                    atol = 0.0
                    if name == 'std':
                        atol = 1e-6 + (2e-7 * amax if z.dtype == np.float32 else 0.0)
                    elif name == 'var':
                        atol = 1e-6 + (2e-7 * amax * amax if z.dtype == np.float32 else 0.0)
                    err = close(np.asarray(got[name], dtype=np.float64), np.asarray(want[name], dtype=np.float64),
                                rtol=1e-5 if loose else 1e-9, atol=atol)
                    if err:
                        return desc + " " + name, err
                return desc, None
            cats = rng.integers(0, int(rng.choice([2, 9, 40])), shape).astype(np.int32)
            got = zonal.crosstab(zagg, agg_of(cats, backend))
            want = orc.crosstab_2d(zones, cats)
            for col in want:
                err = close(np.asarray(got[col]), np.asarray(want[col]))
                if err:
                    return desc + f" col={col}", err
            return desc, None
        if op == "fuse":
            k = circle_kernel(1, 1, int(rng.integers(1, 3)))
            with xs.fuse():
                h, s, c, m = xs.hillshade(agg), xs.slope(agg), xs.curvature(agg), focal.apply(agg, k)
            want_h = orc.hillshade(z) if min(shape) >= 2 else np.full(shape, np.nan)       # (np.gradient refuses 1-cell axes)
            for got, want, atol in ((h, want_h, 1e-6), (s, orc.slope(z, 30.0, 30.0), 0), (c, orc.curvature(z, 30.0), 1e-9),
                                    (m, corc.focal_apply(z, k, 'mean'), 1e-30)):
                err = close(got.data, want, rtol=RTOL, atol=atol)
                if err:
                    return desc + f" {got.name}", err
            return desc, None
        if op == "trim":
            zi = make_raster(rng, shape, np.dtype(np.int32), allow_nan=False) % 3
            zi[: int(rng.integers(0, shape[0] + 1))] = 0
            zi[:, int(rng.integers(0, shape[1] + 1)):] = 0
            vals = (0,) if rng.random() < 0.7 else (0, 1)
            got = zonal._match_bounds(xs.DeviceArray.from_numpy(zi) if backend == "hip" else zi, vals, True)
            want = orc.trim_bounds(zi, vals)
            return desc, None if got == want else f"{got} vs {want}"
        if op == "true_color":
            b2, b3 = make_raster(rng, shape, dtype), make_raster(rng, shape, dtype)
            return desc, close(true_color(agg, agg_of(b2, backend), agg_of(b3, backend)).data, orc.true_color(z, b2, b3))
    return desc, "unknown op"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cells", type=int, default=400000)
    ap.add_argument("--big", action="store_true", help="8-17 Mcell rasters: the banded host pipelines")
    ap.add_argument("--windows", action="store_true", help="large-window statistics on rasters with nodata regions / cliffs / inf")
    args = ap.parse_args()
    global BIG, WINDOWS
    BIG, WINDOWS = args.big, args.windows
    rng = np.random.default_rng(args.seed)
    fails = 0
    counts = {}
    for i in range(args.cases):
        sub = np.random.default_rng(rng.integers(0, 2 ** 62))
        try:
            desc, err = one_case(sub, args.max_cells)
        except Exception as exc:                      # noqa: BLE001 -- report and go on
            desc, err = f"case {i}", "EXCEPTION " + "".join(traceback.format_exception_only(type(exc), exc)).strip() + " @ " + traceback.format_exc().splitlines()[-3].strip()
        counts[desc.split()[0]] = counts.get(desc.split()[0], 0) + 1
        if err:
            fails += 1
            print(f"FAIL [{i}] {desc}: {err}", flush=True)
    print(f"{args.cases} cases, {fails} failures; per operator: {counts}")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
