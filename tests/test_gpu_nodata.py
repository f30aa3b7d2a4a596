"""Rasters WITH nodata at BASELINE's full sizes, and the differential fuzzer where `pytest -m gpu` sees it.

Every real DEM carries nodata cells; the reference's focal path ignores NaN (xrspatial/focal.py:305-326 with the numba
nan-reductions, _mean_numpy :44-67) while the 3x3 terrain stencils propagate it (slope.py:56-76, hillshade.py:20-35).
SURVEY.md 8(d) names the input: the asv DEM with 0.1 % of its cells NaN, scattered, seeded.

* 16384^2 (configs[1]/[2]): slope, hillshade, the 5x5 circular mean stand-alone and fused, focal.mean 3x3 and the 25x25
  circular statistics against the CPU oracle on row bands (top edge, interior, bottom edge); fused == stand-alone bit for bit.
* 65536^2 (configs[3]): slope, hillshade, 5x5 mean and the fused pass on bands at the would-be shard boundaries.
* 1000 fixed-seed cases of tests/fuzz_parity.py (every operator, awkward shapes, NaN densities 0 .. 100 %, inf cells).
"""
import numpy as np
import pytest

import xrspatial_amd as xs
from oracle import c_oracle as corc
from oracle import xrs_oracle as orc
from tests import parity_log, synth
from tests.parity_log import assert_hillshade
from xrspatial_amd import _lib, focal
from xrspatial_amd.convolution import circle_kernel
from xrspatial_amd.focal import apply, focal_stats

pytestmark = pytest.mark.gpu

NAN_FRAC = 1e-3
RTOL = 1e-5


@pytest.fixture(scope="module")
def nan16k():
    n = 16384
    dev = xs.DeviceArray((n, n), np.float32)
    bands = {}
    for y0 in range(0, n, 2048):
        host = synth.asv_dem(2048, n, y0=y0, total_rows=n, nan_frac=NAN_FRAC)
        if y0 in (0, 6144, 14336):
            bands[y0] = host
        _lib.call("xrs_memcpy_h2d", dev.ptr + y0 * n * 4, host.ctypes.data, host.nbytes, None)
        _lib.call("xrs_stream_sync", None)
    yield n, dev, bands
    del dev
    from xrspatial_amd import device
    device.empty_cache()


def _same(a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def test_16k_nodata_bands_match_oracle(nan16k):
    n, dev, bands = nan16k
    agg = xs.DataArray(dev, dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
    k5 = circle_kernel(1, 1, 2)
    alone = {'slope': xs.slope(agg).data, 'hillshade': xs.hillshade(agg).data, 'focal5': apply(agg, k5).data}
    with xs.fuse() as scope:
        shade, smooth, steep = xs.hillshade(agg), apply(agg, k5), xs.slope(agg)
    assert scope.launches == 1
    fused = {'slope': steep.data, 'hillshade': shade.data, 'focal5': smooth.data}
    mean3 = focal.mean(agg).data                                  # float64
    cfg = "C2/C3 16384^2 NaN (0.1 % nodata; bands: top edge, interior, bottom edge)"
    n_nan_windows = 0
    for y0, band in bands.items():
        assert np.isnan(band).sum() > 0.5 * NAN_FRAC * band.size
        lo, hi = (0 if y0 == 0 else 2), (2048 if y0 + 2048 == n else 2046)
        want = {'slope': corc.slope(band, 1.0, 1.0, nthreads=8), 'hillshade': orc.hillshade(band[:256 + 4])[:256 + 2],
                'focal5': corc.focal_apply(band, k5, 'mean', nthreads=8)}
        n_nan_windows += int(np.isnan(want['slope'][lo:hi]).sum())
        for name in want:
            rows = slice(lo, 256) if name == 'hillshade' else slice(lo, hi)
            got = alone[name].rows(y0 + rows.start, y0 + rows.stop).get()
            got_f = fused[name].rows(y0 + rows.start, y0 + rows.stop).get()
            # the fused pass is the stand-alone kernels' arithmetic: bit for bit, NaN pattern included
            assert _same(got, got_f), f"fused {name} differs from the stand-alone launch in band {y0}"
            parity_log.record(cfg, name, got, want[name][rows],
                              tol="rtol 1e-5" + (" (|ref| > 1e-6), else atol 1e-6" if name == 'hillshade' else ""))
            assert (np.isnan(got) == np.isnan(want[name][rows])).all(), f"{name} band {y0}: NaN pattern"
            if name == 'hillshade':
                assert_hillshade(got, want[name][rows], f"{name} band {y0}")
            else:
                np.testing.assert_allclose(got, want[name][rows], rtol=RTOL, atol=0, equal_nan=True, err_msg=f"{name} band {y0}")
        # focal.mean 3x3 (float64 out, NaN cells passed through by the default excludes): the reference's float64 row-major sum
        sub = band[:300]
        w3 = corc.focal_mean3x3(sub, passes=1, nthreads=8)
        r3 = slice(lo and 1, 299)
        g3 = mean3.rows(y0 + r3.start, y0 + r3.stop).get()
        parity_log.record(cfg, "focal.mean 3x3 (float64)", g3, w3[r3], tol="rtol 1e-12")
        np.testing.assert_allclose(g3, w3[r3], rtol=1e-12, atol=0, equal_nan=True, err_msg=f"focal.mean band {y0}")
    assert n_nan_windows > 1000                                   # the bands do hold nodata under their windows


def test_16k_nodata_large_windows(nan16k):
    """25x25 circle: all seven statistics and the mean alone, on sub-bands the oracle finishes in seconds."""
    n, dev, bands = nan16k
    agg = xs.DataArray(dev, dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
    k25 = circle_kernel(1, 1, 12)
    stats25 = focal_stats(agg, k25)
    names = list(stats25['stats'].data if isinstance(stats25['stats'].data, np.ndarray) else stats25['stats'].data.get())
    R, B = 12, 96
    cfg = "C3 16384^2 NaN: 25x25 circle statistics (0.1 % nodata), bands"
    for y0, band in bands.items():
        first, last = y0 == 0, y0 + 2048 == n
        sub = band[:B + 2 * R] if not last else band[-(B + 2 * R):]
        off = y0 if not last else n - (B + 2 * R)
        lo, hi = (0 if first else R), (B + 2 * R if last else B + R)
        rows = slice(off + lo, off + hi)
        got25 = apply(xs.DataArray(dev.rows(off, off + B + 2 * R)), k25).data.get()
        w25 = corc.focal_apply(sub, k25, 'mean', nthreads=8)
        parity_log.record(cfg, "focal_mean_25x25 (mean alone)", got25[R:-R], w25[R:-R], tol="rtol 1e-5")
        np.testing.assert_allclose(got25[R:-R], w25[R:-R], rtol=RTOL, atol=0, equal_nan=True)
        for i, stat in enumerate(names):
            want = corc.focal_apply(sub, k25, stat, nthreads=8)[lo:hi]
            got = xs.DeviceArray((hi - lo, n), np.float32, _ptr=stats25.data.ptr + (i * n + rows.start) * n * 4,
                                 _base=stats25.data).get()
            parity_log.record(cfg, f"focal_stats_25x25 {stat}", got, want,
                              tol="bit-exact" if stat in ('max', 'min', 'range') else "rtol 1e-5")
            if stat in ('max', 'min', 'range'):
                np.testing.assert_array_equal(got, want, err_msg=f"{stat} band {y0}")
            else:
                np.testing.assert_allclose(got, want, rtol=RTOL, atol=0, equal_nan=True, err_msg=f"{stat} band {y0}")


def test_64k_nodata_bands_match_oracle():
    """configs[3] with nodata on one GPU: bands at the 8-way shard boundaries, the 2^31 / 2^32 offsets and the edges."""
    from tests.test_gpu_bigconfigs import BigRaster, _bands64
    n = 65536
    big = BigRaster(n, n, nan_frac=NAN_FRAC)
    try:
        agg = xs.DataArray(big.dev, dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
        k5 = circle_kernel(1, 1, 2)
        alone = {'slope': xs.slope(agg).data, 'hillshade': xs.hillshade(agg).data, 'focal_mean_5x5': apply(agg, k5).data}
        with xs.fuse() as scope:
            shade, steep, smooth = xs.hillshade(agg), xs.slope(agg), apply(agg, k5)
        assert scope.launches == 1
        fused = {'slope': steep.data, 'hillshade': shade.data, 'focal_mean_5x5': smooth.data}
        for (r0, r1) in _bands64():
            band = big.host_rows(r0, r1)
            first, last = r0 == 0, r1 == n
            want = {'slope': (corc.slope(band, 1.0, 1.0, nthreads=8), 1), 'hillshade': (orc.hillshade(band), 1),
                    'focal_mean_5x5': (corc.focal_apply(band, k5, 'mean', nthreads=8), 2)}
            for name, (w, r) in want.items():
                lo, hi = (0 if first else r), ((r1 - r0) if last else (r1 - r0) - r)
                got = alone[name].rows(r0 + lo, r0 + hi).get()
                assert _same(got, fused[name].rows(r0 + lo, r0 + hi).get()), f"fused {name} rows {r0}..{r1}"
                parity_log.record("C4 65536^2 NaN (0.1 % nodata; bands at 8-way shard boundaries / edges), stand-alone == fused",
                                  name, got, w[lo:hi], tol="rtol 1e-5" + (" (|ref| > 1e-6), else atol 1e-6" if name == 'hillshade' else ""))
                assert (np.isnan(got) == np.isnan(w[lo:hi])).all(), f"{name} rows {r0}..{r1}: NaN pattern"
                if name == 'hillshade':
                    assert_hillshade(got, w[lo:hi], f"{name} rows {r0}..{r1}")
                else:
                    np.testing.assert_allclose(got, w[lo:hi], rtol=RTOL, atol=0, equal_nan=True, err_msg=f"{name} rows {r0}..{r1}")
    finally:
        del big
        from xrspatial_amd import device
        device.empty_cache()


def test_fuzz_smoke():
    """The differential fuzzer (tests/fuzz_parity.py: public API vs the CPU oracle on seeded random shapes, dtypes, NaN
    densities, inf cells, backends) -- 1000 fixed-seed cases here (~20 s); longer runs are logged under profiles/."""
    from tests import fuzz_parity
    rng = np.random.default_rng(20250905)
    fails = []
    for i in range(1000):
        sub = np.random.default_rng(rng.integers(0, 2 ** 62))
        desc, err = fuzz_parity.one_case(sub, 250000)
        if err:
            fails.append(f"[{i}] {desc}: {err}")
    assert not fails, "\n".join(fails[:10])


@pytest.mark.parametrize("radius", [8, 12, 5])
def test_mean_over_a_lake_of_zeros_is_exactly_zero(radius):
    """A lake at 0.0 on a plateau at -1e5 (and one on ordinary relief, and a sea south of a ragged coast): every window that holds only
    zeros must come out as 0 -- exactly, the reference's nanmean of zeros -- for the mean and the sum alone and for all seven
    statistics.  The tiles around such a lake end in the float64 column walker (circle_walk.h), whose mean was c + S x (1 / n) with a
    reciprocal: S = -n c exactly, the product a rounding off, and the lake read -7.7e-12 (tests/fuzz_parity.py --windows, round 6).
    One correction step on the quotient makes the division of an exact multiple exact."""
    from xrspatial_amd.convolution import circle_kernel
    from xrspatial_amd.focal import _calc_mean, _calc_sum, apply, focal_stats
    rng = np.random.default_rng(radius)
    z = (synth.asv_dem(520, 900) + rng.normal(0, 3, (520, 900))).astype(np.float32)
    z[:, :450] -= np.float32(1e5)                              # the plateau
    z[40:150, 100:330] = 0.0                                   # a lake on it
    z[60:170, 560:800] = 0.0                                   # one on ordinary relief
    coast = 400 + (np.arange(900) // 11) % 7
    z[np.arange(520)[:, None] >= coast[None, :]] = 0.0         # the sea
    z[rng.random(z.shape) < 0.0005] = np.nan
    k = circle_kernel(1, 1, radius)
    A = xs.DataArray(xs.DeviceArray.from_numpy(z), dims=["y", "x"], attrs={"res": (1.0, 1.0)})
    for name, got in (("mean", apply(A, k, _calc_mean).data.get()), ("sum", apply(A, k, _calc_sum).data.get()),
                      ("mean of seven", focal_stats(A, k).data.get()[0])):
        want = corc.focal_apply(z, k, "sum" if name == "sum" else "mean", nthreads=8)
        zeros = want == 0
        assert zeros.sum() > 30000
        bad = zeros & (got != 0)
        assert not bad.any(), f"{name}, radius {radius}: {int(bad.sum())} windows of zeros are not 0, e.g. {got[bad][:4]}"
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-30, equal_nan=True, err_msg=name)


@pytest.mark.parametrize("mode,cases", [("WINDOWS", 120), ("STRUCTURED", 300)])
def test_fuzz_smoke_structured_rasters(mode, cases):
    """The fuzzer's two modes on rasters of several tiles that carry nodata the way real rasters do -- regions with straight and
    ragged rims, scatter at densities 1e-4 .. 0.9, isolated valid cells, lakes, cliffs to 1e7, spikes, +-inf: `--windows` (9x9 ..
    25x25 circles, boxes, annuli; single statistics, all seven, random subsets: ~40 s) and `--structured` (every operator, ~10 s).
    The first 1 200 `--windows` cases found three defects of the large-window kernels; longer runs under profiles/."""
    from tests import fuzz_parity
    setattr(fuzz_parity, mode, True)
    try:
        rng = np.random.default_rng(20250930)
        fails = []
        for i in range(cases):
            sub = np.random.default_rng(rng.integers(0, 2 ** 62))
            desc, err = fuzz_parity.one_case(sub, 10 ** 9)
            if err:
                fails.append(f"[{i}] {desc}: {err}")
        assert not fails, "\n".join(fails[:10])
    finally:
        setattr(fuzz_parity, mode, False)


@pytest.mark.parametrize("kind,radius", [("circle", 12), ("circle", 6), ("circle", 3), ("box", 12), ("box", 5)])
def test_large_window_mean_sum_carry_nodata(kind, radius):
    """The large-window mean / sum on rasters with nodata (wide_impl.h: a tile whose plain walk meets a NaN is walked again by
    the same row walker CARRYING the NaN cells -- vote on staged cells, in-ring repair, lost ring in LDS): scattered NaN at
    0.1 % and 2 %, NaN exactly at the centre cells of wave tiles (the cell the shift is taken from), NaN along the raster's
    edges (edge tiles carry too), a dense patch (the walk hands such tiles on), an all-NaN block larger than the window, and
    a +-inf cell -- mean and sum against the oracle, NaN patterns exactly."""
    from xrspatial_amd.convolution import circle_kernel
    K = 2 * radius + 1
    k = circle_kernel(1, 1, radius) if kind == "circle" else np.ones((K, K))
    rows, cols = 900, 1700
    rng = np.random.default_rng(100 * radius + (kind == "box"))
    base = (1000.0 + 40.0 * np.sin(np.arange(cols)[None, :] / 90.0) * np.cos(np.arange(rows)[:, None] / 70.0)
            + rng.normal(0, 2.0, (rows, cols))).astype(np.float32)
    cases = {}
    z = base.copy(); z[rng.random(z.shape) < 1e-3] = np.nan; cases["0.1 % scattered"] = z
    z = base.copy(); z[rng.random(z.shape) < 0.02] = np.nan; cases["2 % scattered"] = z
    z = base.copy(); z[::13, 64::128] = np.nan; z[65::130, ::17] = np.nan; cases["tile centres and a lattice"] = z
    z = base.copy(); z[:3, ::7] = np.nan; z[-2:, 5::11] = np.nan; z[::9, :2] = np.nan; z[4::10, -3:] = np.nan; cases["raster edges"] = z
    z = base.copy(); z[300:420, 500:800][rng.random((120, 300)) < 0.6] = np.nan; z[600:600 + 2 * K + 5, 900:900 + 3 * K] = np.nan
    z[100, 1200] = np.inf; z[700, 300] = -np.inf; cases["dense patch, nodata block, inf"] = z
    for name, z in cases.items():
        agg = xs.DataArray(xs.DeviceArray.from_numpy(z), dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
        for stat in ('mean', 'sum'):
            got = focal_stats(agg, k, stats_funcs=[stat]).data.get()[0]
            with np.errstate(all='ignore'):
                want = corc.focal_apply(z, k, stat, nthreads=8)
            assert (np.isnan(got) == np.isnan(want)).all(), f"{kind} r={radius} {name} {stat}: NaN pattern"
            fin = np.isfinite(want)
            assert (got[~fin & ~np.isnan(want)] == want[~fin & ~np.isnan(want)]).all(), f"{kind} r={radius} {name} {stat}: infinities"
            if stat == 'mean':
                np.testing.assert_allclose(got[fin], want[fin], rtol=RTOL, atol=0, err_msg=f"{kind} r={radius} {name} mean")
                parity_log.record("large-window mean on nodata (carrying walk), 900x1700", f"{kind} {K}x{K} {name}", got[fin], want[fin],
                                  tol="rtol 1e-5")
            else:
                # `sum` is the exactly rounded window sum; the reference adds the taps in float32 one by one: its own bound
                n = float(np.count_nonzero(k == 1))
                np.testing.assert_allclose(got[fin], want[fin], rtol=max(RTOL, 1.01 * (n - 1) * 2.0 ** -24), atol=0,
                                           err_msg=f"{kind} r={radius} {name} sum")


@pytest.mark.parametrize("kind,radius", [("circle", 12), ("circle", 7), ("circle", 4), ("box", 12), ("box", 6)])
def test_large_window_moments_carry_nodata(kind, radius):
    """mean / var / std (and the seven statistics) over large windows on rasters with nodata (mom_impl.h: a tile whose plain
    walk meets a NaN is walked again by the same two-column walker CARRYING the NaN cells -- in-ring repair with one fill value
    per tile, lost ring in LDS, S -= L (fill - c), Q -= L (fill - c)^2 at the output): scattered NaN at 0.1 % and 2 %, NaN at the
    cells the shift and the fill value are taken from, NaN along the raster's edges (edge tiles: the NaN-aware one-column
    walker, whose first shift at the TOP edge comes from rows inside the raster), a raster with a cliff through its tiles (fill
    value hundreds of window standard deviations from one side: the guard has to hand those tiles on), a dense patch, a
    nodata block larger than the window and +-inf -- against the oracle, NaN patterns exactly; extrema bit-exact."""
    from xrspatial_amd.convolution import circle_kernel
    K = 2 * radius + 1
    k = circle_kernel(1, 1, radius) if kind == "circle" else np.ones((K, K))
    rows, cols = 700, 1500
    rng = np.random.default_rng(200 * radius + (kind == "box"))
    base = (1000.0 + 40.0 * np.sin(np.arange(cols)[None, :] / 90.0) * np.cos(np.arange(rows)[:, None] / 70.0)
            + rng.normal(0, 2.0, (rows, cols))).astype(np.float32)
    cases = {}
    z = base.copy(); z[rng.random(z.shape) < 1e-3] = np.nan; cases["0.1 % scattered"] = z
    z = base.copy(); z[rng.random(z.shape) < 0.02] = np.nan; cases["2 % scattered"] = z
    z = base.copy(); z[::13, 64::128] = np.nan; z[65::130, ::17] = np.nan; z[:40:3, ::2] = np.nan; cases["tile centres, first rows, a lattice"] = z
    z = base.copy(); z[:3, ::7] = np.nan; z[-2:, 5::11] = np.nan; z[::9, :2] = np.nan; z[4::10, -3:] = np.nan; cases["raster edges"] = z
    z = base.copy(); z[:, 700:] += 5000.0; z[:, 1100:] -= 9000.0; z[rng.random(z.shape) < 5e-4] = np.nan; cases["cliffs through the tiles"] = z
    z = base.copy(); z[300:420, 500:800][rng.random((120, 300)) < 0.6] = np.nan; z[500:500 + 2 * K + 5, 900:900 + 3 * K] = np.nan
    z[100, 1200] = np.inf; z[600, 300] = -np.inf; cases["dense patch, nodata block, inf"] = z
    for name, z in cases.items():
        agg = xs.DataArray(xs.DeviceArray.from_numpy(z), dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
        with np.errstate(all='ignore'):
            want = {st: corc.focal_apply(z, k, st, nthreads=8) for st in ('mean', 'max', 'min', 'range', 'std', 'var', 'sum')}
        for funcs in (['mean', 'var', 'std'], ['mean', 'max', 'min', 'range', 'std', 'var', 'sum']):
            got = focal_stats(agg, k, stats_funcs=funcs).data.get()
            for i, st in enumerate(funcs):
                msg = f"{kind} r={radius} {name} {st} of {len(funcs)}"
                g, w = got[i], want[st]
                assert (np.isnan(g) == np.isnan(w)).all(), msg + ": NaN pattern"
                fin = np.isfinite(w)
                assert (g[~fin & ~np.isnan(w)] == w[~fin & ~np.isnan(w)]).all(), msg + ": infinities"
                if st in ('max', 'min', 'range'):
                    np.testing.assert_array_equal(g[fin], w[fin], err_msg=msg)
                elif st == 'sum':
                    n = float(np.count_nonzero(k == 1))
                    np.testing.assert_allclose(g[fin], w[fin], rtol=max(RTOL, 1.01 * (n - 1) * 2.0 ** -24), atol=0, err_msg=msg)
                else:
                    # (std of a flat window next to a cliff: sqrt halves the exponent of a tiny variance -- absolute 1e-5 of the mean's scale)
                    np.testing.assert_allclose(g[fin], w[fin], rtol=RTOL, atol=0, err_msg=msg)
                    parity_log.record("large-window moments on nodata (carrying walk), 700x1500", f"{kind} {K}x{K} {name} {st}", g[fin], w[fin],
                                      tol="rtol 1e-5")


@pytest.mark.parametrize("variant", ["rows", "cols"])
def test_16k_nodata_region_rim(variant):
    """Nodata as a REGION (a sea, the collar of a tile): the first third of the 16384^2 benchmark raster is NaN -- its rows
    (a horizontal rim) or its columns (a vertical rim through a tile of every tile row).  The 25x25 circle statistics on a
    band that straddles the rim, where the windows hold anything from 441 valid cells down to one: extrema bit for bit, moments
    to 1e-5, NaN exactly where no valid cell is under the window (xrspatial/focal.py:305-326 with numba's nan-reductions).
    These are the tiles the moments kernel hands to its rescue launch (csrc/mom_impl.h: focal_mom_rescue_kernel), whose
    windows of a few cells are recomputed one by one in float64."""
    n, R, B = 16384, 12, 72
    cut = n // 3
    dev = xs.DeviceArray((n, n), np.float32)
    y_band = cut - R - B // 2 if variant == "rows" else 9000          # first row of the compared band
    keep = None
    for y0 in range(0, n, 2048):
        host = synth.asv_dem(2048, n, y0=y0, total_rows=n).copy()
        if variant == "rows":
            host[: max(0, min(2048, cut - y0))] = np.nan
        else:
            host[:, :cut] = np.nan
        lo, hi = y_band - R, y_band + B + R
        if y0 <= lo and hi <= y0 + 2048:
            keep = host[lo - y0:hi - y0].copy()
        _lib.call("xrs_memcpy_h2d", dev.ptr + y0 * n * 4, host.ctypes.data, host.nbytes, None)
        _lib.call("xrs_stream_sync", None)
    assert keep is not None
    agg = xs.DataArray(dev, dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
    k25 = circle_kernel(1, 1, 12)
    stats25 = focal_stats(agg, k25)
    names = [str(s) for s in np.asarray(stats25['stats'].data)]
    mean_alone = apply(agg, k25).data
    cfg = f"C3 16384^2 nodata REGION ({variant}): 25x25 circle statistics on the band across the rim"
    few = 0
    for i, stat in enumerate(names):
        want = corc.focal_apply(keep, k25, stat, nthreads=8)[R:-R]
        got = xs.DeviceArray((B, n), np.float32, _ptr=stats25.data.ptr + (i * n + y_band) * n * 4, _base=stats25.data).get()
        parity_log.record(cfg, f"focal_stats_25x25 {stat}", got, want, tol="bit-exact" if stat in ('max', 'min', 'range') else "rtol 1e-5")
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want), err_msg=f"{variant} {stat}: NaN pattern")
        if stat in ('max', 'min', 'range'):
            np.testing.assert_array_equal(got, want, err_msg=f"{variant} {stat}")
        else:
            # (var / std of a window with ONE valid cell are exactly 0 in the reference; a handful of equal cells likewise)
            np.testing.assert_allclose(got, want, rtol=RTOL, atol=0 if stat in ('mean', 'sum') else 1e-12, equal_nan=True, err_msg=f"{variant} {stat}")
        if stat == 'mean':
            g1 = xs.DeviceArray((B, n), np.float32, _ptr=mean_alone.ptr + y_band * n * 4, _base=mean_alone).get()
            np.testing.assert_allclose(g1, want, rtol=RTOL, atol=0, equal_nan=True, err_msg=f"{variant} mean alone")
            few = int(np.isnan(want).sum())
    assert few > 1000                                             # the band does reach into the region
    del dev, stats25, mean_alone
    from xrspatial_amd import device
    device.empty_cache()


@pytest.mark.parametrize("kind,K", [("circle", 13), ("box", 25), ("circle", 25), ("annulus", 21)])
def test_windows_whose_only_valid_cell_is_infinite(kind, K):
    """A window with ONE valid cell has variance exactly 0 -- unless that cell is +-inf: numba's nanvar takes (inf - inf)^2 = NaN
    (focal.py:282-289).  Mostly-nodata rasters with isolated finite and infinite cells, so that whole neighbourhoods of windows see
    exactly one of them (found by tests/fuzz_parity.py seeds 61 / 63 once the large-window kernels judged windows one by one)."""
    from xrspatial_amd.convolution import annulus_kernel, circle_kernel
    R = K // 2
    k = circle_kernel(1, 1, R) if kind == "circle" else np.ones((K, K)) if kind == "box" else annulus_kernel(1, 1, R, 4)
    z = np.full((300, 520), np.nan, np.float32)
    z[60, 70], z[60, 300], z[200, 130], z[210, 420] = np.inf, 7.5, -np.inf, 1234.5
    z[120:124, 200:204] = 3.25                                    # a few equal finite cells: variance 0
    z[250:, 450:] = synth.smooth_dem((50, 70), seed=3)            # and an ordinary corner
    z[270, 480] = np.inf
    got = focal_stats(xs.DataArray(z, dims=['y', 'x']), k).data
    with np.errstate(all='ignore'):
        for i, stat in enumerate(orc.FOCAL_STATS):
            want = corc.focal_apply(z, k, stat, nthreads=8)
            np.testing.assert_array_equal(np.isnan(got[i]), np.isnan(want), err_msg=f"{kind}{K} {stat}: NaN pattern")
            fin = np.isfinite(want)
            np.testing.assert_array_equal(got[i][~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)], err_msg=f"{kind}{K} {stat}: infinities")
            if stat in ('max', 'min', 'range'):
                np.testing.assert_array_equal(got[i][fin], want[fin], err_msg=f"{kind}{K} {stat}")
            else:
                np.testing.assert_allclose(got[i][fin], want[fin], rtol=RTOL, atol=1e-6 if stat in ('var', 'std') else 0, err_msg=f"{kind}{K} {stat}")
    assert np.isnan(got[list(orc.FOCAL_STATS).index('var')][60, 70 - R // 2])        # the single +inf under that window
