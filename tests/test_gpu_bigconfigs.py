"""BASELINE configs[3] and [4] at full size on ONE MI355X, through the public API / C ABI:

* 65536 x 65536 float32 DEM (16 GiB; 4.29e9 cells): slope, hillshade, the 5x5 circular focal mean and the fused pass,
  compared with the CPU oracle on row bands around every would-be 8-way shard boundary (rows 8192*k +- 16) -- which
  include the rows where the byte offset crosses 2^31 / 2^32 / 2^33 and the element offset crosses 2^31 -- plus the first
  and the last 64 rows (element offsets up to 2^32 - 1, the raster's bottom edge).
* 32768 x 32768 float32 raster with 1000 int32 zones, blocky and scattered: zonal.stats against an exact host
  reduction (counts bit-exact; min / max exact; sums, means, variances against float64 host sums).

The rasters are built from one generated 4096-row block repeated down the raster with a per-copy offset, so that the
host can regenerate any band without 4e9 random draws while every 4096-row copy still differs (an index that wraps by
2^31 or 2^32 cells lands on different values).  Errors are recorded for the parity report (tests/parity_log.py)."""
import numpy as np
import pytest

import xrspatial_amd as xs
from oracle import c_oracle as corc
from oracle import xrs_oracle as orc
from tests import parity_log, synth
from tests.parity_log import assert_hillshade
from xrspatial_amd import _lib
from xrspatial_amd.convolution import circle_kernel
from xrspatial_amd.focal import apply

pytestmark = pytest.mark.gpu

PERIOD = 4096


class BigRaster:
    """rows x cols float32, content(y, x) = base[y % PERIOD, x] + step * (y // PERIOD), resident in HBM."""

    def __init__(self, rows, cols, step=0.37, nan_frac=0.0):
        self.rows, self.cols, self.step = rows, cols, np.float32(step)
        self.base = synth.asv_dem(PERIOD, cols)
        if nan_frac:
            self.base[np.random.default_rng(11).random(self.base.shape) < nan_frac] = np.nan
        self.dev = xs.DeviceArray((rows, cols), np.float32)
        for b in range(rows // PERIOD):
            block = self.base + self.step * np.float32(b)
            _lib.call("xrs_memcpy_h2d", self.dev.ptr + b * PERIOD * cols * 4, block.ctypes.data, block.nbytes, None)
            _lib.call("xrs_stream_sync", None)

    def host_rows(self, r0, r1):
        out = np.empty((r1 - r0, self.cols), np.float32)
        y = r0
        while y < r1:
            b, o = divmod(y, PERIOD)
            n = min(PERIOD - o, r1 - y)
            out[y - r0:y - r0 + n] = self.base[o:o + n] + self.step * np.float32(b)
            y += n
        return out


@pytest.fixture(scope="module")
def dem64k():
    big = BigRaster(65536, 65536)
    yield big
    del big
    from xrspatial_amd import device
    device.empty_cache()


def _bands64():
    n = 65536
    bands = [(0, 64), (n - 64, n)]
    bands += [(k * 8192 - 16, k * 8192 + 16) for k in range(1, 8)]
    return bands


def test_s64_pipeline_bands_match_oracle(dem64k):
    n = 65536
    agg = xs.DataArray(dem64k.dev, dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
    k5 = circle_kernel(1, 1, 2)
    outs = {'slope': xs.slope(agg).data, 'hillshade': xs.hillshade(agg).data, 'focal_mean_5x5': apply(agg, k5).data}
    for (r0, r1) in _bands64():
        band = dem64k.host_rows(r0, r1)
        first, last = r0 == 0, r1 == n
        # rows of the band whose windows the band holds (the raster's own edges count as held)
        want = {'slope': (corc.slope(band, 1.0, 1.0, nthreads=8), 1), 'hillshade': (orc.hillshade(band), 1),
                'focal_mean_5x5': (corc.focal_apply(band, k5, 'mean', nthreads=8), 2)}
        for name, (w, r) in want.items():
            lo, hi = (0 if first else r), ((r1 - r0) if last else (r1 - r0) - r)
            got = outs[name].rows(r0 + lo, r0 + hi).get()
            atol = 1e-6 if name == 'hillshade' else 0.0       # hillshade ends in (shaded + 1) / 2: absolute float32 accuracy near 0
            parity_log.record("C4 65536^2 (one GPU, bands at 8-way shard boundaries / 2^31, 2^32 offsets / edges)", name, got,
                              w[lo:hi], tol="rtol 1e-5" + (" (|ref| > 1e-6), else atol 1e-6" if atol else ""))
            if name == 'hillshade':                           # relative bar on every cell that is not numerically zero
                assert_hillshade(got, w[lo:hi], f"{name} rows {r0}..{r1}")
            else:
                np.testing.assert_allclose(got, w[lo:hi], rtol=1e-5, atol=atol, equal_nan=True, err_msg=f"{name} rows {r0}..{r1}")
    del outs
    # the fused pass (one read, three products) must reproduce the three stand-alone launches; compare with the oracle too
    with xs.fuse() as scope:
        shade, steep, smooth = xs.hillshade(agg), xs.slope(agg), apply(agg, k5)
    assert scope.launches == 1
    fused = {'slope': steep.data, 'hillshade': shade.data, 'focal_mean_5x5': smooth.data}
    for (r0, r1) in _bands64():
        band = dem64k.host_rows(r0, r1)
        first, last = r0 == 0, r1 == n
        want = {'slope': (corc.slope(band, 1.0, 1.0, nthreads=8), 1), 'hillshade': (orc.hillshade(band), 1),
                'focal_mean_5x5': (corc.focal_apply(band, k5, 'mean', nthreads=8), 2)}
        for name, (w, r) in want.items():
            lo, hi = (0 if first else r), ((r1 - r0) if last else (r1 - r0) - r)
            got = fused[name].rows(r0 + lo, r0 + hi).get()
            atol = 1e-6 if name == 'hillshade' else 0.0
            parity_log.record("C4 65536^2 fused pass (hillshade + slope + 5x5 mean, one read)", name, got, w[lo:hi],
                              tol="rtol 1e-5" + (" (|ref| > 1e-6), else atol 1e-6" if atol else ""))
            if name == 'hillshade':
                assert_hillshade(got, w[lo:hi], f"fused {name} rows {r0}..{r1}")
            else:
                np.testing.assert_allclose(got, w[lo:hi], rtol=1e-5, atol=atol, equal_nan=True,
                                           err_msg=f"fused {name} rows {r0}..{r1}")


def test_s64_nan_frame_and_row_identity(dem64k):
    """Size-independent properties at 65536^2: the NaN frame is exactly one cell wide at every edge, and rows PERIOD apart
    (same base block, different offset) give the same slope -- slope is invariant under a constant offset up to rounding --
    while the hillshade of row y is NOT the hillshade of row y + 32768 wrapped (catches 2^31-cell index wraps)."""
    n = 65536
    agg = xs.DataArray(dem64k.dev, dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
    s = xs.slope(agg).data
    top, bot = s.rows(0, 2).get(), s.rows(n - 2, n).get()
    assert np.isnan(top[0]).all() and np.isnan(bot[1]).all()
    assert not np.isnan(top[1, 1:-1]).any() and not np.isnan(bot[0, 1:-1]).any()
    mid = s.rows(40000, 40002).get()
    assert np.isnan(mid[:, [0, -1]]).all() and not np.isnan(mid[:, 1:-1]).any()
    a, b = s.rows(100, 164).get(), s.rows(100 + 8 * PERIOD, 164 + 8 * PERIOD).get()      # 2^31 cells apart
    np.testing.assert_allclose(a, b, rtol=0, atol=5e-3)             # same gradients (the offset 2.96 only moves roundings)
    m = apply(agg, circle_kernel(1, 1, 2)).data
    ma, mb = m.rows(100, 164).get(), m.rows(100 + 8 * PERIOD, 164 + 8 * PERIOD).get()
    np.testing.assert_allclose(mb - ma, 8 * 0.37, rtol=1e-4)          # the 5x5 mean moves by exactly the offset


# ------------------------------------------------------------------ zonal.stats, 32768^2, 1000 int32 zones
def _zones_band(kind, r0, nrows, cols):
    if kind == 'blocky':
        return synth.block_zones(nrows, cols, n_zones=1000, block=1024, y0=r0)
    # scattered: 8x8-cell patches with hashed ids
    i = (np.arange(r0, r0 + nrows, dtype=np.uint64) >> np.uint64(3))[:, None]
    j = (np.arange(cols, dtype=np.uint64) >> np.uint64(3))[None, :]
    h = (i * np.uint64(2654435761) + j * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xffffffff)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xffffffff)
    h ^= h >> np.uint64(13)
    return (h % np.uint64(1000)).astype(np.int32)


def _patch_partials(big, P=8):
    """Exact host partials of the values over PxP patches (both zone layouts are constant on 8x8 patches): count int64,
    float64 sum and sum of squares, float32 min / max -- one pass over the raster, shared by both layouts."""
    rows, cols = big.rows, big.cols
    shp_out = (rows // P, cols // P)
    out = {'cnt': np.empty(shp_out, np.int64), 'sum': np.empty(shp_out), 'sq': np.empty(shp_out),
           'min': np.empty(shp_out, np.float32), 'max': np.empty(shp_out, np.float32)}
    for r0 in range(0, rows, PERIOD):
        v = big.host_rows(r0, r0 + PERIOD)
        ok = np.isfinite(v)
        v64 = np.where(ok, v, 0.0).astype(np.float64)
        shp = (PERIOD // P, P, cols // P, P)
        dst = slice(r0 // P, (r0 + PERIOD) // P)
        out['cnt'][dst] = ok.reshape(shp).sum(axis=(1, 3))
        out['sum'][dst] = v64.reshape(shp).sum(axis=(1, 3))
        np.multiply(v64, v64, out=v64)
        out['sq'][dst] = v64.reshape(shp).sum(axis=(1, 3))
        out['min'][dst] = np.where(ok, v, np.float32(np.inf)).reshape(shp).min(axis=(1, 3))
        out['max'][dst] = np.where(ok, v, np.float32(-np.inf)).reshape(shp).max(axis=(1, 3))
    return out


def _host_zonal(kind, part, rows, cols, n_zones=1000, P=8):
    zi = np.concatenate([_zones_band(kind, r0, PERIOD, cols)[::P, ::P] for r0 in range(0, rows, PERIOD)]).ravel()
    cnt = np.bincount(zi, weights=part['cnt'].ravel(), minlength=n_zones).astype(np.int64)
    sm = np.bincount(zi, weights=part['sum'].ravel(), minlength=n_zones)
    sq = np.bincount(zi, weights=part['sq'].ravel(), minlength=n_zones)
    order = np.argsort(zi, kind='stable')
    ids, starts = np.unique(zi[order], return_index=True)
    mn, mx = np.full(n_zones, np.inf, np.float32), np.full(n_zones, -np.inf, np.float32)
    mn[ids] = np.minimum.reduceat(part['min'].ravel()[order], starts)
    mx[ids] = np.maximum.reduceat(part['max'].ravel()[order], starts)
    return ids, cnt, sm, sq, mn, mx


@pytest.fixture(scope="module")
def vals32k():
    big = BigRaster(32768, 32768, nan_frac=0.001)
    big.partials = _patch_partials(big)
    yield big
    del big
    from xrspatial_amd import device
    device.empty_cache()


@pytest.mark.parametrize("kind", ["blocky", "scattered"])
def test_s32_zonal_stats_1000_zones(vals32k, kind):
    n = 32768
    zones = xs.DeviceArray((n, n), np.int32)
    for r0 in range(0, n, PERIOD):
        z = _zones_band(kind, r0, PERIOD, n)
        _lib.call("xrs_memcpy_h2d", zones.ptr + r0 * n * 4, z.ctypes.data, z.nbytes, None)
        _lib.call("xrs_stream_sync", None)
    names = ['mean', 'max', 'min', 'sum', 'std', 'var', 'count']
    df = xs.zonal_stats(xs.DataArray(zones, dims=['y', 'x']), xs.DataArray(vals32k.dev, dims=['y', 'x']), stats_funcs=names)
    ids, cnt, sm, sq, mn, mx = _host_zonal(kind, vals32k.partials, n, n)
    assert len(ids) == 1000 and df['zone'].tolist() == ids.tolist()
    # (the comparison is against an EXACT float64 host reduction of the same raster, not against the oracle's float32
    #  pairwise NumPy semantics -- zonal.py:144-163 -- which sit ~1e-6 from it by construction)
    cfg = f"C5 zonal.stats 32768^2, 1000 int32 zones ({kind}), vs an exact float64 host reduction"
    np.testing.assert_array_equal(df['count'].to_numpy().astype(np.int64), cnt)          # bit-exact integer counts
    assert int(cnt.sum()) == int(np.isfinite(vals32k.base).sum()) * (n // PERIOD)
    np.testing.assert_array_equal(df['min'].to_numpy().astype(np.float32), mn)
    np.testing.assert_array_equal(df['max'].to_numpy().astype(np.float32), mx)
    mean = sm / cnt
    var = (sq - sm * sm / cnt) / cnt
    parity_log.record(cfg, 'count', df['count'].to_numpy(), cnt, tol="bit-exact")
    for col, want in (('sum', sm), ('mean', mean), ('var', var), ('std', np.sqrt(var))):
        parity_log.record(cfg, col, df[col].to_numpy(), want, tol="rtol 1e-5 (measured ~1e-12: float64 sums on both sides)")
        np.testing.assert_allclose(df[col].to_numpy(), want, rtol=1e-9 if col in ('sum', 'mean') else 1e-6, err_msg=col)
    parity_log.record(cfg, 'min', df['min'].to_numpy(), mn, tol="bit-exact")
    parity_log.record(cfg, 'max', df['max'].to_numpy(), mx, tol="bit-exact")
