"""The oracle against the reference's own code, EXECUTED (tests/golden/make_reference_exec.py).

Two tiers, both CPU-only:
  * live (skipped where /root/reference is not mounted, e.g. the GPU box): the reference's functions are lifted from
    their files with `ast` and run in this process; the oracle must equal them BIT FOR BIT, and the committed fixture
    must be what the script writes today;
  * fixture (runs everywhere): the oracle against tests/golden/reference_exec.npz.  Bit for bit in the image the
    fixture was written in (= wherever the reference is mounted).  On another CPU NumPy may dispatch its SIMD
    sin / cos / arctan and np.argsort's unstable sort differently (an ulp in a float64 intermediate; another
    permutation of equal zone ids, hence another float32 summation order): there the float results are held to 1e-6
    and everything integral (counts, extrema, majority, NaN patterns) stays exact.

Rows pinned this way: f3 geodesic slope / aspect, a3 hillshade, a13 zonal.stats, f4 crosstab,
a9 kernel builders, a10 / f1 the float32 normalized ratio and sipi.  The rows whose reference loops mix Python
literals with float32 cells cannot be pinned like this (see the generator's docstring): they stay vector-pinned.
"""
import itertools
import json
import os

import numpy as np
import pytest

from oracle import xrs_oracle as orc
from tests.golden import make_reference_exec as rx

needs_reference = pytest.mark.skipif(not rx.have_reference(), reason="/root/reference is not mounted")
ORDER_DEPENDENT = {"mean", "sum", "std", "var"}


@pytest.fixture(scope="module")
def fixture():
    assert os.path.exists(rx.OUT), "tests/golden/reference_exec.npz is missing: run tests/golden/make_reference_exec.py"
    return rx.load()


def same(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.dtype.kind in "fc" or b.dtype.kind in "fc":
        np.testing.assert_array_equal(np.isnan(a), np.isnan(b), err_msg=what)
        ok = (a == b) | (np.isnan(a) & np.isnan(b))
        assert ok.all(), f"{what}: {np.count_nonzero(~ok)} cells differ, first {a[~ok][:3]} vs {b[~ok][:3]}"
    else:
        np.testing.assert_array_equal(a, b, err_msg=what)


def close(a, b, what=""):
    """Fixture tier: exact where the fixture was generated, 1e-6 elsewhere (module docstring)."""
    if rx.have_reference():
        return same(a, b, what)
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=1e-6, atol=1e-6,
                               equal_nan=True, err_msg=what)


def names(store, prefix):
    return sorted({k.split("/")[1] for k in store if k.startswith(prefix + "/")}, key=lambda s: (len(s), s))


# ---------------------------------------------------------------------------------------------------------------------
# fixture tier
# ---------------------------------------------------------------------------------------------------------------------
def test_fixture_geodesic(fixture):
    cases = names(fixture, "geo")
    assert len(cases) >= 30
    for n in cases:
        p = f"geo/{n}"
        elev, lat, lon, zf = fixture[p + "/elev"], fixture[p + "/lat"], fixture[p + "/lon"], float(fixture[p + "/z_factor"])
        assert fixture[p + "/slope"].dtype == np.float32
        close(orc.geodesic_slope(elev, lat, lon, zf), fixture[p + "/slope"], p + " slope")
        close(orc.geodesic_aspect(elev, lat, lon, zf), fixture[p + "/aspect"], p + " aspect")


def test_fixture_hillshade(fixture):
    for n in names(fixture, "hs"):
        p = f"hs/{n}"
        want = fixture[p + "/out"]
        got = orc.hillshade(fixture[p + "/data"], float(fixture[p + "/azimuth"]), float(fixture[p + "/altitude"]))
        assert got.dtype == want.dtype
        close(got, want, p)


def _check_zonal_table(got, want_cols, exact):
    assert list(got) == [c for c, _ in want_cols]
    for c, want in want_cols:
        g = np.asarray(got[c])
        if exact or c not in ORDER_DEPENDENT:
            same(g, want, f"column {c}")
        else:
            np.testing.assert_allclose(g, want, rtol=1e-6, equal_nan=True, err_msg=f"column {c}")


def _zonal_args(store, p):
    return (store[p + "/zones"], store[p + "/values"], json.loads(str(store[p + "/zone_ids"])),
            json.loads(str(store[p + "/nodata"])))


def test_fixture_zonal_stats(fixture):
    for n in names(fixture, "zs"):
        p = f"zs/{n}"
        zones, values, zone_ids, nodata = _zonal_args(fixture, p)
        with np.errstate(all="ignore"):
            got = orc.zonal_stats(zones, values, zone_ids=zone_ids, nodata_values=nodata)
            arr = orc.zonal_stats(zones, values, zone_ids=zone_ids, nodata_values=nodata, return_type='array')
        _check_zonal_table(got, rx.table(fixture, p + "/table"), exact=False)
        np.testing.assert_allclose(arr, fixture[p + "/array"], rtol=1e-6, equal_nan=True, err_msg=p)
        same(arr[[1, 2, 6, 7]], fixture[p + "/array"][[1, 2, 6, 7]], p + " max/min/count/majority planes")


def _check_crosstab(got, want_cols, what):
    got_cols = list(got)
    assert [float(c) if not isinstance(c, str) else c for c in got_cols] == [c for c, _ in want_cols], what
    for gc, (c, want) in zip(got_cols, want_cols):
        same(np.asarray(got[gc], dtype=np.asarray(want).dtype), want, f"{what} column {c}")


def test_fixture_crosstab_2d(fixture):
    for n in names(fixture, "ct"):
        p = f"ct/{n}"
        a = json.loads(str(fixture[p + "/args"]))
        with np.errstate(all="ignore"):
            got = orc.crosstab_2d(fixture[p + "/zones"], fixture[p + "/values"], a["zone_ids"], a["cat_ids"], a["nodata"], a["agg"])
        _check_crosstab(got, rx.table(fixture, p + "/table"), p)


def test_fixture_band_ratios(fixture):
    for n in names(fixture, "ms"):
        p = f"ms/{n}"
        a, b, c = fixture[p + "/a"], fixture[p + "/b"], fixture[p + "/c"]
        with np.errstate(all="ignore"):
            same(orc.normalized_ratio(a, b), fixture[p + "/normalized_ratio"], p + " normalized ratio")
            same(orc.sipi(a, b, c), fixture[p + "/sipi"], p + " sipi")


# ---------------------------------------------------------------------------------------------------------------------
# live tier: the reference runs in this process
# ---------------------------------------------------------------------------------------------------------------------
@needs_reference
def test_live_fixture_is_reproducible(fixture):
    """The committed fixture is what the generator writes from the mounted reference today (bit for bit; the zonal float
    columns too -- same machine, same NumPy, same permutation)."""
    fresh = rx.run_all()
    assert sorted(fresh) == sorted(fixture)
    for k in fresh:
        a, b = np.asarray(fresh[k]), fixture[k]
        assert a.dtype == b.dtype, k
        if a.dtype.kind in "US":
            assert str(a) == str(b), k
        else:
            same(a, b, k)


@needs_reference
def test_live_geodesic_random():
    """Fresh random grids (not in the fixture): the oracle equals the executed reference bit for bit."""
    _, ref_slope, ref_aspect = rx.ref_geodesic()
    rng = np.random.default_rng(99)
    for trial in range(10):
        H, W = int(rng.integers(3, 9)), int(rng.integers(3, 9))
        lat0, lon0 = float(rng.uniform(-86, 86)), float(rng.uniform(-180, 359))
        cell = float(rng.choice([1 / 3600, 1 / 1200, 0.01]))
        LAT, LON = np.meshgrid(lat0 - cell * np.arange(H), lon0 + cell * np.arange(W), indexing="ij")
        elev = rng.normal(800, 150, (H, W)).astype([np.float32, np.float64][trial % 2])
        if trial % 3 == 0:
            elev[rng.integers(0, H), rng.integers(0, W)] = np.nan
        zf = [1.0, 0.3048][trial % 2]
        with np.errstate(all="ignore"):
            same(orc.geodesic_slope(elev, LAT, LON, zf), ref_slope(elev, LAT, LON, orc.WGS84_A2, orc.WGS84_B2, zf), f"slope {trial}")
            same(orc.geodesic_aspect(elev, LAT, LON, zf), ref_aspect(elev, LAT, LON, orc.WGS84_A2, orc.WGS84_B2, zf), f"aspect {trial}")


@needs_reference
def test_live_hillshade_and_zonal_random():
    run_numpy = rx.ref_hillshade()
    zn = rx.ref_zonal()
    rng = np.random.default_rng(5)
    with np.errstate(all="ignore"):
        for trial in range(20):
            z = rng.normal(0, 50, (int(rng.integers(3, 40)), int(rng.integers(3, 40)))).astype([np.float32, np.float64, np.int16][trial % 3])
            az, alt = float(rng.uniform(0, 360)), float(rng.uniform(0, 90))
            same(orc.hillshade(z, az, alt), run_numpy(z, az, alt), f"hillshade {trial}")
        for trial in range(30):
            H, W = int(rng.integers(2, 40)), int(rng.integers(2, 40))
            zones = rng.integers(0, 7, (H, W)).astype([np.int32, np.float64][trial % 2])
            if trial % 2:
                zones[rng.random((H, W)) < 0.1] = np.nan
            values = rng.choice(rng.normal(0, 10, 12), (H, W)).astype([np.float32, np.float64][trial % 3 == 0])
            values[rng.random((H, W)) < 0.1] = np.nan
            nodata = [None, float(values.ravel()[1])][trial % 2] if np.isfinite(values.ravel()[1]) else None
            zone_ids = [None, [1, 3, 99]][trial % 4 == 3]
            df = zn["_stats_numpy"](zones, values, zone_ids, dict(zn["_DEFAULT_STATS"]), nodata, "pandas.DataFrame")
            got = orc.zonal_stats(zones, values, zone_ids=zone_ids, nodata_values=nodata)
            _check_zonal_table(got, [(c, df[c].values) for c in df.columns], exact=True)
            same(orc.zonal_stats(zones, values, zone_ids=zone_ids, nodata_values=nodata, return_type='array'),
                 zn["_stats_numpy"](zones, values, zone_ids, dict(zn["_DEFAULT_STATS"]), nodata, "xarray.DataArray"), f"zonal array {trial}")


@needs_reference
def test_live_kernel_builders():
    """xrspatial_amd.convolution's builders (host Python, row a9) against the reference's, argument for argument: values,
    shapes, dtypes and exception types."""
    from xrspatial_amd import convolution as mine
    ref = rx.ref_kernels()

    def outcome(fn, *args):
        try:
            r = fn(*args)
            return ("ok", np.asarray(r).dtype, np.asarray(r).shape, np.asarray(r).tobytes())     # (bytes: NaN taps compare equal)
        except Exception as e:                                        # noqa: BLE001 (the TYPE is what is compared)
            return ("raise", type(e).__name__)

    sizes = [0.5, 1, 1.0, 2, 3.7, 10, 30.0]
    radii = [0.4, 1, 2, 2.5, 3, 12, 30, 45.5, "1", "3m", "2 km", "10ft", "0.01mile", "-1", "abc", "3 parsec", 0, -2]
    n = 0
    for cx, cy, r in itertools.product(sizes, sizes, radii):
        assert outcome(mine.circle_kernel, cx, cy, r) == outcome(ref["circle_kernel"], cx, cy, r), (cx, cy, r)
        n += 1
    for cx, cy, ro, ri in itertools.product(sizes[:5], sizes[:5], radii[:9] + ["5m", "-1"], [0, 0.5, 1, 2, "1m", 40, "x"]):
        assert outcome(mine.annulus_kernel, cx, cy, ro, ri) == outcome(ref["annulus_kernel"], cx, cy, ro, ri), (cx, cy, ro, ri)
        n += 1
    customs = [np.ones((3, 3)), np.ones((2, 3)), np.ones((3, 4)), np.array([[0, 1, 0], [1, 2, 1], [0, 1, 0]]), [[1, 1, 1]] * 3,
               np.ones((5, 1)), np.ones((1, 1)), np.array([[1, np.nan, 1]] * 3), np.ones(3), "kernel", np.ones((3, 3), dtype=bool)]
    for k in customs:
        assert outcome(mine.custom_kernel, k) == outcome(ref["custom_kernel"], k), k
        n += 1
    for s in ["1", "3m", "2 km", "10ft", "1.5 miles", "7 mls", " 4 m", "1e3", "", "m", "-3m", "0", "1 2 3"]:
        assert outcome(mine._get_distance, s) == outcome(ref["_get_distance"], s), s
        n += 1
    assert n > 1000


@needs_reference
def test_live_host_resolution_helpers():
    """a14: the package's resolution / validation helpers against the reference's, on the same DataArray objects: `res` as tuple,
    list, ndarray, scalar, malformed; coordinates only (ascending, descending, non-uniform); units; named dimensions; shapes
    and backends that must be refused -- values and exception types."""
    import xrspatial_amd as xs
    from xrspatial_amd import convolution as mine_conv, utils as mine
    ref = rx.ref_host_utils()

    def outcome(fn, *args, **kw):
        try:
            r = fn(*args, **kw)
            return ("ok", tuple(float(v) for v in r) if r is not None else None)
        except Exception as e:                                        # noqa: BLE001 (the TYPE is what is compared)
            return ("raise", type(e).__name__)

    data = np.zeros((7, 11), np.float32)
    rasters = []
    for res in ((0.5, 0.5), [2, 3.5], np.array([1.0, 2.0]), 30, 12.5, (1, 2, 3), "10", None, (np.float32(2.0), 2.0), ("a", 1.0)):
        rasters.append(xs.DataArray(data, dims=['y', 'x'], attrs={} if res is None else {'res': res},
                                    coords={'y': np.linspace(70, 10, 7), 'x': np.linspace(-5, 45, 11)}))
    rasters.append(xs.DataArray(data, dims=['lat', 'lon'], coords={'lat': np.array([0, 1, 2, 4, 7, 11, 16.0]), 'lon': np.arange(11.0) * 3}))
    rasters.append(xs.DataArray(data, dims=['y', 'x'], attrs={'unit': 'km'}, coords={'y': np.arange(7.0), 'x': np.arange(11.0)}))
    rasters.append(xs.DataArray(data, dims=['y', 'x'], attrs={'unit': 'ft', 'res': (3, 3)}))
    rasters.append(xs.DataArray(data, dims=['y', 'x'], attrs={'unit': 'parsec', 'res': (3, 3)}))
    rasters.append(xs.DataArray(np.zeros((1, 1), np.float32), dims=['y', 'x'], coords={'y': [1.0], 'x': [2.0]}))
    n = 0
    with np.errstate(all="ignore"):
        for r in rasters:
            for name, kw in (("get_dataarray_resolution", {}), ("calc_res", {}), ("calc_res", {'xdim': r.dims[-1], 'ydim': r.dims[-2]})):
                assert outcome(getattr(mine, name), r, **kw) == outcome(ref[name], r, **kw), (name, r.attrs, r.dims)
                n += 1
            assert outcome(mine_conv.calc_cellsize, r) == outcome(ref["calc_cellsize"], r), ("calc_cellsize", r.attrs)
            got, want = outcome(mine.get_xy_range, r), outcome(ref["get_xy_range"], r)
            assert got[0] == want[0] and (got[0] == "raise" or np.allclose(np.ravel(got[1]), np.ravel(want[1]))) or got == want
            n += 2
    other = xs.DataArray(np.zeros((7, 12), np.float32), dims=['y', 'x'])
    for args in ((rasters[0],), (rasters[0], other), (rasters[0], rasters[1]), (rasters[0], rasters[1], other)):
        assert outcome(lambda *a: mine.validate_arrays(*a), *args) == outcome(lambda *a: ref["validate_arrays"](*a), *args)
        n += 1
    assert n >= 60


@needs_reference
def test_live_dataset_adapters():
    """a14: `supports_dataset` / `supports_dataset_bands` (dataset_support.py) lifted from the reference and applied to the same
    probe functions as the package's decorators, on the package's Dataset: results per variable, the `name` pass-through, band
    aliases, what is refused and how."""
    import functools
    import inspect
    import xrspatial_amd as xs
    from xrspatial_amd import _xr, dataset_support as mine
    ref = rx.lift("dataset_support.py", ["supports_dataset", "supports_dataset_bands"], {"functools": functools, "inspect": inspect, "xr": _xr})

    def one(agg, factor=2.0, name='one'):
        return xs.DataArray(np.asarray(agg.data) * factor, dims=agg.dims, name=name)

    def no_name(agg, offset=0.0):
        return xs.DataArray(np.asarray(agg.data) + offset, dims=agg.dims)

    def bands(nir_agg, red_agg, soil=1.0, name='idx'):
        return xs.DataArray(np.asarray(nir_agg.data) - soil * np.asarray(red_agg.data), dims=nir_agg.dims, name=name)

    a = xs.DataArray(np.arange(6.0).reshape(2, 3), dims=['y', 'x'])
    b = xs.DataArray(np.arange(6.0).reshape(2, 3) * 10, dims=['y', 'x'])
    ds = xs.Dataset({'b4': a, 'b8': b}, attrs={'crs': 'EPSG:4326'})

    def outcome(fn, *args, **kw):
        try:
            r = fn(*args, **kw)
        except Exception as e:                                        # noqa: BLE001
            return ("raise", type(e).__name__, str(e))
        if isinstance(r, _xr.Dataset):
            return ("dataset", dict(r.attrs), {k: (np.asarray(r[k].data).tolist(), r[k].name) for k in r.data_vars})
        return ("array", np.asarray(r.data).tolist(), r.name)

    for fn in (one, no_name):
        m, r = mine.supports_dataset(fn), ref["supports_dataset"](fn)
        assert m.__name__ == r.__name__ == fn.__name__
        for args, kw in (((ds,), {}), ((ds, 3.0), {}), ((a,), {}), ((a,), {'name': 'x'} if fn is one else {'offset': 1.0}), ((ds,), {'bogus': 1})):
            assert outcome(m, *args, **kw) == outcome(r, *args, **kw), (fn.__name__, kw)
    m = mine.supports_dataset_bands(nir='nir_agg', red='red_agg')(bands)
    r = ref["supports_dataset_bands"](nir='nir_agg', red='red_agg')(bands)
    for args, kw in (((ds,), {'nir': 'b8', 'red': 'b4'}), ((ds,), {'nir': 'b8', 'red': 'b4', 'soil': 0.5, 'name': 'savi'}), ((ds,), {'nir': 'b8'}),
                     ((ds,), {'nir': 'b8', 'red': 'b9'}), ((b, a), {}), ((b, a), {'soil': 2.0}), ((ds,), {'nir': 'b8', 'red': 'b4', 'bogus': 1})):
        assert outcome(m, *args, **kw) == outcome(r, *args, **kw), kw
