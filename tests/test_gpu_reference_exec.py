"""The HIP path against outputs of the reference's OWN code (tests/golden/reference_exec.npz, written by
tests/golden/make_reference_exec.py from functions lifted out of /root/reference and executed there).  No oracle in
between: what is compared is the MI355X result and what xrspatial's NumPy / Numba-CPU functions returned.

Tolerances: north_star's 1e-5 relative for floating point (the tests state where they are tighter); counts, extrema,
majority, category tables and NaN patterns exact."""
import json

import numpy as np
import pytest

import xrspatial_amd as xs
from tests import parity_log
from tests.golden import make_reference_exec as rx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fixture():
    return rx.load()


def names(store, prefix):
    return sorted({k.split("/")[1] for k in store if k.startswith(prefix + "/")}, key=lambda s: (len(s), s))


def host(a):
    return a.get() if isinstance(a, xs.DeviceArray) else np.asarray(a)


def _geo_raster(elev, LAT, LON, two_d, backend):
    if two_d:
        agg = xs.DataArray(elev, dims=['y', 'x'], coords={'lat': xs.DataArray(LAT, dims=['y', 'x']),
                                                          'lon': xs.DataArray(LON, dims=['y', 'x'])})
    else:
        agg = xs.DataArray(elev, dims=['lat', 'lon'], coords={'lat': LAT[:, 0].copy(), 'lon': LON[0].copy()})
    if backend == 'hip':
        agg.data = xs.DeviceArray.from_numpy(elev)
    return agg


@pytest.mark.parametrize("backend", ["numpy", "hip"])
def test_geodesic_equals_executed_reference(fixture, backend):
    """f3: xrspatial/geodesic.py:174-229 (`_cpu_geodesic_slope/_aspect`) executed -> slope / aspect in degrees, float32.
    1e-5 relative (+1e-5 degrees absolute: a flat raster's slope is ~1e-6 degrees of float64 noise in the reference
    itself); the -1 of flat aspect, the NaN ring and the NaN neighbourhoods exactly."""
    cases = names(fixture, "geo")
    for n in cases:
        p = f"geo/{n}"
        elev, LAT, LON, zf = fixture[p + "/elev"], fixture[p + "/lat"], fixture[p + "/lon"], float(fixture[p + "/z_factor"])
        z_unit = {1.0: 'meter', 0.3048: 'foot'}[zf]
        for two_d in (False, True):
            agg = _geo_raster(elev, LAT, LON, two_d, backend)
            for fn, key in ((xs.slope, "slope"), (xs.aspect, "aspect")):
                got = host(fn(agg, method='geodesic', z_unit=z_unit).data)
                want = fixture[f"{p}/{key}"]
                assert got.dtype == np.float32 and got.shape == want.shape
                np.testing.assert_array_equal(np.isnan(got), np.isnan(want), err_msg=f"{p} {key}")
                flat = want == -1.0
                np.testing.assert_array_equal(got == -1.0, flat, err_msg=f"{p} {key}: flat cells")
                if key == "aspect":
                    # a bearing is a point on a circle: 359.99999 and 0.00001 are 2e-5 degrees apart
                    d = np.abs(got.astype(np.float64) - want)
                    d = np.minimum(d, 360.0 - d)
                    ok = ~(d > 1e-5 * np.maximum(np.abs(want), 1.0))
                    assert ok.all(), f"{p} aspect: {got[~ok][:3]} vs {want[~ok][:3]}"
                else:
                    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5, equal_nan=True, err_msg=f"{p} {key}")
                if not two_d:
                    parity_log.record("f3 geodesic: HIP vs the reference's own CPU kernels, executed", key, got, want,
                                      tol="rtol 1e-5 + 1e-5 deg")


def test_hillshade_equals_executed_reference(fixture):
    """a3: xrspatial/hillshade.py:20-35 (`_run_numpy`) executed."""
    from tests.parity_log import assert_hillshade
    for n in names(fixture, "hs"):
        p = f"hs/{n}"
        z = fixture[p + "/data"]
        agg = xs.DataArray(z, dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
        got = host(xs.hillshade(agg, azimuth=float(fixture[p + "/azimuth"]), angle_altitude=float(fixture[p + "/altitude"])).data)
        want = fixture[p + "/out"]
        assert got.dtype == want.dtype
        assert_hillshade(got, want, f"a3 hillshade vs executed reference {p}")


def _args(store, p):
    return (store[p + "/zones"], store[p + "/values"], json.loads(str(store[p + "/zone_ids"])), json.loads(str(store[p + "/nodata"])))


@pytest.mark.parametrize("backend", ["numpy", "hip"])
def test_zonal_stats_equals_executed_reference(fixture, backend):
    """a13: xrspatial/zonal.py:280-332 (`_stats_numpy` with the default eight statistics incl. majority) executed:
    the DataFrame and the back-projected (stats, y, x) array."""
    for n in names(fixture, "zs"):
        p = f"zs/{n}"
        zones, values, zone_ids, nodata = _args(fixture, p)
        zagg, vagg = xs.DataArray(zones, dims=['y', 'x']), xs.DataArray(values, dims=['y', 'x'])
        if backend == 'hip':
            zagg.data, vagg.data = xs.DeviceArray.from_numpy(zones), xs.DeviceArray.from_numpy(values)
        kw = {} if nodata is None else {'nodata_values': nodata}
        df = xs.zonal.stats(zagg, vagg, zone_ids=zone_ids, **kw)
        want = rx.table(fixture, p + "/table")
        assert list(df.columns) == [c for c, _ in want], p
        for c, col in want:
            got = np.asarray(df[c].values)
            if c in ("zone", "count", "max", "min", "majority"):
                np.testing.assert_array_equal(np.asarray(got, dtype=np.float64), np.asarray(col, dtype=np.float64), err_msg=f"{p} {c}")
            else:
                # float32 values: the reference adds in float32 (pairwise); the device accumulates in float64
                tol = 1e-5 if values.dtype == np.float32 else 1e-9
                scale = np.nanmax(np.abs(col[np.isfinite(col)])) if np.isfinite(col).any() else 1.0
                np.testing.assert_allclose(got, col, rtol=tol, atol=tol * max(scale, 1.0), equal_nan=True, err_msg=f"{p} {c}")
            parity_log.record("a13 zonal.stats: HIP vs the reference's _stats_numpy, executed", c, got, col,
                              tol="exact" if c in ("zone", "count", "max", "min", "majority") else "1e-5")
        arr = xs.zonal.stats(zagg, vagg, zone_ids=zone_ids, return_type='xarray.DataArray', **kw)
        got = host(arr.data)
        ref = fixture[p + "/array"]
        assert got.shape == ref.shape
        np.testing.assert_array_equal(np.isnan(got), np.isnan(ref), err_msg=p)
        np.testing.assert_array_equal(got[[1, 2, 6, 7]], ref[[1, 2, 6, 7]], err_msg=p + " max/min/count/majority planes")


def _frame_equal(df, want, what, exact=True):
    cols = list(df.columns)
    assert [c if isinstance(c, str) else float(c) for c in cols] == [c for c, _ in want], what
    for gc, (c, col) in zip(cols, want):
        got = np.asarray(df[gc].values, dtype=np.float64)
        col = np.asarray(col, dtype=np.float64)
        if exact:
            np.testing.assert_array_equal(got, col, err_msg=f"{what} column {c}")
        else:
            np.testing.assert_allclose(got, col, rtol=1e-5, atol=1e-6, equal_nan=True, err_msg=f"{what} column {c}")


def test_crosstab_2d_equals_executed_reference(fixture):
    """f4: xrspatial/zonal.py:670-812 executed -- including what the reference does with `cat_ids` that select a strict
    subset of the categories (a selected column also counts the unselected categories sorted in front of it)."""
    for n in names(fixture, "ct"):
        p = f"ct/{n}"
        a = json.loads(str(fixture[p + "/args"]))
        zagg = xs.DataArray(fixture[p + "/zones"], dims=['y', 'x'])
        vagg = xs.DataArray(fixture[p + "/values"], dims=['y', 'x'])
        df = xs.zonal.crosstab(zagg, vagg, zone_ids=a["zone_ids"], cat_ids=a["cat_ids"], nodata_values=a["nodata"], agg=a["agg"])
        _frame_equal(df, rx.table(fixture, p + "/table"), p, exact=(a["agg"] == "count"))


def test_crosstab_3d_equals_executed_reference(fixture):
    for n in names(fixture, "ct3"):
        p = f"ct3/{n}"
        a = json.loads(str(fixture[p + "/args"]))
        values = fixture[p + "/values"]
        vagg = xs.DataArray(values, dims=['layer', 'y', 'x'], coords={'layer': a["layers"]})
        zagg = xs.DataArray(fixture[p + "/zones"], dims=['y', 'x'])
        df = xs.zonal.crosstab(zagg, vagg, zone_ids=a["zone_ids"], cat_ids=a["cat_ids"], nodata_values=a["nodata"], agg=a["agg"])
        _frame_equal(df, rx.table(fixture, p + "/table"), p, exact=a["agg"] in ("count", "min", "max"))


def test_band_ratios_equal_executed_reference(fixture):
    """a10 / f1: `_normalized_ratio_cpu` (ndvi, nbr, nbr2, ndmi) and `_sipi_cpu` executed on float32 bands: bit for bit."""
    for n in names(fixture, "ms"):
        p = f"ms/{n}"
        a, b, c = (xs.DataArray(fixture[f"{p}/{k}"], dims=['y', 'x']) for k in "abc")
        for fn in (xs.multispectral.ndvi, xs.multispectral.nbr, xs.multispectral.nbr2, xs.multispectral.ndmi):
            got = host(fn(a, b).data)
            want = fixture[p + "/normalized_ratio"]
            assert got.dtype == np.float32
            np.testing.assert_array_equal(got.view(np.uint32) | (np.isnan(got) * np.uint32(0x7fffffff)),
                                          want.view(np.uint32) | (np.isnan(want) * np.uint32(0x7fffffff)), err_msg=p)
        got = host(xs.multispectral.sipi(a, b, c).data)
        want = fixture[p + "/sipi"]
        np.testing.assert_array_equal(got.view(np.uint32) | (np.isnan(got) * np.uint32(0x7fffffff)),
                                      want.view(np.uint32) | (np.isnan(want) * np.uint32(0x7fffffff)), err_msg=p)
