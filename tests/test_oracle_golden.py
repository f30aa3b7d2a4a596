"""Pin the CPU oracle against every golden vector the reference's tests hold for
the hot path.  No GPU, no HIP library involved."""
import numpy as np
import pytest

from oracle import xrs_oracle as orc


def test_slope_qgis(golden):
    # xrspatial/tests/test_slope.py:22-52: res=(1,1), rtol 1e-5 on the interior
    out = orc.slope(golden["dem_nan_row"], 1, 1)
    assert out.dtype == np.float32
    np.testing.assert_allclose(out[1:-1, 1:-1], golden["qgis_slope"][1:-1, 1:-1],
                               rtol=1e-5, equal_nan=True)
    assert np.isnan(out[0]).all() and np.isnan(out[-1]).all()
    assert np.isnan(out[:, 0]).all() and np.isnan(out[:, -1]).all()


def test_aspect_qgis(golden):
    # xrspatial/tests/test_aspect.py:20-47
    out = orc.aspect(golden["dem_nan_row"])
    assert out.dtype == np.float32
    np.testing.assert_allclose(out[1:-1, 1:-1], golden["qgis_aspect"][1:-1, 1:-1],
                               rtol=1e-5, equal_nan=True)


@pytest.mark.parametrize("which", ["curv_convex", "curv_concave"])
def test_curvature_spikes(golden, which):
    # xrspatial/tests/test_curvature.py:27-83, res=(1,1) -> cellsize 1
    out = orc.curvature(golden[which + "__0"], 1.0)
    np.testing.assert_allclose(out, golden[which + "__1"], rtol=1e-6, equal_nan=True)
    assert out.dtype == np.float32


@pytest.mark.parametrize("shape", [(2, 4), (10, 15)])
@pytest.mark.parametrize("dtype", [np.int32, np.int64, np.uint32, np.uint64, np.float32, np.float64])
def test_curvature_flat(shape, dtype):
    # xrspatial/tests/test_curvature.py:14-24, 62-69
    out = orc.curvature(np.zeros(shape, dtype=dtype), 1)
    exp = np.zeros(shape, np.float32)
    exp[0] = exp[-1] = np.nan
    exp[:, 0] = exp[:, -1] = np.nan
    np.testing.assert_array_equal(out, exp)


def test_hillshade_docstring():
    # xrspatial/hillshade.py:153-170 (the only numeric known answer the reference has)
    data = np.array([[0., 0., 0., 0., 0.],
                     [0., 1., 0., 2., 0.],
                     [0., 0., 3., 0., 0.],
                     [0., 0., 0., 0., 0.],
                     [0., 0., 0., 0., 0.]])
    exp = np.array([[0.71130913, 0.44167341, 0.71130913],
                    [0.95550163, 0.71130913, 0.52478473],
                    [0.71130913, 0.88382559, 0.71130913]])
    out = orc.hillshade(data)
    np.testing.assert_allclose(out[1:-1, 1:-1], exp, rtol=0, atol=2e-7)
    assert np.isnan(out[0]).all() and np.isnan(out[:, -1]).all()


def test_hillshade_gaussian_positive():
    # xrspatial/tests/test_hillshade.py:17-41
    x = np.linspace(0, 50, 101)
    X, Y = np.meshgrid(x, x, sparse=True)
    g = np.exp((-(X - 25) ** 2 - (Y - 25) ** 2) / 50.0) / 12.5
    out = orc.hillshade(g)
    assert np.nanmean(out) > 0 and out[60, 60] > 0


def test_terrain_compass_rose():
    # xrspatial/analytics.py:22-76 docstring: +/-1 spikes on a 5x8 zero grid
    data = np.zeros((5, 8), dtype=np.float64)
    data[2, 2] = 1
    data[2, 5] = -1
    s = orc.slope(data, 1, 1)
    a = orc.aspect(data)
    c = orc.curvature(data, 1)
    np.testing.assert_allclose(s[1:4, 1:4], [[10.024988, 14.036243, 10.024988],
                                             [14.036243, 0., 14.036243],
                                             [10.024988, 14.036243, 10.024988]], rtol=1e-6)
    np.testing.assert_allclose(a[1:4, 1:4], [[315., 0., 45.], [270., -1., 90.], [225., 180., 135.]])
    np.testing.assert_allclose(a[1:4, 4:7], [[135., 180., 225.], [90., -1., 270.], [45., 0., 315.]])
    np.testing.assert_allclose(c[1:4, 1:4], [[0, -100, 0], [-100, 400, -100], [0, -100, 0]])
    np.testing.assert_allclose(c[1:4, 4:7], [[0, 100, 0], [100, -400, 100], [0, 100, 0]])


def test_kernels(golden):
    # xrspatial/tests/test_focal.py:126-197
    np.testing.assert_array_equal(orc.circle_kernel(1, 1, 1), golden["kernel_circle_1_1_1"])
    np.testing.assert_array_equal(orc.annulus_kernel(2, 2, 2, 1), golden["kernel_annulus_2_2_2_1"])
    assert orc.circle_kernel(1, 1, 2).shape == (5, 5) and orc.circle_kernel(1, 1, 2).sum() == 13
    assert orc.circle_kernel(1, 1, 12).shape == (25, 25) and orc.circle_kernel(1, 1, 12).sum() == 441
    # convolution.py:167-181 docstring
    assert orc.circle_kernel(1, 2, 3).tolist() == [[0, 0, 0, 1, 0, 0, 0], [1] * 7, [0, 0, 0, 1, 0, 0, 0]]


def test_convolve_2d(golden):
    # xrspatial/tests/test_focal.py:113-225
    d = golden["conv_data"]
    np.testing.assert_allclose(orc.convolve_2d(d, golden["conv_custom__0"]),
                               golden["conv_custom__1"], equal_nan=True)
    np.testing.assert_allclose(orc.convolve_2d(d, golden["kernel_circle_1_1_1"]),
                               golden["conv_expected_circle"], equal_nan=True)
    np.testing.assert_allclose(orc.convolve_2d(d, golden["kernel_annulus_2_2_2_1"]),
                               golden["conv_expected_annulus"], equal_nan=True)


def test_convolution_2d_docstring():
    # xrspatial/convolution.py:431-453
    data = np.arange(24).reshape(4, 6)
    kernel = orc.circle_kernel(1, 1, 1)
    out = orc.convolve_2d(data, kernel)
    np.testing.assert_array_equal(out[1:3, 1:5], [[35, 40, 45, 50], [65, 70, 75, 80]])
    assert out.dtype == np.float32 and np.isnan(out[0]).all()


def test_focal_stats_4x4(golden):
    # xrspatial/tests/test_focal.py:353-394; order mean,max,min,range,std,var,sum
    out = orc.focal_stats(golden["focal_stats__0"], golden["focal_stats__1"])
    assert out.dtype == np.float32
    np.testing.assert_allclose(out, golden["focal_stats__2"], rtol=1e-6, equal_nan=True)


def test_focal_apply_docstring():
    # xrspatial/focal.py:371-392 (circle kernel mean over a 4x4 of ones-like arange)
    data = np.arange(20, dtype=np.float64).reshape(4, 5)
    k = orc.circle_kernel(2, 2, 3)
    out = orc.focal_apply(data, k, 'mean')
    exp = np.array([[2., 2.25, 3.25, 4.25, 5.33333333],
                    [5.25, 6., 7., 8., 8.75],
                    [10.25, 11., 12., 13., 13.75],
                    [13.66666667, 14.75, 15.75, 16.75, 17.]])
    np.testing.assert_allclose(out, exp, rtol=1e-6)


def test_focal_mean_docstring():
    # xrspatial/focal.py:195-234
    data = np.zeros((5, 5))
    data[2, 2] = 9
    one = orc.focal_mean3x3(data)
    exp1 = np.zeros((5, 5))
    exp1[1:4, 1:4] = 1
    np.testing.assert_allclose(one, exp1)
    two = orc.focal_mean3x3(data, passes=2)
    exp2 = np.array([[0.25, 1 / 3, 0.5, 1 / 3, 0.25],
                     [1 / 3, 4 / 9, 2 / 3, 4 / 9, 1 / 3],
                     [0.5, 2 / 3, 1.0, 2 / 3, 0.5],
                     [1 / 3, 4 / 9, 2 / 3, 4 / 9, 1 / 3],
                     [0.25, 1 / 3, 0.5, 1 / 3, 0.25]])
    np.testing.assert_allclose(two, exp2, rtol=1e-12)
    # NaN centre cells are in the default excludes -> passed through
    d = data.copy()
    d[0, 0] = np.nan
    assert np.isnan(orc.focal_mean3x3(d)[0, 0])


def test_multispectral_qgis(golden):
    # xrspatial/tests/test_multispectral.py:114-283, 363-470
    nir, red, blue = golden["ms_nir"], golden["ms_red"], golden["ms_blue"]
    out = orc.normalized_ratio(nir, red)
    assert out.dtype == np.float32
    np.testing.assert_allclose(out, golden["qgis_ndvi"], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(orc.savi(nir, red, 0.0), golden["qgis_ndvi"], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(orc.savi(nir, red, 1.0), golden["qgis_savi"], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(orc.evi(nir, red, blue), golden["qgis_evi"], rtol=1e-6, equal_nan=True)
    # other normalized-ratio indices share the kernel (nbr, nbr2, ndmi)
    np.testing.assert_allclose(orc.normalized_ratio(nir, golden["ms_swir2"]), golden["qgis_nbr"],
                               rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(orc.normalized_ratio(golden["ms_swir1"], golden["ms_swir2"]),
                               golden["qgis_nbr2"], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(orc.normalized_ratio(nir, golden["ms_swir1"]), golden["qgis_ndmi"],
                               rtol=1e-6, equal_nan=True)


def test_multispectral_other_indices_qgis(golden):
    # xrspatial/tests/test_multispectral.py:114-128, 235-283, 429-590 (arvi, gci, sipi, ebbi)
    nir, red, blue, green = golden["ms_nir"], golden["ms_red"], golden["ms_blue"], golden["ms_green"]
    np.testing.assert_allclose(orc.arvi(nir, red, blue), golden["qgis_arvi"], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(orc.gci(nir, green), golden["qgis_gci"], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(orc.sipi(nir, red, blue), golden["qgis_sipi"], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(orc.ebbi(red, golden["ms_swir1"], golden["ms_tir"]), golden["qgis_ebbi"],
                               rtol=1e-6, equal_nan=True)
    for dtype in ("uint8", "uint16"):
        n, r, b, exp = (golden["uint_arvi__%d" % i] for i in range(4))
        np.testing.assert_allclose(orc.arvi(n.astype(dtype), r.astype(dtype), b.astype(dtype)), exp, rtol=1e-6)
        n, r, b, exp = (golden["uint_sipi__%d" % i] for i in range(4))
        np.testing.assert_allclose(orc.sipi(n.astype(dtype), r.astype(dtype), b.astype(dtype)), exp, rtol=1e-6)
        r, sw, t, exp = (golden["uint_ebbi__%d" % i] for i in range(4))
        np.testing.assert_allclose(orc.ebbi(r.astype(dtype), sw.astype(dtype), t.astype(dtype)), exp, rtol=1e-6)


@pytest.mark.parametrize("dtype", ["uint8", "uint16"])
def test_multispectral_uint(golden, dtype):
    # xrspatial/tests/test_multispectral.py:286-336
    b1, b2, exp = (golden["uint_nratio__%d" % i] for i in range(3))
    np.testing.assert_allclose(orc.normalized_ratio(b1.astype(dtype), b2.astype(dtype)), exp, rtol=1e-6)
    n, r, b, exp = (golden["uint_evi__%d" % i] for i in range(4))
    np.testing.assert_allclose(orc.evi(n.astype(dtype), r.astype(dtype), b.astype(dtype)), exp, rtol=1e-6)
    n, r, exp = (golden["uint_savi__%d" % i] for i in range(3))
    np.testing.assert_allclose(orc.savi(n.astype(dtype), r.astype(dtype)), exp, rtol=1e-6)


def _check_table(res, exp, rtol=1e-5, atol=1e-7):
    assert list(res) == list(exp) or set(res) == set(exp)
    assert (np.asarray(res['zone']) == np.asarray(exp['zone'])).all()
    for col in exp:
        if col != 'zone':
            np.testing.assert_allclose(res[col], exp[col], rtol=rtol, atol=atol, equal_nan=True)


def test_zonal_default(golden, golden_tables):
    # xrspatial/tests/test_zonal.py:31-90, 408-426
    res = orc.zonal_stats(golden["zonal_zones"], golden["zonal_values"])
    _check_table(res, golden_tables["zonal_default"])
    arr = orc.zonal_stats(golden["zonal_zones"], golden["zonal_values"], return_type='array')
    np.testing.assert_allclose(arr, golden["zonal_default_da"], rtol=1e-5, atol=1e-7, equal_nan=True)


def test_zonal_zone_ids(golden, golden_tables):
    # xrspatial/tests/test_zonal.py:131-202, 453-494
    ids = golden_tables["zonal_zone_ids__0"]
    res = orc.zonal_stats(golden["zonal_zones"], golden["zonal_values"], zone_ids=ids)
    _check_table(res, golden_tables["zonal_zone_ids__1"])
    arr = orc.zonal_stats(golden["zonal_zones"], golden["zonal_values"], zone_ids=ids, return_type='array')
    np.testing.assert_allclose(arr, golden["zonal_zone_ids_da__1"], rtol=1e-5, atol=1e-7, equal_nan=True)


def test_zonal_custom(golden, golden_tables):
    # xrspatial/tests/test_zonal.py:204-237, 497-544
    funcs = {'double_sum': lambda v: v.sum() * 2, 'range': lambda v: v.max() - v.min()}
    nodata, ids, exp = (golden_tables["zonal_custom__%d" % i] for i in range(3))
    res = orc.zonal_stats(golden["zonal_zones"], golden["zonal_values"], zone_ids=ids,
                          stats_funcs=funcs, nodata_values=nodata)
    _check_table(res, exp)
    arr = orc.zonal_stats(golden["zonal_zones"], golden["zonal_values"], zone_ids=ids,
                          stats_funcs=funcs, nodata_values=nodata, return_type='array')
    np.testing.assert_allclose(arr, golden["zonal_custom_da__2"], equal_nan=True)


def test_zonal_majority_ties():
    # xrspatial/tests/test_zonal.py:567-590
    z = np.array([[1, 1, 1, 1], [1, 1, 2, 2], [2, 2, 2, 2]])
    v = np.array([[1, 1, 2, 2], [3, 3, 5, 5], [5, 5, 6, 6]])
    res = orc.zonal_stats(z, v, stats_funcs=['majority'])
    assert res['zone'].tolist() == [1, 2] and res['majority'].tolist() == [1, 5]


def test_zonal_qgis(golden, golden_tables):
    # xrspatial/tests/test_zonal.py:340-385, 593-601
    exp = golden_tables["zonal_qgis"]
    res = orc.zonal_stats(golden["zones_8x6"], golden["dem"],
                          stats_funcs=[k for k in exp if k != 'zone'])
    _check_table(res, exp, atol=1e-5)
    assert res['count'].tolist() == exp['count']


def test_hotspots(golden):
    # xrspatial/tests/test_focal.py:426-470 and the docstring example focal.py:1058-1072
    got, _ = orc.hotspots(golden["hotspots__0"], golden["hotspots__1"])
    np.testing.assert_array_equal(got, golden["hotspots__2"])
    assert got.dtype == np.int8
    data = np.array([[0, 1000, 1000, 0, 0, 0], [0, 0, 0, -1000, -1000, 0], [0, -900, -900, 0, 0, 0], [0, 100, 1000, 0, 0, 0]])
    got, _ = orc.hotspots(data, np.array([[1, 1, 0]]))
    np.testing.assert_array_equal(got, [[0, 0, 95, 0, 0, 0], [0, 0, 0, 0, -90, 0], [0, 0, -90, 0, 0, 0], [0, 0, 0, 0, 0, 0]])
    with pytest.raises(ZeroDivisionError):
        orc.hotspots(np.zeros((10, 20)), np.ones((3, 3)))


def test_trim_crop_reference_cases():
    """The reference's own trim / crop tests (test_zonal.py:1047-1211) pin the bounds restatement."""
    from tests.trim_crop_cases import CASES
    for fn, arr, values, want in CASES:
        top, bottom, left, right = (orc.trim_bounds if fn == "trim" else orc.crop_bounds)(arr, values)
        np.testing.assert_array_equal(arr[top:bottom + 1, left:right + 1], want)
    # the scans' far-edge behaviour when nothing stops them, and NaN never matching (`e == val`)
    assert orc.trim_bounds(np.zeros((3, 4)), (0,)) == (2, 0, 3, 0)
    assert orc.crop_bounds(np.zeros((3, 4)), (7,)) == (2, 0, 3, 0)
    nan_edge = np.array([[np.nan, np.nan], [np.nan, 1.0]])
    assert orc.trim_bounds(nan_edge) == (0, 1, 0, 1)                   # default values=(nan,) trims nothing
    assert orc.trim_bounds(nan_edge, (1.0,)) == (0, 1, 0, 1)           # NaN cells are "data"
    assert orc.crop_bounds(nan_edge, (np.nan,)) == (1, 0, 1, 0)


def test_true_color_properties():
    """No golden exists for true_color (the reference only compares numpy with dask); the restatement is held to the
    properties of xrspatial/multispectral.py:1334-1399: band minimum -> sigmoid(th) of 0, maximum -> of 1, constant
    band and NaN cells -> 0, alpha 0 on NaN / <= nodata, monotone in the band value."""
    rng = np.random.default_rng(3)
    r = rng.random((12, 17)) * 4000
    r[2, 3] = np.nan
    r[4, 5] = 0.5
    g = np.full((12, 17), 7.0)
    b = rng.integers(0, 255, (12, 17)).astype(np.int32)
    out = orc.true_color(r, g, b)
    assert out.shape == (12, 17, 4) and out.dtype == np.uint8
    lo, hi = int(255 / (1 + np.exp(10 * 0.125))), int(255 / (1 + np.exp(10 * (0.125 - 1))))
    assert out[..., 0][np.unravel_index(np.nanargmin(r), r.shape)] == lo
    assert out[..., 0][np.unravel_index(np.nanargmax(r), r.shape)] == hi
    assert (out[..., 1] == 0).all() and out[2, 3, 0] == 0
    assert out[2, 3, 3] == 0 and out[4, 5, 3] == 0 and (np.delete(out[..., 3].ravel(), [2 * 17 + 3, 4 * 17 + 5]) == 255).all()
    order = np.argsort(b.ravel(), kind='stable')
    assert (np.diff(out[..., 2].ravel()[order].astype(int)) >= 0).all()


# ---------------------------------------------------------------- geodesic slope / aspect: an analytic pin
def _analytic_geodesic_case(lat0_deg, g_north, g_east, h0=350.0, cell_arcsec=1.0, shape=(9, 11)):
    """Elevations that vary LINEARLY with latitude and longitude around (lat0, 10 E) on the WGS84 ellipsoid:
        h = h0 + g_north * (M + h0) * dphi + g_east * (N + h0) * cos(phi0) * dlambda
    (M, N: meridional / prime-vertical radii of curvature).  (M + h0) dphi and (N + h0) cos(phi0) dlambda are the local
    northing / easting in metres, so the surface has gradient (g_east, g_north) in the tangent plane of its centre cell:
        slope = atan(hypot(g_east, g_north)),  aspect = compass bearing of steepest DESCENT = atan2(-g_east, -g_north).
    The reference's sphere-flattening term (e^2 + n^2) / 2R is symmetric and drops out of a centred plane fit; what is
    left are O(cell size) effects (the rows' differing cos(phi)): ~1e-5 relative at 1 arc-second."""
    a, b = 6378137.0, 6356752.314245
    phi0 = np.deg2rad(lat0_deg)
    w = np.sqrt(a * a * np.cos(phi0) ** 2 + b * b * np.sin(phi0) ** 2)
    N = a * a / w
    M = a * a * b * b / w ** 3
    step = np.deg2rad(cell_arcsec / 3600.0)
    H, W = shape
    dphi = -(np.arange(H) - H // 2) * step                   # row 0 is the northernmost (descending latitude)
    dlam = (np.arange(W) - W // 2) * step
    LAT = np.rad2deg(phi0 + dphi)[:, None] * np.ones((1, W))
    LON = (10.0 + np.rad2deg(dlam))[None, :] * np.ones((H, 1))
    elev = h0 + g_north * (M + h0) * dphi[:, None] + g_east * (N + h0) * np.cos(phi0) * dlam[None, :]
    slope = np.degrees(np.arctan(np.hypot(g_east, g_north)))
    aspect = np.degrees(np.arctan2(-g_east, -g_north)) % 360.0
    return elev, LAT, LON, slope, aspect


GEO_CASES = [(0.0, 0.05, 0.0), (45.0, 0.0, 0.08), (45.0, 0.03, -0.04), (-33.0, -0.2, 0.1), (60.0, 0.5, 0.5), (78.0, -0.01, -0.02)]


@pytest.mark.parametrize("lat0,g_north,g_east", GEO_CASES)
def test_geodesic_oracle_matches_closed_form(lat0, g_north, g_east):
    """Pins the geodesic part of the oracle (xrspatial/geodesic.py:40-229 restated), for which the reference's tests hold
    properties only: ECEF conversion on WGS84, the East / North / Up frame and its signs, the centred plane fit and the
    slope / aspect conventions must reproduce a surface whose tangent-plane gradient is known in closed form."""
    elev, LAT, LON, slope, aspect = _analytic_geodesic_case(lat0, g_north, g_east)
    got_s = orc.geodesic_slope(elev, LAT, LON)[4, 5]
    got_a = orc.geodesic_aspect(elev, LAT, LON)[4, 5]
    np.testing.assert_allclose(got_s, slope, rtol=1e-4)
    assert abs((got_a - aspect + 180.0) % 360.0 - 180.0) < 0.01, (got_a, aspect)
    # float32 elevations (what the device path is fed) stay within the same bar at this relief
    got32 = orc.geodesic_slope(elev.astype(np.float32), LAT, LON)[4, 5]
    np.testing.assert_allclose(got32, slope, rtol=2e-3)


def test_geodesic_oracle_level_surface_is_flat():
    """A surface of constant elevation follows the ellipsoid: slope 0 (the reference's curvature term), aspect -1."""
    elev, LAT, LON, _, _ = _analytic_geodesic_case(37.0, 0.0, 0.0, h0=1200.0, cell_arcsec=3.0)
    s = orc.geodesic_slope(elev, LAT, LON)[1:-1, 1:-1]
    assert np.all(s < 2e-3)                       # degrees; the WGS84 / mean-sphere mismatch of the correction is what is left


# ---------------------------------------------------------------- geodesic: an independent 40-digit evaluation of the 3x3 fit
def _geodesic_fit_mp(elev3, lat3, lon3, z_factor=1.0, correction=True):
    """(slope_deg, aspect_deg) at the centre of a 3x3 neighbourhood, evaluated INDEPENDENTLY of oracle/xrs_oracle.py in 40-digit
    arithmetic (mpmath): WGS84 geodetic -> ECEF, the East / North / Up frame of the centre cell as a rotation applied to the
    difference vectors, the sphere-flattening term u += (e^2 + n^2) / 2R of xrspatial/geodesic.py:96-97, and the plane
    u = A e + B n + C as a three-parameter least-squares problem solved by mpmath (not the centred normal equations the
    reference and the oracle spell out)."""
    import mpmath as mp
    mp.mp.dps = 40
    a, b = mp.mpf('6378137.0'), mp.mpf('6356752.314245')
    R = mp.mpf('6370994.884953014')
    d2r = mp.pi / 180

    def ecef(lat, lon, h):
        cl, sl = mp.cos(lat), mp.sin(lat)
        N = a * a / mp.sqrt(a * a * cl * cl + b * b * sl * sl)
        return mp.matrix([(N + h) * cl * mp.cos(lon), (N + h) * cl * mp.sin(lon), (b * b / (a * a) * N + h) * sl])

    latc, lonc = mp.mpf(float(lat3[1, 1])) * d2r, mp.mpf(float(lon3[1, 1])) * d2r
    pc = ecef(latc, lonc, mp.mpf(float(elev3[1, 1])) * z_factor)
    rot = mp.matrix([[-mp.sin(lonc), mp.cos(lonc), 0],
                     [-mp.sin(latc) * mp.cos(lonc), -mp.sin(latc) * mp.sin(lonc), mp.cos(latc)],
                     [mp.cos(latc) * mp.cos(lonc), mp.cos(latc) * mp.sin(lonc), mp.sin(latc)]])
    rows, rhs = [], []
    for i in range(3):
        for j in range(3):
            p = ecef(mp.mpf(float(lat3[i, j])) * d2r, mp.mpf(float(lon3[i, j])) * d2r, mp.mpf(float(elev3[i, j])) * z_factor)
            e, n, u = rot * (p - pc)
            if correction:
                u += (e * e + n * n) / (2 * R)
            rows.append([e, n, 1])
            rhs.append(u)
    A, B, _ = mp.lu_solve(mp.matrix(rows), mp.matrix(rhs))          # (overdetermined: the least-squares solution)
    slope = mp.degrees(mp.atan(mp.sqrt(A * A + B * B)))
    aspect = mp.degrees(mp.atan2(-A, -B)) % 360
    return float(slope), float(aspect)


def _curved_geodesic_case(lat0_deg, cell_deg, seed):
    """A 3x3 patch of a CURVED surface (quadratic in northing / easting, random coefficients, 3-30 % grades) on the
    reference's grid geometry: descending latitudes, ascending longitudes."""
    rng = np.random.default_rng(seed)
    lat = lat0_deg - (np.arange(3) - 1) * cell_deg
    lon = 10.0 + (np.arange(3) - 1) * cell_deg
    LAT, LON = np.meshgrid(lat, lon, indexing='ij')
    north = np.deg2rad(LAT - lat0_deg) * 6371000.0
    east = np.deg2rad(LON - 10.0) * 6371000.0 * np.cos(np.deg2rad(lat0_deg))
    span = north.max() - north.min()
    g = rng.uniform(-0.3, 0.3, 2)
    q = rng.uniform(-0.3, 0.3, 3) / span                          # curvature: the gradient changes by ~0.3 across the patch
    elev = 800.0 + g[0] * north + g[1] * east + q[0] * north ** 2 + q[1] * north * east + q[2] * east ** 2
    return elev, LAT, LON


@pytest.mark.parametrize("lat0", [0.0, 37.0, -58.0, 75.0])
@pytest.mark.parametrize("cell_deg", [1.0 / 3600.0, 30.0 / 3600.0, 0.25])
def test_geodesic_oracle_matches_independent_40_digit_fit(lat0, cell_deg):
    """The geodesic oracle (float64 restatement of xrspatial/geodesic.py:40-229; the reference holds no vectors for it) on
    curved surfaces against the independent 40-digit evaluation above: the ECEF conversion, the ENU projection, the
    curvature-correction term and the plane fit each contribute to these numbers -- at 0.25-degree cells and 75 N the
    correction term alone moves the slope by far more than the tolerance, which the test checks -- and must agree to 1e-7
    relative (float64 rounding of ~6e6 m coordinates differenced over a cell) in slope and 1e-6 degrees in aspect."""
    for seed in range(4):
        elev, LAT, LON = _curved_geodesic_case(lat0, cell_deg, seed)
        want_s, want_a = _geodesic_fit_mp(elev, LAT, LON)
        got_s = float(orc._geodesic_plane_fit(elev, LAT, LON, 1.0)[0][0, 0])      # A, B in float64 (before the float32 store)
        A, B, valid = orc._geodesic_plane_fit(elev, LAT, LON, 1.0)
        assert bool(valid[0, 0])
        got_s = float(np.degrees(np.arctan(np.hypot(A[0, 0], B[0, 0]))))
        got_a = float(np.degrees(np.arctan2(-A[0, 0], -B[0, 0])) % 360.0)
        # float64 cancellation: ECEF coordinates of ~6.4e6 m differenced over one cell, i.e. ~1e-16 * 6.4e6 / cell relative
        cell_m = np.deg2rad(cell_deg) * 6.371e6
        tol = max(1e-9, 20 * 2.2e-16 * 6.4e6 / cell_m)
        np.testing.assert_allclose(got_s, want_s, rtol=tol, err_msg=f"slope lat {lat0} cell {cell_deg} seed {seed}")
        assert abs((got_a - want_a + 180.0) % 360.0 - 180.0) < max(1e-6, 100 * tol), (got_a, want_a)
        # the float32 result the public functions return
        np.testing.assert_allclose(orc.geodesic_slope(elev, LAT, LON)[1, 1], np.float32(want_s), rtol=max(2e-7, tol))
    # the correction term is exercised: without it the large-cell fit is a different number -- 1e-5 relative at 0.25-degree
    # cells and 75 N, four orders of magnitude above the tolerance the oracle is held to there
    elev, LAT, LON = _curved_geodesic_case(75.0, 0.25, 0)
    with_c, _ = _geodesic_fit_mp(elev, LAT, LON)
    without_c, _ = _geodesic_fit_mp(elev, LAT, LON, correction=False)
    assert abs(with_c - without_c) > 5e-6 * with_c


def test_geodesic_fit_approaches_the_analytic_gradient():
    """On a small stencil the fitted plane of a curved surface is its tangent plane at the centre: slope -> atan |grad h|,
    aspect -> bearing of -grad h, to O(cell / curvature radius) -- the closed form behind the geometry (1 arc-second cells)."""
    for lat0 in (12.0, 51.0):
        rng = np.random.default_rng(int(lat0))
        cell = 1.0 / 3600.0
        lat = lat0 - (np.arange(3) - 1) * cell
        lon = 10.0 + (np.arange(3) - 1) * cell
        LAT, LON = np.meshgrid(lat, lon, indexing='ij')
        a, b = 6378137.0, 6356752.314245
        phi0 = np.deg2rad(lat0)
        w = np.sqrt(a * a * np.cos(phi0) ** 2 + b * b * np.sin(phi0) ** 2)
        N, M = a * a / w, a * a * b * b / w ** 3
        north = np.deg2rad(LAT - lat0) * (M + 800.0)
        east = np.deg2rad(LON - 10.0) * (N + 800.0) * np.cos(phi0)
        gn, ge = rng.uniform(-0.3, 0.3, 2)
        elev = 800.0 + gn * north + ge * east + 2e-4 * north ** 2 - 1e-4 * north * east + 3e-4 * east ** 2
        s = float(orc.geodesic_slope(elev, LAT, LON)[1, 1])
        asp = float(orc.geodesic_aspect(elev, LAT, LON)[1, 1])
        np.testing.assert_allclose(s, np.degrees(np.arctan(np.hypot(gn, ge))), rtol=2e-4)
        assert abs((asp - np.degrees(np.arctan2(-ge, -gn)) + 180.0) % 360.0 - 180.0) < 0.02


def test_crosstab_oracle_matches_reference_tables(golden, golden_tables):
    """orc.crosstab_2d / the per-layer zonal statistics behind the 3-D crosstab against the reference's own expectations
    (xrspatial/tests/test_zonal.py:240-336, lifted by make_golden.py)."""
    T = golden_tables
    zones, values = golden["zonal_zones"], golden["zonal_values"]
    got = orc.crosstab_2d(zones, values, zone_ids=[int(z) for z in T["crosstab_2d_count__0"]],
                          cat_ids=[int(c) for c in T["crosstab_2d_count__1"]])
    for col, want in T["crosstab_2d_count__2"].items():
        key = 'zone' if col == 'zone' else int(col)
        np.testing.assert_allclose(np.asarray(got[key], dtype=float), want, err_msg=col)
    got = orc.crosstab_2d(zones, values, zone_ids=[int(z) for z in T["crosstab_2d_percentage__1"]],
                          cat_ids=[int(c) for c in T["crosstab_2d_percentage__2"]], nodata_values=T["crosstab_2d_percentage__0"],
                          agg='percentage')
    for col, want in T["crosstab_2d_percentage__3"].items():
        key = 'zone' if col == 'zone' else int(col)
        np.testing.assert_allclose(np.asarray(got[key], dtype=float), want, err_msg=col)
    # 3-D values (a layer per category, all ones: data_values_3d): every aggregation is zonal.stats of a layer
    layer_vals = np.ones(zones.shape)
    zone_ids = [int(z) for z in T["crosstab_3d__1"]]
    for agg in ('min', 'max', 'mean', 'sum', 'std', 'var', 'count'):
        st = orc.zonal_stats(zones, layer_vals, zone_ids=zone_ids, stats_funcs=[agg])
        for cat in ('cat1', 'cat2', 'cat3', 'cat4'):
            np.testing.assert_allclose(st[agg], T[f"crosstab_3d__2__{agg}"][cat], err_msg=f"{agg} {cat}")
