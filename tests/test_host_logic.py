"""Host-side logic of the drop-in layer, exercised without a GPU.

These mirror the reference's own host-level tests (xrspatial/tests/test_focal.py:178-336 kernel builders and
cell sizes, test_dataset_support.py:111-208 adapter errors, test_multispectral.py argument validation,
test_zonal.py return-type checks, xrspatial/utils.py:146-277 validation / resolution helpers) and check that
every entry point validates its arguments BEFORE touching the device, so a bad call fails the same way on a
box without a GPU as the reference's CPU path does.
"""
import numpy as np
import pytest

import xrspatial_amd as xa
from xrspatial_amd import convolution, focal, geodesic, multispectral, utils, zonal
from xrspatial_amd._xr import DataArray, Dataset
from xrspatial_amd.convolution import annulus_kernel, calc_cellsize, circle_kernel, custom_kernel
from xrspatial_amd.distributed import combine_zonal_partials, shard_halos, shard_rows


def raster(data, res=(0.5, 0.5), attrs=None, dims=('y', 'x')):
    data = np.asarray(data)
    h, w = data.shape[-2:]
    a = {'res': res}
    a.update(attrs or {})
    coords = {dims[-1]: np.linspace(0, (w - 1) * res[0], w), dims[-2]: np.linspace((h - 1) * res[1], 0, h)}
    return DataArray(data, dims=dims, coords=coords, attrs=a)


# ---------------------------------------------------------------- kernels (test_focal.py:178-197)
def test_custom_kernel_rejects_lists_and_even_shapes():
    with pytest.raises(ValueError):
        custom_kernel([1, 0, 0])
    with pytest.raises(ValueError):
        custom_kernel(np.ones((4, 6)))
    k = np.ones((3, 5))
    assert custom_kernel(k) is k


def test_circle_and_annulus_kernels():
    np.testing.assert_array_equal(circle_kernel(1, 1, 1), [[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    np.testing.assert_array_equal(annulus_kernel(2, 2, 2, 1), [[0, 1, 0], [1, 0, 1], [0, 1, 0]])
    k = circle_kernel(1, 1, 12)
    assert k.shape == (25, 25) and k.dtype == np.float64
    assert k[12, 0] == 1 and k[0, 12] == 1 and k[0, 0] == 0
    np.testing.assert_array_equal(k, k[::-1, ::-1])
    np.testing.assert_array_equal(k, k.T)
    # anisotropic cells: wider than tall
    assert circle_kernel(1, 2, 4).shape == (5, 9)
    # radius as a distance string
    np.testing.assert_array_equal(circle_kernel(1000, 1000, '3 km'), circle_kernel(1, 1, 3))
    np.testing.assert_array_equal(circle_kernel(0.3048, 0.3048, '2ft'), circle_kernel(1, 1, 2))


@pytest.mark.parametrize("bad", ['-3', '0', 'abc', '3 parsecs', '3 km 2'])
def test_kernel_distance_validation(bad):
    with pytest.raises(ValueError):
        circle_kernel(1, 1, bad)


def test_calc_cellsize():          # test_focal.py:303-312
    data = np.zeros((6, 6))
    assert calc_cellsize(raster(data, res=(1, 1), attrs={'unit': 'km'})) == (1000, 1000)
    assert calc_cellsize(raster(data)) == (0.5, 0.5)
    cx, cy = calc_cellsize(raster(data, res=(2, -3), attrs={'unit': 'ft'}))
    assert cx == pytest.approx(0.6096) and cy == pytest.approx(0.9144)


# ---------------------------------------------------------------- utils (utils.py:146-277)
def test_validate_arrays():
    a = np.zeros((3, 4))
    utils.validate_arrays(a, a.copy())
    with pytest.raises(ValueError):
        utils.validate_arrays(a)
    with pytest.raises(ValueError):
        utils.validate_arrays(a, np.zeros((4, 3)))


def test_resolution_helpers():
    r = raster(np.zeros((4, 5)), res=(10.0, 20.0))
    assert utils.get_dataarray_resolution(r) == (10.0, 20.0)
    # no 'res' attribute: derived from the coordinates
    bare = DataArray(np.zeros((4, 5)), dims=('y', 'x'),
                     coords={'x': np.arange(5) * 2.0, 'y': np.arange(4)[::-1] * 3.0})
    cx, cy = utils.calc_res(bare)
    assert cx == pytest.approx(2.0) and abs(cy) == pytest.approx(3.0)
    assert utils.get_dataarray_resolution(bare)[0] == pytest.approx(2.0)
    # scalar res attribute means square cells
    sq = DataArray(np.zeros((4, 5)), dims=('y', 'x'), attrs={'res': 7})
    assert utils.get_dataarray_resolution(sq) == (7, 7)


def test_array_type_dispatch_without_a_backend():
    mapper = utils.ArrayTypeFunctionMapping(numpy_func=lambda x: 'np', hip_func=None)
    assert mapper(DataArray(np.zeros((2, 2))))(None) == 'np'      # returns the implementation, like upstream
    with pytest.raises(TypeError):
        class Holder:            # neither a host nor a device array behind .data
            data = [[1, 2], [3, 4]]
        mapper(Holder())
    with pytest.raises(NotImplementedError):
        utils.not_implemented_func(None)


# ---------------------------------------------------------------- argument validation ahead of the device
def test_focal_argument_validation():
    r = raster(np.zeros((5, 5)))
    k = np.ones((3, 3))
    with pytest.raises(TypeError):
        focal.apply(np.zeros((5, 5)), k)
    with pytest.raises(ValueError):
        focal.apply(raster(np.zeros((2, 5, 5)), dims=('t', 'y', 'x')), k)
    with pytest.raises(ValueError):
        focal.apply(r, np.ones((2, 3)))
    with pytest.raises(TypeError):
        focal.apply(r, k, func=42)                       # neither a built-in reducer nor a callable
    with pytest.raises(TypeError):
        focal.focal_stats(np.zeros((5, 5)), k)
    with pytest.raises(KeyError):
        focal.focal_stats(r, k, stats_funcs=['median'])
    with pytest.raises(ValueError):
        focal.mean(raster(np.zeros((2, 5, 5)), dims=('t', 'y', 'x')))
    with pytest.raises(TypeError):
        focal.hotspots(np.zeros((5, 5)), k)
    with pytest.raises(ValueError):
        focal.hotspots(raster(np.zeros((2, 5, 5)), dims=('t', 'y', 'x')), k)


def test_builtin_reducer_tokens():
    # the reference passes numba functions (_calc_mean ... focal.py:226-258); here they are device tokens
    for stat in ('mean', 'max', 'min', 'range', 'std', 'var', 'sum'):
        tok = getattr(focal, '_calc_' + stat)
        assert focal._reducer_name(tok) == stat
        assert stat in repr(tok)


def test_multispectral_argument_validation():
    a = raster(np.ones((4, 4)))
    b = raster(np.ones((4, 5)))
    with pytest.raises(ValueError):
        multispectral.ndvi(a, b)
    with pytest.raises(ValueError):
        multispectral.evi(a, a, b)
    with pytest.raises(ValueError):
        multispectral.evi(a, a, a, c1='6')
    with pytest.raises(ValueError):
        multispectral.evi(a, a, a, c2=None)
    with pytest.raises(ValueError):
        multispectral.evi(a, a, a, soil_factor=1.5)
    with pytest.raises(ValueError):
        multispectral.evi(a, a, a, gain=-1)
    with pytest.raises(ValueError):
        multispectral.savi(a, a, soil_factor=-2)
    with pytest.raises(ValueError):
        multispectral.arvi(a, a, b)
    with pytest.raises(ValueError):
        multispectral.ebbi(a, b, a)


def test_dataset_adapter_errors():                      # test_dataset_support.py:111-165, 208
    ds = Dataset({'nir': raster(np.ones((4, 4))), 'red': raster(np.ones((4, 4)))})
    with pytest.raises(TypeError, match="'red' keyword required"):
        multispectral.ndvi(ds, nir='nir')
    with pytest.raises(ValueError, match="not in Dataset"):
        multispectral.ndvi(ds, nir='nir', red='crimson')
    zones = raster(np.zeros((4, 4), dtype=np.int32))
    with pytest.raises(ValueError):
        zonal.stats(zones, ds, return_type='xarray.DataArray')


def test_zonal_argument_validation():
    zones = raster(np.zeros((4, 4), dtype=np.int32))
    vals = raster(np.ones((4, 4)))
    with pytest.raises(ValueError):
        zonal.stats(zones, raster(np.ones((4, 5))))
    with pytest.raises(ValueError):
        zonal.stats(raster(np.zeros((4, 4), dtype=bool)), vals)      # zones must be integers or floats
    with pytest.raises(ValueError):
        zonal.stats(zones, raster(np.zeros((4, 4), dtype=bool)))
    with pytest.raises(ValueError):
        zonal.stats(zones, vals, return_type='dict')
    with pytest.raises(ValueError):
        zonal.stats(zones, vals, stats_funcs={'mine': 'median'})     # dict values must be callables
    with pytest.raises((KeyError, ValueError)):
        zonal.stats(zones, vals, stats_funcs=['mode'])
    with pytest.raises(ValueError):
        zonal.crosstab(zones, vals, agg='median')


def test_geodesic_validation():                          # utils.py:608-713, slope.py:300-330
    assert geodesic.z_factor_of('meter') == 1.0
    assert geodesic.z_factor_of('ft') == pytest.approx(0.3048)
    with pytest.raises(ValueError):
        geodesic.z_factor_of('cubit')
    z = np.zeros((4, 5))
    ok = DataArray(z, dims=('lat', 'lon'), coords={'lat': np.linspace(41, 40, 4), 'lon': np.linspace(-100, -99, 5)})
    lat, lon, is_2d = geodesic.extract_latlon(ok)
    assert not is_2d and lat.shape == (4,) and lon.shape == (5,)
    with pytest.raises(ValueError, match="Latitude"):
        geodesic.extract_latlon(DataArray(z, dims=('lat', 'lon'), coords={'lat': np.linspace(100, 95, 4),
                                                                         'lon': np.linspace(0, 1, 5)}))
    with pytest.raises(ValueError, match="Longitude"):
        geodesic.extract_latlon(DataArray(z, dims=('lat', 'lon'), coords={'lat': np.linspace(10, 9, 4),
                                                                         'lon': np.linspace(400, 401, 5)}))
    with pytest.raises(ValueError):
        geodesic.extract_latlon(DataArray(z, dims=('row', 'col')))    # no recognisable lat/lon coordinate
    with pytest.raises(ValueError):
        xa.slope(ok, method='geodetic')
    with pytest.raises(ValueError):
        xa.aspect(ok, method='geodesic', z_unit='cubit')


# ---------------------------------------------------------------- zonal host arithmetic
def test_finalize_stats_from_partials():
    rng = np.random.default_rng(3)
    v = rng.normal(50, 20, (40, 30))
    z = rng.integers(0, 4, v.shape)
    count = np.array([(z == k).sum() for k in range(4)], dtype=np.int64)
    s1 = np.array([v[z == k].sum() for k in range(4)])
    s2 = np.array([(v[z == k] ** 2).sum() for k in range(4)])
    mn = np.array([v[z == k].min() for k in range(4)])
    mx = np.array([v[z == k].max() for k in range(4)])
    out = zonal.finalize_stats(['mean', 'max', 'min', 'sum', 'std', 'var', 'count'], count, s1, s2, mn, mx)
    for k in range(4):
        np.testing.assert_allclose(out['mean'][k], v[z == k].mean(), rtol=1e-12)
        np.testing.assert_allclose(out['var'][k], v[z == k].var(), rtol=1e-9)
        np.testing.assert_allclose(out['std'][k], v[z == k].std(), rtol=1e-9)
        assert out['count'][k] == count[k] and out['max'][k] == mx[k] and out['min'][k] == mn[k]

    # shifted moments: a raster with a large offset and a small spread (1e6 +- 1e-2).  The unshifted one-pass variance
    # loses ~12 of 16 digits; with the shift it keeps them (zonal.finalize_stats(..., shift))
    w = 1.0e6 + rng.normal(0, 1e-2, v.shape)
    shift = float(w.flat[7])
    d = w - shift
    cnt = np.array([(z == k).sum() for k in range(4)], dtype=np.int64)
    sh = zonal.finalize_stats(['mean', 'sum', 'var', 'std'], cnt, np.array([d[z == k].sum() for k in range(4)]),
                              np.array([(d[z == k] ** 2).sum() for k in range(4)]), mn, mx, None, shift)
    plain = zonal.finalize_stats(['var'], cnt, np.array([w[z == k].sum() for k in range(4)]),
                                 np.array([(w[z == k] ** 2).sum() for k in range(4)]), mn, mx)
    for k in range(4):
        np.testing.assert_allclose(sh['var'][k], w[z == k].var(), rtol=1e-9)
        np.testing.assert_allclose(sh['mean'][k], w[z == k].mean(), rtol=1e-15)
        np.testing.assert_allclose(sh['sum'][k], w[z == k].sum(), rtol=1e-14)
    assert max(abs(plain['var'][k] / w[z == k].var() - 1) for k in range(4)) > 1e-6      # what the shift is for

    # the same partials split over two row shards combine to the same moments (row e1 of SURVEY 8e)
    halves = []
    for rows in (slice(0, 17), slice(17, 40)):
        zz, vv = z[rows], v[rows]
        halves.append((np.array([(zz == k).sum() for k in range(4)], dtype=np.int64),
                       np.array([vv[zz == k].sum() for k in range(4)]),
                       np.array([(vv[zz == k] ** 2).sum() for k in range(4)]),
                       np.array([vv[zz == k].min() for k in range(4)]),
                       np.array([vv[zz == k].max() for k in range(4)])))
    c, a1, a2, lo, hi = combine_zonal_partials(halves)
    np.testing.assert_array_equal(c, count)
    np.testing.assert_allclose(a1, s1, rtol=1e-13)
    np.testing.assert_allclose(a2, s2, rtol=1e-13)
    np.testing.assert_array_equal(lo, mn)
    np.testing.assert_array_equal(hi, mx)


def test_dense_zone_index_host():
    z = np.array([[7, 7, -3], [1000000, 7, -3]], dtype=np.int64)
    ids, idx = zonal._dense_zone_index(z)
    np.testing.assert_array_equal(ids, [-3, 7, 1000000])
    np.testing.assert_array_equal(idx, [[1, 1, 0], [2, 1, 0]])
    assert idx.dtype == np.int32 and ids.dtype == z.dtype
    # float zones: NaN / inf cells belong to no zone, non-integral ids take the sort path
    zf = np.array([[0.5, np.nan, 2.0], [np.inf, 0.5, -1.0]])
    ids, idx = zonal._dense_zone_index(zf)
    np.testing.assert_array_equal(ids, [-1.0, 0.5, 2.0])
    np.testing.assert_array_equal(idx, [[1, -1, 2], [-1, 1, 0]])
    # ids too far apart for a lookup table
    zw = np.array([[-2**40, 5], [2**40, 5]], dtype=np.int64)
    ids, idx = zonal._dense_zone_index(zw)
    np.testing.assert_array_equal(ids, [-2**40, 5, 2**40])
    np.testing.assert_array_equal(idx, [[0, 1], [2, 1]])
    ids, idx = zonal._dense_zone_index(np.full((2, 2), np.nan))
    assert ids.size == 0 and (idx == -1).all()


def test_shard_geometry():
    # shards tile the rows exactly, halos stop at the raster edge
    for rows, n in ((16384, 8), (1001, 3), (7, 7), (5, 8)):
        cuts = [shard_rows(rows, n, r) for r in range(n)]
        assert cuts[0][0] == 0 and cuts[-1][1] == rows
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(n - 1))
        assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
        for r, (a, b) in enumerate(cuts):
            top, bot = shard_halos(n, r, 12)
            assert top == (12 if r > 0 else 0) and bot == (12 if r < n - 1 else 0)


def test_xarray_stand_in_round_trip():
    r = raster(np.arange(12.0).reshape(3, 4), attrs={'unit': 'm'})
    assert r.shape == (3, 4) and r.ndim == 2 and r.dims == ('y', 'x')
    assert list(r.coords) == ['x', 'y'] or set(r.coords) == {'x', 'y'}
    np.testing.assert_array_equal(r.values, np.arange(12.0).reshape(3, 4))
    with pytest.raises(ValueError):
        DataArray(np.zeros((3, 4)), dims=('y',))
    ds = Dataset({'a': r, 'b': r})
    assert list(ds.data_vars) == ['a', 'b']
    with pytest.raises(KeyError):
        ds['c']


def test_recycled_host_blocks():
    """Result arrays come from recycled host blocks: a block returns to the free list only when the last NumPy
    view of it is gone, and the next result of that size reuses it (no fresh page faults)."""
    import gc
    from xrspatial_amd import device
    device.empty_cache()
    a = device.host_empty((1024, 512), np.float32)
    assert a.shape == (1024, 512) and a.dtype == np.float32 and a.flags.c_contiguous and a.flags.writeable
    a[:] = 7
    addr = a.__array_interface__['data'][0]
    view = a[100:200, ::2]
    b = device.host_empty((1024, 512), np.float32)            # `a` still alive: a different block
    assert b.__array_interface__['data'][0] != addr
    del a
    gc.collect()
    assert device._host_pool_bytes == 0 and (view == 7).all()  # the view keeps the block out of the pool
    del view
    gc.collect()
    assert device._host_pool_bytes == 1024 * 512 * 4
    c = device.host_empty((512, 512), np.float64)             # same byte size, other dtype / shape: reused
    assert c.__array_interface__['data'][0] == addr and device._host_pool_bytes == 0
    small = device.host_empty((10, 10), np.float64)           # small results are plain arrays
    assert small.flags.owndata
    del b, c
    gc.collect()
    device.empty_cache()
    assert device._host_pool_bytes == 0 and not device._host_pool


def test_fuse_scope_records_without_touching_the_device():
    """Inside `fuse()` the terrain / focal-mean calls only record; nothing runs until the scope closes."""
    r = raster(np.random.default_rng(1).random((16, 16)).astype(np.float32), res=(2.0, 3.0))
    k = circle_kernel(1, 1, 2)
    assert xa.fused.current() is None
    with pytest.raises(KeyError):                 # an exception inside the scope discards the recorded calls
        with xa.fuse() as scope:
            assert xa.fused.current() is scope
            h = xa.hillshade(r, azimuth=100, angle_altitude=30)
            m = focal.apply(r, k, name='smooth')
            s = xa.slope(r)
            c = xa.curvature(r)
            a = xa.aspect(r)
            with xa.fuse() as inner:              # scopes nest; the inner one is the current one
                assert xa.fused.current() is inner
            assert xa.fused.current() is scope
            raise KeyError('stop')
    assert xa.fused.current() is None and scope.launches == 0
    for res, name in ((h, 'hillshade'), (m, 'smooth'), (s, 'slope'), (c, 'curvature'), (a, 'aspect')):
        assert isinstance(res.data, xa.fused.PendingResult) and res.name == name
        assert res.shape == (16, 16) and res.dims == r.dims and res.attrs == r.attrs
        with pytest.raises(RuntimeError):
            res.values
    # numpy-backed hillshade keeps the reference's result dtype (float64 under NumPy >= 2), everything else float32
    assert h.dtype == (np.float64 if int(np.__version__.split('.')[0]) >= 2 else np.float32) and m.dtype == np.float32
    slots = [call[0] for call in scope._calls]
    assert slots == ['hillshade', 'focal_mean', 'slope', 'curvature', 'aspect']
    assert scope._calls[0][1] == {'light': (100.0, 30.0)} and scope._calls[2][1] == {'cellsize': (2.0, 3.0)}
    # argument validation still happens at the call site
    with xa.fuse():
        with pytest.raises(ValueError):
            xa.slope(r, method='geodetic')
        with pytest.raises(ValueError):
            focal.apply(r, np.ones((2, 2)))
    if not utils.has_hip():                       # closing a non-empty scope needs the GPU: loud failure, no fallback
        with pytest.raises(xa.XrsError):
            with xa.fuse():
                xa.hillshade(r)


def test_every_device_entry_point_refuses_to_run_without_the_gpu():
    """No silent CPU path: with arguments that pass validation, each public function must raise the
    library's own error on a box without a GPU (on a GPU box this test is a no-op)."""
    if utils.has_hip():
        pytest.skip("GPU present")
    r = raster(np.random.default_rng(0).random((16, 16)).astype(np.float32))
    z = raster(np.zeros((16, 16), dtype=np.int32))
    k = np.ones((3, 3))
    calls = [
        lambda: xa.slope(r), lambda: xa.aspect(r), lambda: xa.curvature(r), lambda: xa.hillshade(r),
        lambda: focal.mean(r), lambda: focal.apply(r, k), lambda: focal.focal_stats(r, k),
        lambda: focal.hotspots(r, k), lambda: convolution.convolution_2d(r, k),
        lambda: convolution.convolve_2d(r.data, k), lambda: multispectral.ndvi(r, r),
        lambda: multispectral.evi(r, r, r), lambda: multispectral.savi(r, r),
        lambda: zonal.stats(z, r), lambda: zonal.crosstab(z, z),
    ]
    for call in calls:
        with pytest.raises(xa.XrsError):
            call()


@pytest.mark.parametrize("folded", [False, True])
def test_overlapped_halo_step_order_and_folded_edges(monkeypatch, folded):
    """OverlappedHalo.step on the host side (C ABI: tests/fake_hip.py): interior rows before the wait for the exchange, the
    edges after it -- as two launches or, with `launch_edges`, as ONE call of xrs_raster_pass_edges_f32 -- and either way the
    shard's result is the monolithic pass."""
    import ctypes
    from tests import fake_hip
    from xrspatial_amd import _lib
    from xrspatial_amd.distributed import OverlappedHalo
    fake_hip.install(monkeypatch)
    H, rows, cols = 2, 44, 24
    z = np.random.default_rng(5).normal(100, 5, (rows + 2 * H, cols)).astype(np.float32)
    k = np.ascontiguousarray([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=np.float64)
    out_h, out_f = np.full((rows, cols), -1, np.float32), np.full((rows, cols), -1, np.float32)
    own = z.ctypes.data + H * cols * 4
    log = []

    def launch(first, n, top, bot):
        log.append(("rows", first, n, top, bot))
        off = first * cols * 4
        _lib.call("xrs_raster_pass_f32", own + off, None, None, None, out_h.ctypes.data + off, out_f.ctypes.data + off,
                  k.ctypes.data, 3, 3, None, n, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, top, bot, None)

    def launch_edges(edge, top, bot):
        log.append(("edges", edge, top, bot))
        _lib.call("xrs_raster_pass_edges_f32", own, None, None, None, out_h.ctypes.data, out_f.ctypes.data, k.ctypes.data, 3, 3,
                  None, rows, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, top, bot, edge, None)

    ov = OverlappedHalo(rows, H, edge=16)
    ov.step(lambda stream: log.append(("exchange",)), launch, H, H, launch_edges=launch_edges if folded else None)
    ov.close()
    if folded:
        assert log == [("exchange",), ("rows", 16, 12, 2, 2), ("edges", 16, 2, 2)]
    else:
        assert log == [("exchange",), ("rows", 16, 12, 2, 2), ("rows", 0, 16, 2, 2), ("rows", 28, 16, 2, 2)]
    ref_h, ref_f = np.empty_like(out_h), np.empty_like(out_f)
    _lib.call("xrs_raster_pass_f32", own, None, None, None, ref_h.ctypes.data, ref_f.ctypes.data, k.ctypes.data, 3, 3, None,
              rows, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, H, H, None)
    np.testing.assert_array_equal(out_h, ref_h)
    np.testing.assert_array_equal(out_f, ref_f)


def test_sharded_array_bookkeeping(monkeypatch):
    """ShardedArray without neighbours (world 1) and with a recording transport: when rows are exchanged, how deep,
    what halo_top / halo_bot each rank passes on, and which calls refuse a sharded raster.  (C ABI: tests/fake_hip.py.)"""
    from tests import fake_hip
    from xrspatial_amd import ShardedArray, focal
    from xrspatial_amd.utils import ArrayTypeFunctionMapping
    fake_hip.install(monkeypatch)

    class Recorder:
        def __init__(self, world, rank):
            self.world, self.rank, self.exchanges = world, rank, []

        def halo_exchange(self, base, halo, stream=None):
            self.exchanges.append((base.shape, halo))

    z = np.arange(40 * 12, dtype=np.float32).reshape(40, 12)
    for rank, want in ((0, (0, 2)), (1, (2, 2)), (2, (2, 0))):
        comm = Recorder(3, rank)
        sa = ShardedArray.from_numpy(z, comm, halo_cap=4)
        assert sa.shape == (40, 12) and sa.base.shape == (48, 12) and sa.ptr == sa.base.ptr + 4 * 12 * 4
        assert sa.halos(0) == (0, 0) and comm.exchanges == []             # per-cell operators exchange nothing
        assert sa.halos(2) == want and sa.halos(1) == tuple(min(1, h) for h in want)
        assert comm.exchanges == [((48, 12), 4)]                            # one exchange of halo_cap rows serves both
        sa.touch()
        sa.halos(4)
        assert len(comm.exchanges) == 2
        with pytest.raises(ValueError):
            sa.halos(5)
        out = sa.like(np.float64)
        assert out.shape == sa.shape and out.dtype == np.float64 and out.comm is comm and not out._halo_ok
    with pytest.raises(ValueError):
        ShardedArray.from_numpy(z[:3], Recorder(2, 0), halo_cap=4)         # cannot serve 4 halo rows from 3
    for rank, span in ((0, (0, 14)), (1, (14, 27)), (2, (27, 40))):        # 40 rows over 3 ranks: 14 + 13 + 13
        part = ShardedArray.from_global(z, Recorder(3, rank), halo_cap=4)
        np.testing.assert_array_equal(part.get(), z[span[0]:span[1]])
    solo = ShardedArray.from_numpy(z)
    assert (solo.world, solo.rank) == (1, 0) and solo.halos(3) == (0, 0)
    np.testing.assert_array_equal(ShardedArray.from_global(z).get(), z)
    np.testing.assert_array_equal(solo.get(), z)
    np.testing.assert_array_equal(ShardedArray.from_numpy(z.astype(np.int64)).get(), z.astype(np.float32))
    agg = DataArray(solo, dims=['y', 'x'])
    assert ArrayTypeFunctionMapping(numpy_func=1, hip_func=2, sharded_func=3)(agg) == 3
    with pytest.raises(NotImplementedError):
        ArrayTypeFunctionMapping(numpy_func=1, hip_func=2)(agg)
    # sharded crosstab: both rasters sharded and categorical (round 3); non-integral values or a host-side raster are refused
    with pytest.raises(NotImplementedError):
        zonal.crosstab(DataArray(ShardedArray.from_numpy(z.astype(np.int32)), dims=['y', 'x']),
                       DataArray(ShardedArray.from_numpy(z + np.float32(0.5)), dims=['y', 'x']))
    with pytest.raises(NotImplementedError):
        zonal.crosstab(DataArray(ShardedArray.from_numpy(z.astype(np.int32)), dims=['y', 'x']), DataArray(z, dims=['y', 'x']))
    ct = zonal.crosstab(DataArray(ShardedArray.from_numpy((z.astype(np.int32) // 100)), dims=['y', 'x']),
                        DataArray(ShardedArray.from_numpy(np.floor(z / 7) % 3), dims=['y', 'x']))
    assert list(ct['zone']) == [0, 1, 2, 3, 4] and int(ct.drop(columns='zone').to_numpy().sum()) == z.size
    with pytest.raises(TypeError):
        xa.slope(DataArray(solo, dims=['lat', 'lon'], coords={'lat': np.linspace(1, 2, 40), 'lon': np.linspace(1, 2, 12)}),
                 method='geodesic')


def test_placeholders_and_shards_are_duck_arrays(monkeypatch):
    """xarray wraps an object without converting it only if it looks like a duck array: shape / dtype / ndim plus
    __array_function__ and __array_ufunc__ (xarray.core.utils.is_duck_array).  PendingResult (inside fuse()), ShardedArray
    and ShardedStack must pass that test like DeviceArray does, or real xarray would call np.asarray on them."""
    from xrspatial_amd import fused
    from xrspatial_amd.sharded import ShardedArray, ShardedStack
    from xrspatial_amd.device import DeviceArray
    from tests import fake_hip
    fake_hip.install(monkeypatch)

    def is_duck_array(x):                      # the test xarray applies (core/utils.py)
        return (hasattr(x, "ndim") and hasattr(x, "shape") and hasattr(x, "dtype")
                and hasattr(x, "__array_function__") and hasattr(x, "__array_ufunc__"))

    for cls in (fused.PendingResult, ShardedArray, ShardedStack, DeviceArray):
        assert hasattr(cls, "__array_function__") and hasattr(cls, "__array_ufunc__"), cls
    shard = ShardedArray.from_numpy(np.zeros((4, 8), np.float32))
    assert is_duck_array(shard)
    assert shard.__array_function__(np.sum, (ShardedArray,), (shard,), {}) is NotImplemented
    assert shard.__array_ufunc__(np.add, "__call__", shard, 1) is NotImplemented
    stack = ShardedStack([shard, shard])
    assert is_duck_array(stack) and stack.shape == (2, 4, 8)
    dev = DeviceArray.from_numpy(np.zeros((3, 5), np.float32))
    assert is_duck_array(dev)


def test_user_callables_host_logic(monkeypatch, golden, golden_tables):
    """zonal.stats(stats_funcs={name: callable}) and focal.apply(func=callable): the HOST side of both -- zone slices from
    the grouped values and the valid-cell counts, dtype hand-over, zone_ids selection, DataFrame / back-projection; window
    bands -- over the C-ABI emulation (the device kernels: tests/test_gpu_parity.py).  Reference fixtures:
    xrspatial/tests/test_zonal.py:204-237, 497-544."""
    from tests import fake_hip
    fake_hip.install(monkeypatch)
    funcs = {'double_sum': lambda v: v.sum() * 2, 'range': lambda v: v.max() - v.min()}
    zones = raster(np.nan_to_num(golden["zonal_zones"], nan=-9).astype(np.int32))
    values = raster(golden["zonal_values"])
    nodata, ids, exp = (golden_tables["zonal_custom__%d" % i] for i in range(3))
    df = zonal.stats(zones, values, zone_ids=ids, stats_funcs=funcs, nodata_values=nodata)
    assert list(df.columns) == ['zone', 'double_sum', 'range'] and df['zone'].tolist() == exp['zone']
    for col in ('double_sum', 'range'):
        np.testing.assert_allclose(df[col], exp[col], rtol=1e-12)
    da = zonal.stats(zones, values, zone_ids=ids, stats_funcs=funcs, nodata_values=nodata, return_type='xarray.DataArray')
    np.testing.assert_allclose(np.asarray(da.data), golden["zonal_custom_da__2"], equal_nan=True)
    # order statistics, integer values handed over in their own dtype, a zone without valid cells
    rng = np.random.default_rng(1)
    zz = rng.integers(0, 9, size=(40, 61)).astype(np.int32)
    vv = rng.integers(-20, 21, size=zz.shape).astype(np.int16)
    vv[zz == 4] = 3
    seen = []
    got = zonal.stats(raster(zz), raster(vv), nodata_values=3,
                      stats_funcs={'median': np.median, 'n': lambda v: (seen.append(v.dtype), v.size)[1]})
    assert set(seen) == {np.dtype(np.int16)}
    for i, z in enumerate(np.unique(zz)):
        cell = vv[(zz == z) & (vv != 3)]
        if z == 4:
            assert np.isnan(got['median'][i]) and np.isnan(got['n'][i])
        else:
            assert got['median'][i] == np.median(cell) and got['n'][i] == cell.size
    with pytest.raises(ValueError):
        zonal.stats(raster(zz), raster(vv), stats_funcs={'bad': 'median'})

    # focal.apply with a callable, several bands
    z = rng.normal(size=(23, 17)).astype(np.float32)
    z[3, 4] = np.nan
    k = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], float)
    monkeypatch.setattr(focal, "_WINDOW_BAND_BYTES", 17 * 9 * 4 * 4)          # 4 rows per band
    got = focal.apply(raster(z), k, func=lambda w: np.nansum(w) + np.isnan(w).sum())
    pad = np.full((25, 19), np.nan, np.float32)
    pad[1:-1, 1:-1] = z
    want = np.zeros_like(z)
    for y in range(23):
        for x in range(17):
            w = np.where(k == 1, pad[y:y + 3, x:x + 3], np.nan).astype(np.float32)
            want[y, x] = np.nansum(w) + np.isnan(w).sum()
    np.testing.assert_array_equal(got.data, want)
    assert got.data.dtype == np.float32


def test_zonal_one_pass_window_host_logic(monkeypatch):
    """zonal._one_pass_partials on the CPU stand-in of the C ABI: the window guessed from the strided sample, ids listed like
    np.unique does (a zone with invalid cells only is there with count 0), the overflow flag and the too-wide range sending
    the call to the two-pass route -- and zonal.stats giving the oracle's table either way."""
    from tests import fake_hip
    from oracle import xrs_oracle as orc
    from xrspatial_amd import zonal as zmod
    import xrspatial_amd as xs
    fake_hip.install(monkeypatch)
    rng = np.random.default_rng(3)
    rows, cols = 120, 200
    zones = (5000 + 3 * (((np.arange(rows)[:, None] // 11) * 7 + np.arange(cols)[None, :] // 23) % 60)).astype(np.int32)
    vals = (rng.normal(100, 5, (rows, cols))).astype(np.float32)
    vals[rng.random(vals.shape) < 0.02] = np.nan
    vals[zones == zones[60, 100]] = np.nan                     # a zone without one valid cell
    stats = ['mean', 'max', 'min', 'sum', 'std', 'var', 'count']
    zd, vd = xs.DeviceArray.from_numpy(zones), xs.DeviceArray.from_numpy(vals)
    one = zmod._one_pass_partials(zd, vd, None)
    assert one is not None
    np.testing.assert_array_equal(one[0], np.unique(zones))
    assert one[1][list(one[0]).index(int(zones[60, 100]))] == 0
    got = xs.zonal_stats(xs.DataArray(zd), xs.DataArray(vd), stats_funcs=stats)
    want = orc.zonal_stats(zones, vals, stats_funcs=stats)
    np.testing.assert_array_equal(got['zone'].to_numpy(), want['zone'])
    for col in stats:
        np.testing.assert_allclose(got[col].to_numpy(), want[col], rtol=1e-6, equal_nan=True, err_msg=col)
    # one stray id the sample cannot see -> overflow -> two passes, same table; ids spread wider than any window -> no attempt
    stray = zones.copy()
    stray[7, 9] = 3_000_000
    assert zmod._one_pass_partials(xs.DeviceArray.from_numpy(stray), vd, None) is None
    got = xs.zonal_stats(xs.DataArray(xs.DeviceArray.from_numpy(stray)), xs.DataArray(vd), stats_funcs=['count', 'mean'])
    want = orc.zonal_stats(stray, vals, stats_funcs=['count', 'mean'])
    np.testing.assert_array_equal(got['zone'].to_numpy(), want['zone'])
    np.testing.assert_allclose(got['count'].to_numpy(), want['count'], equal_nan=True)
    wide = (zones * 200).astype(np.int32)
    assert zmod._one_pass_partials(xs.DeviceArray.from_numpy(wide), vd, None) is None


def _second_largest(w):
    v = np.sort(w[np.isfinite(w)])
    return v[-2] if v.size > 1 else np.nan


def test_dask_block_functions_pickle():
    """What the dask slot hands to map_overlap / map_blocks must survive cloudpickle (dask's process and distributed
    schedulers ship it to workers): a module-level function + functools.partial, the lock looked up where the block runs."""
    import functools
    import pickle
    import cloudpickle
    from xrspatial_amd import utils, focal as xfocal
    k = np.ones((3, 3))

    def nested_runner(data, kernel, stat):                     # (the public functions pass closures like this one)
        return xfocal._focal_stats_hip(data, kernel, [stat])[stat]
    for block_func, args in ((xfocal._apply_callable, (k, _second_largest)), (nested_runner, (k, 'mean'))):
        blob = cloudpickle.dumps(functools.partial(utils._run_block, block_func, args, {}))
        back = pickle.loads(blob)
        assert back.func.__name__ == '_run_block' and back.args[1][0].shape == (3, 3)
    assert b'_thread' not in blob


def test_dask_slot_runs_block_by_block(monkeypatch):
    """The dask slot of the public functions (utils.py: dask_overlap / dask_blocks): the reference wraps its numpy runners in
    map_overlap(depth, boundary=nan) / map_blocks (slope.py:86-97, aspect.py:151-160, curvature.py:56-59, hillshade.py:42-45,
    focal.py:70-75 and 329-340, convolution.py:316-327, multispectral.py:845-848) and so does this package, around ITS numpy
    runners.  dask is not installable here: `tests/fake_dask.py` supplies the four calls the slot makes, with dask's overlap
    semantics, and `tests/fake_hip.py` answers the C ABI.  A dask-backed DataArray must give exactly what the numpy-backed one
    gives -- stencils across chunk boundaries included -- and must have been computed chunk by chunk."""
    from tests import fake_dask, fake_hip
    from xrspatial_amd import utils, focal as xfocal, convolution
    import xrspatial_amd as xs
    fake_hip.install(monkeypatch)
    monkeypatch.setattr(utils, "da", fake_dask)
    monkeypatch.setattr(xfocal, "da", fake_dask)
    rng = np.random.default_rng(5)
    z = (100 + np.cumsum(rng.normal(0, 1, (37, 53)), axis=1) + rng.normal(0, 3, (37, 53))).astype(np.float32)
    z[7, 9] = np.nan
    z[20:23, 30:33] = np.nan
    coords = {'y': np.arange(37)[::-1] * 2.0, 'x': np.arange(53) * 2.0}
    host = xs.DataArray(z, dims=['y', 'x'], coords=coords, attrs={'res': (2.0, 2.0)})

    def lazy(values=z, chunks=(16, 20)):
        return xs.DataArray(fake_dask.from_array(values, chunks), dims=['y', 'x'], coords=coords, attrs={'res': (2.0, 2.0)})

    k5 = convolution.circle_kernel(2, 2, 4)
    cases = {
        'slope': lambda a: xs.slope(a), 'aspect': lambda a: xs.aspect(a), 'curvature': lambda a: xs.curvature(a),
        'hillshade': lambda a: xs.hillshade(a, 315, 30),
        'focal.mean x2': lambda a: xfocal.mean(a, passes=2),
        'focal.apply': lambda a: xfocal.apply(a, k5),
        'focal_stats': lambda a: xfocal.focal_stats(a, k5, stats_funcs=['max', 'mean', 'std']),
        'convolution_2d': lambda a: convolution.convolution_2d(a, k5 / k5.sum()),
        # a user callable: the reference's _apply_dask_numpy runs it per chunk (focal.py:329-340), never on the whole raster
        'focal.apply(callable)': lambda a: xfocal.apply(a, k5, func=_second_largest),
    }
    for name, fn in cases.items():
        want = fn(host)
        arg = lazy()
        got = fn(arg)
        assert isinstance(got.data, fake_dask.Array), name                      # lazy in, lazy out -- like upstream
        assert got.dims == want.dims and got.name == want.name, name
        np.testing.assert_array_equal(got.data.compute(), np.asarray(want.data), err_msg=name)
        assert got.data.compute().dtype == np.asarray(want.data).dtype, name
        seen = got.data.blocks_seen
        assert len(seen) >= 6 and max(s[0] for s in seen) < 37 and max(s[1] for s in seen) < 53, (name, seen)   # 3 x 3 chunks, never the whole
    # an integer raster: cast to float32 before the NaN boundary is attached (slope.py:89)
    zi = (np.nan_to_num(z, nan=0.0) * 10).astype(np.int32)
    np.testing.assert_array_equal(xs.slope(lazy(zi)).data.compute(),
                                  np.asarray(xs.slope(xs.DataArray(zi.astype(np.float32), dims=['y', 'x'], coords=coords,
                                                                   attrs={'res': (2.0, 2.0)})).data))
    # per-cell indices: map_blocks over equally chunked bands
    nir = rng.random((37, 53)).astype(np.float32)
    red = rng.random((37, 53)).astype(np.float32)
    want = xs.ndvi(xs.DataArray(nir, dims=['y', 'x']), xs.DataArray(red, dims=['y', 'x']))
    got = xs.ndvi(lazy(nir), lazy(red))
    assert isinstance(got.data, fake_dask.Array)
    np.testing.assert_array_equal(got.data.compute(), want.data)
    # hotspots: the two global scalars first (eagerly), then convolution + z-score + classes per chunk (focal.py:940-976)
    spots = z.copy()
    spots[:12, :14] += 60.0                      # a hot corner and a cold one: several confidence classes in the result
    spots[-12:, -14:] -= 60.0
    want = xfocal.hotspots(xs.DataArray(spots, dims=['y', 'x'], coords=coords), k5)
    got = xfocal.hotspots(lazy(spots), k5)
    assert isinstance(got.data, fake_dask.Array) and got.data.compute().dtype == np.int8 and got.attrs['unit'] == '%'
    np.testing.assert_array_equal(got.data.compute(), np.asarray(want.data))
    assert len(np.unique(np.asarray(want.data))) > 1
    with pytest.raises(ZeroDivisionError):
        xfocal.hotspots(lazy(np.full((37, 53), 7.0, np.float32)), k5)
    # zonal.stats: per-block partial sums combined like _stats_dask_numpy (zonal.py:181-277); majority is not a block statistic
    from xrspatial_amd import zonal as xzonal
    monkeypatch.setattr(xzonal, "is_dask", lambda d: isinstance(d, fake_dask.Array))
    zvals = ((np.arange(37)[:, None] // 9) * 3 + np.arange(53)[None, :] // 20).astype(np.int32)
    zones = xs.DataArray(fake_dask.from_array(zvals, (16, 20)), dims=['y', 'x'])
    seven = ['mean', 'max', 'min', 'sum', 'std', 'var', 'count']
    lazy_vals = lazy()
    got = xzonal.stats(zones, lazy_vals, stats_funcs=seven + ['majority'], zone_ids=[0, 4, 7, 99], nodata_values=float(z[0, 0]))
    want = xzonal.stats(xs.DataArray(zvals, dims=['y', 'x']), host, stats_funcs=seven, zone_ids=[0, 4, 7, 99], nodata_values=float(z[0, 0]))
    assert list(got.columns) == ['zone'] + seven and got['zone'].tolist() == [0, 4, 7]
    for col in got.columns:
        np.testing.assert_allclose(np.asarray(got[col], dtype=np.float64), np.asarray(want[col], dtype=np.float64), rtol=1e-12, err_msg=col)
    assert len(lazy_vals.data.blocks_seen) == 9 and max(lazy_vals.data.blocks_seen) == (16, 20)       # block by block
    with pytest.raises(ValueError):
        xzonal.stats(zones, lazy(), stats_funcs={'n': len})
    with pytest.raises(ValueError):
        xzonal.stats(zones, lazy(), return_type='xarray.DataArray')
    with pytest.raises(ValueError):
        xzonal.stats(xs.DataArray(fake_dask.from_array(zvals, (10, 20)), dims=['y', 'x']), lazy(), stats_funcs=seven)
    # crosstab, 2-D values: per-block count tables added (zonal.py:813-916), then the same frame as the numpy backend
    cats = ((np.arange(37)[:, None] * 7 + np.arange(53)[None, :] * 3) % 5 + 10).astype(np.float64)
    cats[3, 4] = np.nan
    lazy_cats = xs.DataArray(fake_dask.from_array(cats, (16, 20)), dims=['y', 'x'])
    for kw in ({}, {'nodata_values': 12}, {'zone_ids': [0, 4, 99], 'cat_ids': [10, 14], 'agg': 'percentage'}):
        got = xzonal.crosstab(zones, lazy_cats, **kw)
        want = xzonal.crosstab(xs.DataArray(zvals, dims=['y', 'x']), xs.DataArray(cats, dims=['y', 'x']), **kw)
        assert list(got.columns) == list(want.columns)
        np.testing.assert_array_equal(got.to_numpy(dtype=np.float64), want.to_numpy(dtype=np.float64))
