"""Worker for test_gpu_parity.py::test_sharded_api_*: one rank of a row-sharded run through the PUBLIC API.

Every rank builds a ShardedArray from its rows of a seeded raster and calls the same functions a single-GPU user
calls (slope, hillshade, focal.mean, focal.apply, convolution_2d, ndvi, fuse(), zonal.stats); the neighbours' rows
travel through the shard's transport -- tests/host_transport.HostTransport over gloo by default (which also works when all ranks
share ONE GPU, as on the test box), distributed.Comm (RCCL) with XRS_TEST_TRANSPORT=rccl on a multi-GPU node.
test_distributed_cpu.py runs the same worker without a GPU (XRS_TEST_FAKE_HIP=1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(outdir):
    if os.environ.get("XRS_TEST_FAKE_HIP") == "1":
        # CPU suite: the C ABI is answered by tests/fake_hip.py (oracle arithmetic), the host logic under test is real
        from tests import fake_hip
        fake_hip.install()
    import xrspatial_amd as xs
    from tests import synth
    from xrspatial_amd import focal, zonal
    from xrspatial_amd.convolution import circle_kernel, convolution_2d
    from xrspatial_amd.distributed import Comm, shard_rows
    from tests.host_transport import HostTransport
    from xrspatial_amd.sharded import ShardedArray
    import torch.distributed as dist

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = Comm.from_torch_distributed(dist) if os.environ.get("XRS_TEST_TRANSPORT") == "rccl" else HostTransport(dist)
    H, W = 150, 300
    full = synth.smooth_dem((H, W), nan_frac=0.01)
    red = synth.smooth_dem((H, W), seed=5) + 50.0
    zones_full = synth.block_zones(H, W, n_zones=9, block=11).astype(np.int32)
    y0, y1 = shard_rows(H, world, rank)

    def shard(a, **kw):
        return xs.DataArray(ShardedArray.from_numpy(a[y0:y1], comm, **kw), dims=['y', 'x'], attrs={'res': (30.0, 30.0)})

    dem = shard(full)
    out = {}
    out['slope'] = xs.slope(dem).data.get()
    out['aspect'] = xs.aspect(dem).data.get()
    out['curvature'] = xs.curvature(dem).data.get()
    out['hillshade'] = xs.hillshade(dem).data.get()
    out['mean3'] = focal.mean(dem, passes=3).data.get()
    k5 = circle_kernel(1, 1, 2)
    k7 = circle_kernel(1, 1, 3)
    out['apply5'] = focal.apply(dem, k5).data.get()
    out['max7'] = focal.apply(dem, k7, focal._calc_max).data.get()
    out['conv5'] = convolution_2d(dem, k5).data.get()
    out['ndvi'] = xs.ndvi(dem, shard(red)).data.get()
    stats = focal.focal_stats(dem, k5, stats_funcs=['mean', 'max', 'std'])
    assert stats.dims == ('stats', 'y', 'x') and stats.shape == (3, y1 - y0, W)
    out['stats5'] = stats.data.get()
    out['hot7'] = focal.hotspots(dem, k7).data.get()
    out['chain'] = focal.mean(xs.slope(dem)).data.get()                 # a result is a shard again: its halos get exchanged
    with xs.fuse() as scope:
        f_h, f_s, f_m = xs.hillshade(dem), xs.slope(dem), focal.apply(dem, k5)
    assert scope.launches == 1
    out['fused_hillshade'], out['fused_slope'], out['fused_apply5'] = f_h.data.get(), f_s.data.get(), f_m.data.get()
    table = zonal.stats(shard(zones_full), dem, stats_funcs=['mean', 'max', 'min', 'sum', 'std', 'var', 'count'])
    for col in table.columns:
        out['zonal_' + col] = np.asarray(table[col])
    # return_type='xarray.DataArray': every rank back-projects the agreed table onto its own rows (planes are shards again)
    back = zonal.stats(shard(zones_full), dem, zone_ids=[1, 4, 7, 99], stats_funcs=['mean', 'count', 'max'],
                       return_type='xarray.DataArray')
    assert back.dims == ('stats', 'y', 'x') and back.shape == (3, y1 - y0, W) and [str(v) for v in np.asarray(back.coords['stats'])] == ['mean', 'count', 'max']
    out['zonal_back'] = back.data.get()
    # focal.apply with a user callable: windows are gathered across the shard boundary (halo rows), the callable runs here

    def second_largest(w):
        v = np.sort(w[np.isfinite(w)])
        return v[-2] if v.size > 1 else np.nan

    called = focal.apply(dem, k5, func=second_largest)
    assert isinstance(called.data, ShardedArray) and called.data.dtype == np.float32
    out['call5'] = called.data.get()
    # crosstab: every rank counts its rows, the (zones x categories) tables are added, every rank holds the whole frame
    cats_full = ((np.arange(H)[:, None] * 7 + np.arange(W)[None, :] * 3) % 5 + 10).astype(np.int32)
    ct = zonal.crosstab(shard(zones_full), shard(cats_full), nodata_values=12)
    out['crosstab_cols'] = np.asarray([int(c) if c != 'zone' else -1 for c in ct.columns])
    out['crosstab'] = ct.to_numpy(dtype=np.float64)
    out['crosstab_pct'] = zonal.crosstab(shard(zones_full), shard(cats_full), zone_ids=[1, 4, 7], cat_ids=[10, 14],
                                         agg='percentage').to_numpy(dtype=np.float64)
    # what a sharded raster cannot do fails loudly
    for bad in (lambda: zonal.stats(shard(zones_full), dem), lambda: focal.apply(shard(full, halo_cap=2), k7),
                lambda: focal.apply(shard(full, halo_cap=2), k7, func=second_largest),
                lambda: zonal.stats(shard(zones_full), dem, stats_funcs={'n': len}),
                lambda: zonal.crosstab(shard(zones_full), dem),
                lambda: xs.slope(xs.DataArray(dem.data, dims=['lat', 'lon'], coords={'lat': np.linspace(1, 2, y1 - y0),
                                                                                   'lon': np.linspace(1, 2, W)}), method='geodesic')):
        try:
            bad()
        except (NotImplementedError, TypeError, ValueError):
            continue
        raise AssertionError("an unsupported sharded call did not raise")
    # the transport itself, on a short float64 shard (fewer than 2 * halo_cap rows: whole-shard staging on the host path)
    mine = np.arange(20 * 8, dtype=np.float64).reshape(20, 8)
    small = ShardedArray.from_numpy(mine + 1000 * rank, comm)
    assert small.halos(3) == ((3 if rank > 0 else 0), (3 if rank < world - 1 else 0))
    base = small.base.get()
    np.testing.assert_array_equal(base[16:36], mine + 1000 * rank)
    if rank > 0:
        np.testing.assert_array_equal(base[:16], mine[4:] + 1000 * (rank - 1))
    if rank < world - 1:
        np.testing.assert_array_equal(base[36:], mine[:16] + 1000 * (rank + 1))
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), y0=y0, y1=y1, **out)
    dist.barrier()
    if hasattr(comm, "destroy"):
        comm.destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
