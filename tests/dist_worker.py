"""Worker for tests/test_distributed_cpu.py: one rank of a gloo process group (CPU only).

Each rank owns a contiguous block of rows of a seeded raster, runs the row-shard protocol of the
multi-GPU path (xrspatial_amd.distributed: shard_rows / shard_halos / halo exchange / zonal partial
all-reduce) with the CPU oracle standing in for the HIP kernels, and writes its slice of the results."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(outdir):
    import torch.distributed as dist
    from oracle import xrs_oracle as orc
    from tests import synth
    from tests.host_transport import halo_exchange_host, zonal_allreduce_host
    from xrspatial_amd.distributed import shard_halos, shard_rows

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    H, W, HALO = 61, 48, 2
    full = synth.smooth_dem((H, W), nan_frac=0.02)           # every rank can regenerate the raster
    zones = synth.block_zones(H, W, n_zones=7, block=5)
    y0, y1 = shard_rows(H, world, rank)
    ht, hb = shard_halos(world, rank, HALO)
    rows = y1 - y0

    # shard buffer with spare halo rows, owned rows in the middle; halos start as garbage
    buf = np.full((rows + 2 * HALO, W), -12345.0, dtype=np.float32)
    buf[HALO:HALO + rows] = full[y0:y1]
    halo_exchange_host(dist, buf, HALO)

    # what a kernel sees with (halo_top, halo_bot): the owned rows plus ht rows above and hb below
    view = buf[HALO - ht: HALO + rows + hb]
    k = orc.circle_kernel(1, 1, 2)
    slope = orc.slope(view, 30.0, 30.0)[ht:ht + rows]
    hill = orc.hillshade(view)[ht:ht + rows]
    focal = orc.focal_apply(view, k, 'mean')[ht:ht + rows]
    conv = orc.convolve_2d(view, k)[ht:ht + rows]
    # 3x3 ops only see 1 halo row: true raster edges must come out as the NaN border
    # zonal partials on the owned rows, then the all-reduce
    z, v = zones[y0:y1].ravel(), full[y0:y1].ravel().astype(np.float64)
    ok = np.isfinite(v)
    nz = 7
    cnt = np.bincount(z[ok], minlength=nz).astype(np.uint64)
    s1 = np.bincount(z[ok], weights=v[ok], minlength=nz)
    s2 = np.bincount(z[ok], weights=v[ok] ** 2, minlength=nz)
    mn = np.full(nz, np.inf)
    mx = np.full(nz, -np.inf)
    np.minimum.at(mn, z[ok], v[ok])
    np.maximum.at(mx, z[ok], v[ok])
    cnt, s1, s2, mn, mx = zonal_allreduce_host(dist, cnt, s1, s2, mn, mx)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), y0=y0, y1=y1, slope=slope, hill=hill, focal=focal,
             conv=conv, cnt=cnt, s1=s1, s2=s2, mn=mn, mx=mx)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
