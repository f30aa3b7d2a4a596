"""BASELINE config 5 on ONE GPU: zonal.stats over a 32768 x 32768 float32 raster with 1000 int32 zones,
device-resident inputs.  Prints the partial-sum kernel time (8 B/cell read-only) and the whole
`zonal.stats` call (zone indexing on the device, partials, majority excluded / included)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import xrspatial_amd as xs  # noqa: E402
from tests import synth  # noqa: E402
from tools.kbench import Timer  # noqa: E402
from xrspatial_amd import _lib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    _lib.require_device()
    L = _lib.call
    cells = n * n
    vals = xs.DeviceArray((n, n), np.float32)
    zones = xs.DeviceArray((n, n), np.int32)
    band = 2048
    v0 = synth.asv_dem(band, n, y0=0, total_rows=n)
    v0[np.random.default_rng(0).random(v0.shape) < 0.001] = np.nan
    for y0 in range(0, n, band):
        z = synth.block_zones(band, n, y0=y0)
        L("xrs_memcpy_h2d", zones.ptr + y0 * n * 4, z.ctypes.data, z.nbytes, None)
        L("xrs_memcpy_h2d", vals.ptr + y0 * n * 4, v0.ctypes.data, v0.nbytes, None)
        L("xrs_stream_sync", None)
    nz = 1000
    zc = xs.DeviceArray((nz,), np.uint64)
    zs, zq = xs.DeviceArray((nz,), np.float64), xs.DeviceArray((nz,), np.float64)
    zmn, zmx = xs.DeviceArray((nz,), np.float32), xs.DeviceArray((nz,), np.float32)

    def partials():
        L("xrs_zonal_init", zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, nz, None)
        L("xrs_zonal_partials_f32", zones.ptr, vals.ptr, cells, nz, 0.0, 0, 0.0, zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, None)

    med, mn = Timer().time(partials, 10, warmup=2)
    print(f"zonal partials kernel ({n}x{n}, 1000 zones): {med:.3f} ms = {cells * 8 / med / 1e6:.0f} GB/s, "
          f"{cells / med / 1e3:.0f} Mcells/s")
    zagg, vagg = xs.DataArray(zones, dims=['y', 'x']), xs.DataArray(vals, dims=['y', 'x'])
    names = ['mean', 'max', 'min', 'sum', 'std', 'var', 'count']
    xs.zonal_stats(zagg, vagg, stats_funcs=names)
    t0 = time.perf_counter()
    df = xs.zonal_stats(zagg, vagg, stats_funcs=names)
    t1 = time.perf_counter()
    print(f"zonal.stats (7 partial-sum stats, device-resident, zone ids mapped on the device): {1e3 * (t1 - t0):.1f} ms "
          f"= {cells / (t1 - t0) / 1e6:.0f} Mcells/s; zones={len(df)}, total count={int(df['count'].sum())}")
    assert int(df['count'].sum()) == int(np.isfinite(v0).sum()) * (n // band)
    if n <= 32768:
        t0 = time.perf_counter()
        df = xs.zonal_stats(zagg, vagg)
        t1 = time.perf_counter()
        print(f"zonal.stats (all 8 default stats incl. majority by device sort): {1e3 * (t1 - t0):.1f} ms")


if __name__ == "__main__":
    main()
