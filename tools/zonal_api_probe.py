"""Where the public zonal.stats call spends its time beyond the kernel (cProfile over 30 calls on resident 16384^2 rasters)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs  # noqa: E402
from tests import synth  # noqa: E402

n = int(os.environ.get("N", "16384"))
zones = xs.DeviceArray.from_numpy(synth.block_zones(n, n) + np.int32(5000))
vals = xs.DeviceArray.from_numpy(np.tile(synth.asv_dem(2048, n), (n // 2048, 1)))
za, va = xs.DataArray(zones, dims=['y', 'x']), xs.DataArray(vals, dims=['y', 'x'])
seven = ['mean', 'max', 'min', 'sum', 'std', 'var', 'count']
for _ in range(3):
    xs.zonal_stats(za, va, stats_funcs=seven)
ts = []
for _ in range(20):
    t = time.perf_counter(); xs.zonal_stats(za, va, stats_funcs=seven); ts.append((time.perf_counter() - t) * 1e3)
print(f"api ms per call: median {np.median(ts):.3f} min {min(ts):.3f}")
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    xs.zonal_stats(za, va, stats_funcs=seven)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
