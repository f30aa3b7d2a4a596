"""rocprofv3 --kernel-trace --stats target: the large-window moments / extrema kernels on the benchmark DEM with 0.1 % nodata
(per-kernel durations of focal_stats with 25x25 circle and box masks).  cd /tmp && rocprofv3 --kernel-trace --stats -- python tools/mom_nan_prof.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xrspatial_amd as xs
from xrspatial_amd import focal
from xrspatial_amd.convolution import circle_kernel
from tests import synth
z = synth.asv_dem(16384, 16384).copy()
z[np.random.default_rng(7).random(z.shape) < 0.001] = np.nan
A = xs.DataArray(xs.DeviceArray.from_numpy(z), dims=["y", "x"], attrs={"res": (1.0, 1.0)})
k = circle_kernel(1, 1, 12)
for _ in range(6):
    focal.focal_stats(A, k, stats_funcs=['mean', 'var', 'std'])
    focal.focal_stats(A, k)
    focal.focal_stats(A, np.ones((25, 25)))
xs.synchronize() if hasattr(xs, "synchronize") else None
