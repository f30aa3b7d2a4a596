"""rocprofv3 --kernel-trace --stats target: the seven statistics over annulus_kernel(1, 1, 10, 6) on the benchmark DEM, clean (CLEAN=1)
or with 0.1 % scattered nodata -- which launch pays for the nodata.   cd /tmp && rocprofv3 --kernel-trace --stats -- python tools/ann_nan_prof.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xrspatial_amd as xs
from xrspatial_amd import focal
from xrspatial_amd.convolution import annulus_kernel
from tests import synth
z = synth.asv_dem(16384, 16384).copy()
if not os.environ.get("CLEAN"):
    z[np.random.default_rng(7).random(z.shape) < 0.001] = np.nan
A = xs.DataArray(xs.DeviceArray.from_numpy(z), dims=["y", "x"], attrs={"res": (1.0, 1.0)})
k = annulus_kernel(1, 1, 10, 6)
for _ in range(6):
    focal.focal_stats(A, k)
