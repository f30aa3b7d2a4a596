"""rocprofv3 --kernel-trace --stats target: the large-window kernels on a raster whose left third is one nodata region
(tools/nan_probe.py's "left third NaN" case): which kernel pays for the region's rim.
    cd /tmp && rocprofv3 --kernel-trace --stats -- python tools/mom_region_prof.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xrspatial_amd as xs
from xrspatial_amd import focal
from xrspatial_amd.convolution import circle_kernel
from tests import synth
n = 16384
z = synth.asv_dem(n, n).copy()
mode = os.environ.get("REGION", "third")
if mode == "third":
    z[:, : n // 3] = np.nan
elif mode == "aligned":            # the region ends on a wave-tile boundary (column 5376 = 42 * 128)
    z[:, :5376] = np.nan
elif mode == "rows":               # the top third
    z[: n // 3, :] = np.nan
elif mode == "scatter5":           # 5 % of the cells, scattered: every tile is dense with nodata
    synth.scatter_nodata(z, 0.05, 99)
elif mode == "scatter1":           # 1 %
    synth.scatter_nodata(z, 0.01, 99)
A = xs.DataArray(xs.DeviceArray.from_numpy(z), dims=["y", "x"], attrs={"res": (1.0, 1.0)})
k = circle_kernel(1, 1, 12)
for _ in range(4):
    focal.focal_stats(A, k, stats_funcs=['mean', 'var', 'std'])
    focal.focal_stats(A, k, stats_funcs=['mean'])
