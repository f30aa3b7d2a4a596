"""Debug probe: where does a moments result differ from the oracle on the 'holes' raster of
test_third_generation_walkers_interior_and_rim_tiles?  usage: mom_holes_probe.py [radius] [circle|box] [rows] [cols]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs  # noqa: E402
from oracle import c_oracle as corc  # noqa: E402
from tests import synth  # noqa: E402
from xrspatial_amd.convolution import circle_kernel  # noqa: E402
from xrspatial_amd.focal import focal_stats  # noqa: E402

radius = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kind = sys.argv[2] if len(sys.argv) > 2 else 'circle'
shape = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (300, 700)
K = 2 * radius + 1
k = circle_kernel(1, 1, radius) if kind == 'circle' else np.ones((K, K))
holes = synth.smooth_dem(shape, seed=radius).copy()
holes[200, 300] = np.nan
holes[150:150 + 2 * K + 3, 500:500 + 2 * K + 5] = np.nan
holes[260, 200] = np.inf
holes[170, 650] = -np.inf
holes[220:220 + 3 * K, 30:30 + 3 * K] = 1234.5
stats = tuple(os.environ.get('PROBE_STATS', 'mean,var,std').split(','))
got = focal_stats(xs.DataArray(holes, dims=['y', 'x']), k, stats_funcs=list(stats)).data
with np.errstate(all='ignore'):
    want = {s: corc.focal_apply(holes, k, s, nthreads=8) for s in stats}
for i, s in enumerate(stats):
    g, w = np.asarray(got[i], np.float64), np.asarray(want[s], np.float64)
    with np.errstate(all='ignore'):
        bad = (np.isnan(g) != np.isnan(w)) | (~np.isnan(g) & ~np.isnan(w) & (g != w) & ~(np.abs(g - w) <= 1e-5 * np.abs(w)))
    ys, xs_ = np.nonzero(bad)
    print(s, "mismatches:", len(ys))
    for y, x in list(zip(ys, xs_))[:12]:
        print("   ", y, x, "got", g[y, x], "want", w[y, x])
