"""Debug probe for the small-window strip walker: one raster with interior tiles, through the public API, vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import xrspatial_amd as xs
from oracle import c_oracle as corc
from tests import synth
from xrspatial_amd.convolution import circle_kernel
from xrspatial_amd.focal import focal_stats

which = sys.argv[1] if len(sys.argv) > 1 else "all"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shape = (700, 1500)
z = synth.smooth_dem(shape, seed=3)
if which == "nan":
    z[300, 700] = np.nan
    z[400:420, 900:960] = np.nan
    z[500, 100] = np.inf
    z[200:240, 300:380] = 777.25
k = circle_kernel(1, 1, K // 2) if "box" not in which else np.ones((K, K))
stats = ['mean', 'max', 'min', 'range', 'std', 'var', 'sum'] if which != "mean_var" else ['mean', 'var']
agg = xs.DataArray(z, dims=['y', 'x'])
print("calling", which, K, flush=True)
got = focal_stats(agg, k, stats_funcs=stats)
print("returned", flush=True)
with np.errstate(all='ignore'):
    for i, st in enumerate(stats):
        want = corc.focal_apply(z, k, st, nthreads=8)
        g = got.data[i]
        bad = ~((g == want) | (np.isnan(g) & np.isnan(want)) | (np.abs(g - want) <= 2e-6 * np.abs(want)))
        print(f"{st:6s} mismatches {int(bad.sum()):8d}  max rel {np.nanmax(np.where(np.isfinite(want) & (want != 0), np.abs(g - want) / np.abs(want), 0)):.3e}", flush=True)
        if bad.any():
            ys, xs_ = np.nonzero(bad)
            print("   first at", ys[0], xs_[0], g[ys[0], xs_[0]], want[ys[0], xs_[0]], " rows", np.unique(ys)[:10], " cols", np.unique(xs_)[:10])
