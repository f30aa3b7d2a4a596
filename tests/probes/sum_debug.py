"""Where does the window sum of the 'holes' raster of test_separable_box_walk[25] leave its tolerance?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import xrspatial_amd as xs
from oracle import c_oracle as corc
from tests import synth
from xrspatial_amd.focal import focal_stats
K = 25
k = np.ones((K, K))
rows, cols = 700, 1500
z2 = synth.smooth_dem((rows, cols), seed=K).copy()
z2[100, 300] = np.nan
z2[400, 1200] = np.inf
z2[500:560, 600:700] = 1234.567
z2[0:40, 0:60] = -5.25
for env in ("1", "0"):
    os.environ["XRS_MOM_RESCUE"] = env
    got = np.asarray(focal_stats(xs.DataArray(z2, dims=['y', 'x']), k, stats_funcs=['mean', 'std', 'var', 'sum']).data)[3]
    want = corc.focal_apply(z2, k, 'sum', nthreads=8)
    z64 = np.nan_to_num(z2.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    absz = np.abs(z64).astype(np.float32)
    sum_abs = corc.focal_apply(absz, k, 'sum', nthreads=8).astype(np.float64)
    n = 625.0
    bound = (n - 1) * 2.0 ** -24 * sum_abs
    fin = np.isfinite(got) & np.isfinite(want)
    d = np.where(fin, np.abs(got.astype(np.float64) - want.astype(np.float64)), 0)
    ref = np.abs(want.astype(np.float64))
    tol = np.where(ref >= 0.1 * sum_abs, 1e-5 * ref, np.maximum(1e-5 * ref, 1.01 * bound + 1e-30))
    over = np.where(fin, d - tol, -1)
    ys, xs_ = np.nonzero(over > 0)
    print(f"rescue={env}: {len(ys)} windows beyond tolerance")
    # the exactly rounded sum for comparison
    for y, x in list(zip(ys, xs_))[:10]:
        win = z2[max(0, y - 12):y + 13, max(0, x - 12):x + 13].astype(np.float64)
        exact = np.nansum(win)
        print(f"   ({y},{x}) got {got[y, x]!r} reference (sequential float32) {want[y, x]!r} exact {exact!r}  valid {np.isfinite(win).sum()}  over by {over[y, x]:.3g}")
