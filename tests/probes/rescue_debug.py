"""Where do the moments differ from the oracle on the 'nodata' case of test_third_generation_walkers (ragged region + scattered NaN)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import xrspatial_amd as xs
from oracle import c_oracle as corc
from tests import synth
from xrspatial_amd.convolution import circle_kernel
from xrspatial_amd.focal import focal_stats
rng = np.random.default_rng(11)
for radius, shape in ((12, (560, 1330)), (7, (450, 900))):
    k = circle_kernel(1, 1, radius)
    for _ in range(1):
        nodata = synth.asv_dem(*shape).copy()
        edge = shape[1] // 3 + (np.arange(shape[0]) // 7) % 5
        nodata[np.arange(shape[1])[None, :] < edge[:, None]] = np.nan
        nodata[rng.random(shape) < 0.002] = np.nan
    for env in ("1", "0"):
        os.environ["XRS_MOM_RESCUE"] = env
        got = np.asarray(focal_stats(xs.DataArray(nodata, dims=['y', 'x']), k, stats_funcs=['mean', 'var']).data)
        for i, st in enumerate(('mean', 'var')):
            want = corc.focal_apply(nodata, k, st, nthreads=8)
            with np.errstate(all='ignore'):
                bad = ~((np.abs(got[i] - want) <= 5e-6 * np.abs(want)) | (np.isnan(got[i]) & np.isnan(want)))
            ys, xs_ = np.nonzero(bad)
            print(f"r={radius} rescue={env} {st}: {bad.sum()} bad", end="")
            if bad.any():
                print(f"  rows {ys.min()}..{ys.max()} cols {xs_.min()}..{xs_.max()}; distinct cols {len(np.unique(xs_))}, distinct rows {len(np.unique(ys))}; "
                      f"sample got {got[i][ys[0], xs_[0]]} want {want[ys[0], xs_[0]]} at {(ys[0], xs_[0])}")
                hist = np.bincount(xs_ // 64)
                print("    bad cells per 64-column block:", {int(b): int(c) for b, c in enumerate(hist) if c})
                hist = np.bincount(ys // 8)
                print("    bad cells per 8-row block:", {int(b) * 8: int(c) for b, c in enumerate(hist) if c})
            else:
                print()
    if os.environ.get("RESCUE_MARK"):
        os.environ["XRS_MOM_RESCUE"] = "1"
        got = np.asarray(focal_stats(xs.DataArray(nodata, dims=['y', 'x']), k, stats_funcs=['mean', 'var']).data)
        want = corc.focal_apply(nodata, k, 'mean', nthreads=8)
        marked = got[0] == -12345.0
        with np.errstate(all='ignore'):
            bad = ~((np.abs(got[0] - want) <= 5e-6 * np.abs(want)) | (np.isnan(got[0]) & np.isnan(want))) & ~marked
        print(f"r={radius}: marked (flagged by the guard) {marked.sum()}, wrong and NOT marked {bad.sum()}")
        ys, xs_ = np.nonzero(bad)
        for y, x in list(zip(ys, xs_))[:8]:
            win = nodata[max(0, y - radius):y + radius + 1, max(0, x - radius):x + radius + 1]
            print(f"    ({y},{x}) got {got[0][y, x]} want {want[y, x]}  valid cells in the bounding box {np.isfinite(win).sum()}")
        ys, xs_ = np.nonzero(marked)
        if len(ys):
            print("    marked cols", xs_.min(), xs_.max(), "rows", ys.min(), ys.max())
