"""Debug probe: how good are the NaN-aware float32 walker's results on the tiles it hands to the exact walker?  Run with a
library built with EXTRA=-DXRS_MOM_NO_FALLBACK (XRS_LIB=...): results of tiles that failed their guard stay as the float32
walker wrote them.  Raster: white noise about 1000 (tools/nan_probe.py) or the steep DEM, left third NaN with a ragged edge."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs  # noqa: E402
from oracle import c_oracle as corc  # noqa: E402
from tests import synth  # noqa: E402
from xrspatial_amd.convolution import circle_kernel  # noqa: E402
from xrspatial_amd.focal import focal_stats  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
ragged = len(sys.argv) > 2 and sys.argv[2] == "ragged"
rows, cols = 131 * 6, 128 * 12
rng = np.random.default_rng(0)
z = (1000 + rng.random((rows, cols), dtype=np.float32) * 50) if kind == "noise" else synth.smooth_dem((rows, cols))
edge = cols // 3 + ((np.arange(rows) // 7) % 5 if ragged else 0)
z = z.copy()
z[np.arange(cols)[None, :] < np.broadcast_to(edge, (rows,))[:, None]] = np.nan
k = circle_kernel(1, 1, 12)
got = focal_stats(xs.DataArray(z, dims=['y', 'x']), k, stats_funcs=['mean', 'var', 'std']).data
with np.errstate(all='ignore'):
    want = {s: corc.focal_apply(z, k, s, nthreads=8) for s in ('mean', 'var', 'std')}
for i, s in enumerate(('mean', 'var', 'std')):
    g, w = np.asarray(got[i], np.float64), np.asarray(want[s], np.float64)
    with np.errstate(all='ignore'):
        rel = np.abs(g - w) / np.abs(w)
    rel[~np.isfinite(rel)] = 0
    mism = np.isnan(g) != np.isnan(w)
    print(s, "NaN mismatches", int(mism.sum()), "max rel", rel.max(), "cells > 1e-5:", int((rel > 1e-5).sum()), "> 2e-6:", int((rel > 2e-6).sum()))
    if s == 'var':
        bad = np.argwhere(rel > 2e-6)[:10]
        for y, x in bad:
            n = int(np.isfinite(z[max(0, y - 12):y + 13, max(0, x - 12):x + 13][k[max(0, 12 - y):, max(0, 12 - x):][:min(rows, y + 13) - max(0, y - 12), :min(cols, x + 13) - max(0, x - 12)] == 1]).sum())
            print("   ", y, x, "n =", n, "got", g[y, x], "want", w[y, x], "rel", rel[y, x])
