"""Debug probe (round 3): where does the float32 moments walker hand tiles to the exact walker?  Runs the 25x25 circle
mean / var / std on an asv DEM with the no-fallback build (a library built with EXTRA=-DXRS_MOM_NO_FALLBACK, loaded through XRS_LIB) and reports, per wave tile
(128 columns x 131 rows), whether the fast path wrote it completely and how far it is from a float64 reference."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs  # noqa: E402
from oracle import c_oracle as corc  # noqa: E402
from tests import synth  # noqa: E402
from xrspatial_amd.convolution import circle_kernel  # noqa: E402
from xrspatial_amd.focal import focal_stats  # noqa: E402

rows, cols = 131 * 8, 128 * 12
z = synth.asv_dem(rows, cols) if len(sys.argv) < 2 or sys.argv[1] == "asv" else synth.smooth_dem((rows, cols))
k = circle_kernel(1, 1, 12)
sentinel = np.float32(-12345.0)
got = focal_stats(xs.DataArray(z, dims=['y', 'x']), k, stats_funcs=['mean', 'var', 'std']).data
want = {s: corc.focal_apply(z, k, s, nthreads=8) for s in ('mean', 'var', 'std')}
for ty in range(rows // 131):
    line = []
    for tx in range(cols // 128):
        sl = (slice(ty * 131, ty * 131 + 131), slice(tx * 128, tx * 128 + 128))
        g, w = np.asarray(got[1][sl], np.float64), np.asarray(want['var'][sl], np.float64)
        ok = np.isfinite(g) & np.isfinite(w)
        rel = np.abs(g[ok] - w[ok]) / np.abs(w[ok])
        line.append("%8.1e" % rel.max())
    print(ty, " ".join(line))
