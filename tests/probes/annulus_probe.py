"""Annuli of every inner radius on a raster with interior tiles: which (outer, inner) radii differ from the oracle, and where.
    python tests/probes/annulus_probe.py [R ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import xrspatial_amd as xs  # noqa: E402
from oracle import c_oracle as corc  # noqa: E402
from tests import synth  # noqa: E402
from xrspatial_amd import focal  # noqa: E402
from xrspatial_amd.convolution import annulus_kernel  # noqa: E402

radii = [int(r) for r in sys.argv[1:]] or [10, 11, 12]
rows, cols = 393, 900
smooth = synth.asv_dem(rows, cols).astype(np.float32)
rng = np.random.default_rng(5)
noisy = (1000.0 + 300.0 * rng.standard_normal((rows, cols))).astype(np.float32)
cliff = smooth.copy(); cliff[:, 450:] += 3000.0
for name, z in (("smooth", smooth), ("noisy", noisy), ("cliff", cliff)):
    A = xs.DataArray(xs.DeviceArray.from_numpy(z), dims=["y", "x"], attrs={"res": (1.0, 1.0)})
    for R in radii:
        for ri in range(1, R):
            k = annulus_kernel(1, 1, R, ri)
            got = focal.focal_stats(A, k, stats_funcs=["mean", "var"]).data.get()
            line = f"{name:6s} R={R} ri={ri} taps={int(k.sum())}:"
            for i, stat in enumerate(("mean", "var")):
                want = corc.focal_apply(z, k, stat, nthreads=8).astype(np.float64)
                g = got[i].astype(np.float64)
                rel = np.abs(g - want) / np.maximum(np.abs(want), 1e-30)
                bad = rel > 5e-6
                if bad.any():
                    ys, xs_ = np.nonzero(bad)
                    line += f"  {stat} {int(bad.sum())} bad (max {rel.max():.2g}) rows {ys.min()}..{ys.max()} cols {xs_.min()}..{xs_.max()}" \
                            f" xmod128 {np.bincount(xs_ % 128, minlength=128).argmax()} x&1 {np.bincount(xs_ & 1)}"
                else:
                    line += f"  {stat} ok ({rel.max():.1e})"
            print(line, flush=True)
