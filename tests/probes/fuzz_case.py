"""Re-run single cases of tests/fuzz_parity.py --windows and say where and by how much the statistics differ.
    python tests/probes/fuzz_case.py SEED CASE [CASE ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import xrspatial_amd as xs  # noqa: E402
from oracle import c_oracle as corc  # noqa: E402
from oracle import xrs_oracle as orc  # noqa: E402
from tests import fuzz_parity as fz  # noqa: E402
from xrspatial_amd import focal  # noqa: E402

fz.WINDOWS = True
seed, cases = int(sys.argv[1]), [int(c) for c in sys.argv[2:]]
rng = np.random.default_rng(seed)
subs = [int(rng.integers(0, 2 ** 62)) for _ in range(max(cases) + 1)]
for ci in cases:
    sub = np.random.default_rng(subs[ci])
    shape = fz.pick_shape(sub, 1e9)
    backend = str(sub.choice(["numpy", "hip"]))
    str(sub.choice(["slope", "aspect", "curvature", "hillshade", "mean", "apply", "focal_stats", "convolve", "ndvi", "evi",
                    "zonal", "crosstab", "hotspots", "fuse", "trim", "true_color"]))
    np.dtype(sub.choice([np.float32, np.float32, np.float64, np.int16, np.uint8, np.int32]))
    op, dtype = str(sub.choice(["apply", "focal_stats", "focal_stats"])), np.dtype(sub.choice([np.float32, np.float32, np.float64]))
    z = fz.make_raster(sub, shape, dtype)
    k = fz.random_kernel(sub)
    z32 = z.astype(np.float32)
    print(f"== seed {seed} case {ci}: {op} {shape} {dtype} {backend} k={k.shape} taps={int(k.sum())}  nan {np.isnan(z).mean():.3f} inf {int(np.isinf(z).sum())} "
          f"range {np.nanmin(z32[np.isfinite(z32)]) if np.isfinite(z32).any() else None} .. {np.nanmax(z32[np.isfinite(z32)]) if np.isfinite(z32).any() else None}")
    for env in ("1", "0"):
        os.environ["XRS_MOM_RESCUE"] = env
        if op == "apply":
            got = [fz.host(focal.apply(fz.agg_of(z, backend), k, getattr(focal, "_calc_" + st)).data) for st in orc.FOCAL_STATS]
        else:
            got = fz.host(focal.focal_stats(fz.agg_of(z, backend), k).data)
        for i, stat in enumerate(orc.FOCAL_STATS):
            with np.errstate(all="ignore"):
                want = corc.focal_apply(z, k, stat, nthreads=8)
                g = got[i].astype(np.float64); w = want.astype(np.float64)
                nanm = np.isnan(g) != np.isnan(w)
                fin = np.isfinite(g) & np.isfinite(w)
                rel = np.where(fin, np.abs(g - w) / np.maximum(np.abs(w), 1e-300), 0.0)
                absd = np.where(fin, np.abs(g - w), 0.0)
                infm = (~fin & ~np.isnan(w) & ~np.isnan(g)) & (g != w)
            tol = 1e-6 if stat in ("max", "min", "range") else 5e-6
            bad = (rel > tol) & (absd > 1e-30)
            if nanm.any() or infm.any() or bad.any():
                ys, xs_ = np.nonzero(bad | nanm | infm)
                j = int(np.argmax(rel))
                y, x = np.unravel_index(j, rel.shape)
                n_valid = int(np.isfinite(z32[max(0, y - k.shape[0] // 2):y + k.shape[0] // 2 + 1, max(0, x - k.shape[1] // 2):x + k.shape[1] // 2 + 1]).sum())
                print(f"   rescue={env} {stat:6s}: {int(bad.sum())} cells beyond {tol:g} (max rel {rel.max():.3g} at ({y},{x}): got {g[y, x]!r} want {w[y, x]!r}, "
                      f"{n_valid} valid cells in its bounding box), NaN mismatches {int(nanm.sum())}, inf mismatches {int(infm.sum())}; rows {ys.min()}..{ys.max()} cols {xs_.min()}..{xs_.max()}")
