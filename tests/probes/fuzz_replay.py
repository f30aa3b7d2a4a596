"""Replay one case of tests/fuzz_parity.py (any mode) and print every comparison it makes, not only the first failure.
    python tests/probes/fuzz_replay.py --structured 121 4"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import fuzz_parity as fz  # noqa: E402

args = sys.argv[1:]
for flag, name in (("--windows", "WINDOWS"), ("--structured", "STRUCTURED"), ("--big", "BIG")):
    if flag in args:
        setattr(fz, name, True)
        args.remove(flag)
seed, case = int(args[0]), int(args[1])
rng = np.random.default_rng(seed)
subs = [int(rng.integers(0, 2 ** 62)) for _ in range(case + 1)]
plain_close = fz.close


def loud_close(got, want, rtol=fz.RTOL, atol=0.0):
    g, w = fz.host(got), np.asarray(want)
    err = plain_close(got, want, rtol=rtol, atol=atol)
    if g.shape == w.shape and g.dtype.kind == "f":
        with np.errstate(all="ignore"):
            fin = np.isfinite(g) & np.isfinite(w)
            rel = np.where(fin, np.abs(g.astype(np.float64) - w) / np.maximum(np.abs(w), 1e-300), 0.0)
        bad = rel > rtol
        print(f"   compare {g.shape} {g.dtype} rtol {rtol:g} atol {np.max(atol):g}: {int(bad.sum())} beyond rtol, max rel {rel.max():.3g}" +
              (f"; worst got {g.flat[rel.argmax()]!r} want {w.flat[rel.argmax()]!r}" if bad.any() else ""))
        if bad.any() and g.ndim == 1:
            for i in np.nonzero(bad)[0][:12]:
                print(f"      [{i}] got {g[i]!r} want {w[i]!r} rel {rel[i]:.3g}")
    return err


fz.close = loud_close
desc, err = fz.one_case(np.random.default_rng(subs[case]), 10 ** 9 if (fz.WINDOWS or fz.STRUCTURED) else 400000)
print(desc, "->", err)
