"""Parity report: every comparison a GPU test makes can be recorded as (config, op) -> max relative error on
|ref| > eps, max absolute error, cells compared.  With XRS_PARITY_REPORT=<path> in the environment the session writes the
table as JSON when it ends (tests/conftest.py); profiles/r02/parity.json is such a file from the GPU box."""
import json
import os

import numpy as np

_rows = {}
EPS = 1e-6          # |ref| <= EPS cells count for max_abs only (max-rel on |ref| > eps)


def record(config, op, got, want, tol=None, eps=EPS, note=None):
    """Accumulate the error of `got` against `want`; returns (max_rel, max_abs) of this call."""
    got = np.asarray(got, dtype=np.float64).ravel()
    want = np.asarray(want, dtype=np.float64).ravel()
    both_nan = np.isnan(got) & np.isnan(want)
    nan_mismatch = int(np.count_nonzero(np.isnan(got) != np.isnan(want)))
    fin = np.isfinite(got) & np.isfinite(want)
    inf_mismatch = int(np.count_nonzero(~fin & ~both_nan & ~(np.isnan(got) != np.isnan(want)) & (got != want)))
    d = np.abs(got[fin] - want[fin])
    big = np.abs(want[fin]) > eps
    max_abs = float(d.max()) if d.size else 0.0
    max_rel = float((d[big] / np.abs(want[fin][big])).max()) if big.any() else 0.0
    key = (config, op)
    r = _rows.setdefault(key, {"config": config, "op": op, "cells": 0, "cells_ref_above_eps": 0, "max_rel": 0.0,
                               "max_abs": 0.0, "nan_or_inf_mismatch": 0, "eps": eps})
    r["cells"] += int(got.size)
    r["cells_ref_above_eps"] += int(np.count_nonzero(big))
    r["max_rel"] = max(r["max_rel"], max_rel)
    r["max_abs"] = max(r["max_abs"], max_abs)
    r["nan_or_inf_mismatch"] += nan_mismatch + inf_mismatch
    if tol is not None:
        r["tolerance"] = tol
    if note:
        r["note"] = note
    return max_rel, max_abs


def flush(path=None):
    path = path or os.environ.get("XRS_PARITY_REPORT")
    if not path or not _rows:
        return None
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    rows = sorted(_rows.values(), key=lambda r: (r["config"], r["op"]))
    with open(path, "w") as fh:
        json.dump({"what": "max relative error on |ref| > eps and max absolute error of the HIP path against the CPU oracle, "
                           "per op and BASELINE config, collected by the -m gpu tests", "rows": rows}, fh, indent=1)
    return path


RTOL = 1e-5


def assert_hillshade(got, want, err_msg=""):
    """north_star's bar for hillshade: 1e-5 RELATIVE on every cell whose reference value is not (numerically) zero -- no absolute
    term; only cells with |reference| <= 1e-6 (a fully shadowed facet: (shaded + 1) / 2 of float32 terms that cancel) are
    held to 1e-6 absolute instead."""
    got, want = np.asarray(got), np.asarray(want)
    tiny = np.abs(want) <= 1e-6
    np.testing.assert_allclose(np.where(tiny, want, got), want, rtol=RTOL, atol=0.0, equal_nan=True, err_msg=err_msg)
    np.testing.assert_allclose(np.where(tiny, got, 0.0), np.where(tiny, want, 0.0), rtol=0.0, atol=1e-6, equal_nan=True, err_msg=err_msg + " (cells at zero)")
