"""Large-window focal kernels on the GPU box: parity against the C oracle and same-box A/B timing of the
first-generation column walkers (flag XRS_FOCAL_EXACT_MOMENTS) against the wide row walker / second-generation walker.

    python tests/focal_large_check.py [--out gpurun_out/focal_large.json] [--size 16384] [--skip-parity]

Never stops at the first failure: every case is reported (max relative / absolute error, mismatching cells).
"""
import argparse
import ctypes
import json
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import xrspatial_amd as xs  # noqa: E402
from oracle import c_oracle as corc  # noqa: E402
from tests import synth  # noqa: E402
from xrspatial_amd import _lib  # noqa: E402
from xrspatial_amd.convolution import circle_kernel  # noqa: E402
from xrspatial_amd.focal import focal_stats  # noqa: E402

STATS = ['mean', 'max', 'min', 'range', 'std', 'var', 'sum']


def err(got, want):
    """(max relative error where |want| > 1e-30, max absolute error, cells where exactly one side is NaN)"""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    nan_mismatch = int(np.count_nonzero(np.isnan(got) != np.isnan(want)))
    ok = np.isfinite(got) & np.isfinite(want)
    inf_mismatch = int(np.count_nonzero(~ok & ~np.isnan(got) & ~np.isnan(want) & (got != want)))
    d = np.abs(got[ok] - want[ok])
    big = np.abs(want[ok]) > 1e-30
    rel = float((d[big] / np.abs(want[ok][big])).max()) if big.any() else 0.0
    return rel, (float(d.max()) if d.size else 0.0), nan_mismatch + inf_mismatch


def parity(report):
    rng = np.random.default_rng(5)
    cases = []
    for kind in ("circle", "box"):
        for radius in (3, 4, 6, 9, 11, 12):
            K = 2 * radius + 1
            k = circle_kernel(1, 1, radius) if kind == "circle" else np.ones((K, K))
            for label, z in (
                ("asv 300x700", synth.asv_dem(300, 700)),
                ("smooth 263x1100", synth.smooth_dem((263, 1100))),
                ("smooth+nan 150x331", synth.smooth_dem((150, 331), nan_frac=0.02, seed=radius)),
                ("zero-mean 140x300", rng.normal(0, 3, (140, 300)).astype(np.float32)),
                ("narrow %dx70" % (K - 2), synth.smooth_dem((K - 2, 70), seed=3)),
            ):
                cases.append((kind, radius, k, label, z))
    for kind, radius, k, label, z in cases:
        name = f"{kind} r={radius} {label}"
        try:
            agg = xs.DataArray(z, dims=['y', 'x'])
            t0 = time.time()
            got_mean = focal_stats(agg, k, ['mean']).data[0]
            got_all = focal_stats(agg, k, STATS).data
            want = {s: corc.focal_apply(z, k, s, nthreads=8) for s in STATS}
            # the reference's float32 sequential sum carries up to (n-1) * 2^-24 * sum|v| of rounding error itself
            absz = np.abs(np.nan_to_num(z, nan=0.0, posinf=0.0, neginf=0.0))
            sum_bound = (k.sum() - 1) * 2.0 ** -24 * corc.focal_apply(absz, k, 'sum', nthreads=8).astype(np.float64)
            row = {"case": name, "seconds": None}
            rel, ab, bad = err(got_mean, want['mean'])
            row["mean_only"] = {"max_rel": rel, "max_abs": ab, "mismatch": bad}
            worst = rel if bad == 0 else float("inf")
            for i, s in enumerate(STATS):
                rel, ab, bad = err(got_all[i], want[s])
                ent = {"max_rel": rel, "max_abs": ab, "mismatch": bad}
                if s == 'sum':
                    with np.errstate(all='ignore'):
                        okc = np.isfinite(got_all[i]) & np.isfinite(want[s])
                        over = np.abs(got_all[i][okc].astype(np.float64) - want[s][okc]) - sum_bound[okc]
                    ent["max_excess_over_reference_rounding_bound"] = float(over.max()) if over.size else 0.0
                    ent["ok"] = bool(bad == 0 and (rel <= 1e-5 or ent["max_excess_over_reference_rounding_bound"] <= 0))
                elif s in ('max', 'min', 'range'):
                    ent["ok"] = bool(bad == 0 and ab == 0.0)
                else:
                    ent["ok"] = bool(bad == 0 and rel <= 1e-5)
                row[s] = ent
            row["mean_only"]["ok"] = bool(worst <= 1e-5)
            row["ok"] = all(v["ok"] for kx, v in row.items() if isinstance(v, dict))
            row["seconds"] = round(time.time() - t0, 2)
        except Exception:          # noqa: BLE001
            row = {"case": name, "ok": False, "exception": traceback.format_exc()[-1500:]}
        report["parity"].append(row)
        flag = "ok " if row.get("ok") else "BAD"
        detail = "" if row.get("ok") else json.dumps({kx: v for kx, v in row.items() if isinstance(v, dict) and not v.get("ok")})[:600]
        print(flag, name, detail, flush=True)
        if "exception" in row:
            print(row["exception"], flush=True)


def timing(report, n):
    L = _lib.call
    dem = xs.DeviceArray((n, n), np.float32)
    band = synth.asv_dem(2048, n, y0=0, total_rows=n)
    for y0 in range(0, n, 2048):
        L("xrs_memcpy_h2d", dem.ptr + y0 * n * 4, band.ctypes.data, band.nbytes, None)
    L("xrs_stream_sync", None)
    outs_dev = [xs.DeviceArray((n, n), np.float32) for _ in range(7)]
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    L("xrs_event_create", ctypes.byref(e0))
    L("xrs_event_create", ctypes.byref(e1))

    def run(k, mask, reps=4, flags=0):
        kk = np.ascontiguousarray(k, dtype=np.float64)
        ptrs = (ctypes.c_void_p * 7)()
        for i in range(7):
            if mask >> i & 1:
                ptrs[i] = outs_dev[i].ptr
        fn = lambda: L("xrs_focal_stats_f32_ex", dem.ptr, ptrs, mask, n, n, n, n, kk.ctypes.data, k.shape[0], k.shape[1],  # noqa: E731
                       None, 0, 0, flags, None)
        fn()
        L("xrs_stream_sync", None)
        L("xrs_event_record", e0, None)
        for _ in range(reps):
            fn()
        L("xrs_event_record", e1, None)
        L("xrs_event_sync", e1)
        ms = ctypes.c_float()
        L("xrs_event_elapsed_ms", e0, e1, ctypes.byref(ms))
        return ms.value / reps

    for _ in range(20):                                    # clocks
        L("xrs_copy_f32", dem.ptr, outs_dev[0].ptr, n * n, None)
    L("xrs_stream_sync", None)
    L("xrs_event_record", e0, None)
    for _ in range(10):
        L("xrs_copy_f32", dem.ptr, outs_dev[0].ptr, n * n, None)
    L("xrs_event_record", e1, None)
    L("xrs_event_sync", e1)
    ms = ctypes.c_float()
    L("xrs_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    copy_gbs = 8.0 * n * n / (ms.value / 10 * 1e-3) / 1e9
    report["copy_gbs"] = round(copy_gbs, 1)
    print("streaming copy:", round(copy_gbs, 1), "GB/s", flush=True)
    for kind in ("circle", "box"):
        for radius in (12, 6, 4):
            K = 2 * radius + 1
            k = circle_kernel(1, 1, radius) if kind == "circle" else np.ones((K, K))
            for what, mask, nbytes in (("mean", 1, 8), ("all7", 127, 32), ("mean+var+std", 1 | 16 | 32, 16), ("sum", 64, 8)):
                row = {"mask": f"{kind} r={radius}", "stats": what}
                for gen, flags in (("gen1", 1), ("gen2", 0)):       # (gen1: XRS_FOCAL_EXACT_MOMENTS, the float64 column walkers)
                    try:
                        t = run(k, mask, flags=flags)
                        row[gen + "_ms"] = round(t, 4)
                        row[gen + "_gbs"] = round(nbytes * n * n / (t * 1e-3) / 1e9, 1)
                        row[gen + "_frac_of_copy"] = round(nbytes * n * n / (t * 1e-3) / 1e9 / copy_gbs, 3)
                    except Exception as exc:      # noqa: BLE001
                        row[gen + "_error"] = repr(exc)[:300]
                report["timing"].append(row)
                print(json.dumps(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/focal_large.json")
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--skip-parity", action="store_true")
    ap.add_argument("--skip-timing", action="store_true")
    args = ap.parse_args()
    _lib.require_device()
    report = {"parity": [], "timing": []}
    if not args.skip_parity:
        parity(report)
    if not args.skip_timing:
        timing(report, args.size)
    report["all_parity_ok"] = all(r.get("ok") for r in report["parity"])
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(report, fh, indent=1)
    print("parity:", "ALL OK" if report["all_parity_ok"] else "FAILURES", " cases:", len(report["parity"]))


if __name__ == "__main__":
    main()
