"""Experiment (not part of the library): numpy emulation, operation by operation in float32, of the float32 moment walker
planned for round 3 (walk3_impl.h) -- lane-local prefix sums of w = v - c and w^2 over the 2 + 2R cells under a lane's
two windows, one subtraction per distinct half-width, float32 ring accumulation down the rows, the lane's shift c
trailing the walk (own-column value ~R rows behind, replaced every `period` rows with an exact-algebra re-centring of
the partial sums).  Prints the error of mean / var / std against a float64 two-pass reference, so that the guard
constants can be chosen from data before any HIP is written.

    python experiments/f32_moments_emul.py [--period 10] [--qring64]
"""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import synth  # noqa: E402

f32 = np.float32


def hw_circle(R, dy):
    h = 0
    while (h + 1) ** 2 + dy * dy <= R * R:
        h += 1
    return h


def emulate(z, R=12, NC=2, period=10, lead=None, qring64=False, shift_mode="trail"):
    """Returns mean, var (float32 arrays) for the interior rows/cols of z, computed the way the kernel would."""
    H, W = z.shape
    K = 2 * R + 1
    HL = NC * ((R + NC - 1) // NC)
    NV = NC + 2 * HL
    hws = [hw_circle(R, abs(dy)) for dy in range(-R, R + 1)]
    ntaps = sum(2 * h + 1 for h in hws)
    levels = sorted(set(hws))
    # lanes: x0 = HL + NC * l, needs x0 - HL >= 0 and x0 + NC - 1 + HL < W
    nl = (W - 2 * HL) // NC
    x0 = HL + NC * np.arange(nl)
    qdt = np.float64 if qring64 else f32
    accS = np.zeros((K, nl, NC), f32)
    accQ = np.zeros((K, nl, NC), qdt)
    nacc = np.zeros(K, np.int64)          # cells accumulated per slot (same for every lane: interior)
    mean = np.full((H, W), np.nan, f32)
    var = np.full((H, W), np.nan, f32)
    if lead is None:
        lead = period // 2
    c = z[0, x0].astype(f32).copy()
    snap = np.zeros(nl, f32)
    snap_ms = np.zeros(nl, f32)
    snap_sum = np.zeros(nl, f32)
    idx_cols = x0[:, None] - HL + np.arange(NV)[None, :]
    for t in range(H):
        if t % period == 0 and t > 0:
            # re-centre: new shift = own-column value at row t - R + lead (already walked)
            if shift_mode == "trail":
                rr = max(t - R + lead, 0)
                cn = z[rr, x0].astype(f32)
            elif shift_mode == "rowmean":
                # the widest centred run of row t - R + lead, as the kernel has it (a float32 sum about the old shift)
                cn = (c + (snap * f32(1.0 / (2 * R + 1))).astype(f32)).astype(f32)
            elif shift_mode == "roundmean":
                # mean of the widest runs of all `period` rows of the round just walked
                cn = (c + (snap_sum * f32(1.0 / ((2 * R + 1) * period))).astype(f32)).astype(f32)
                snap_sum = np.zeros(nl, f32)
            elif shift_mode == "avg":
                # mean of that row's widest run and of the window that completed with it (centred R rows higher)
                cn = (c + (f32(0.5) * ((snap * f32(1.0 / (2 * R + 1))).astype(f32) + snap_ms)).astype(f32)).astype(f32)
            else:
                cn = c
            delta = (cn - c).astype(f32)             # exact (Sterbenz) for close values
            for j in range(K):
                N = f32(nacc[j])
                if nacc[j] == 0:
                    continue
                for o in range(NC):
                    S = accS[j, :, o]
                    t1 = (N * delta).astype(f32)
                    S2 = (S - t1).astype(f32)
                    u = (S + S2).astype(f32)
                    if qring64:
                        accQ[j, :, o] = accQ[j, :, o] - delta.astype(np.float64) * u.astype(np.float64)
                    else:
                        accQ[j, :, o] = (accQ[j, :, o] - (delta * u).astype(f32)).astype(f32)
                    accS[j, :, o] = S2
            c = cn
        w = (z[t][idx_cols] - c[:, None]).astype(f32)            # (nl, NV)
        w2 = (w * w).astype(f32)
        P = np.empty_like(w)
        PQ = np.empty_like(w)
        P[:, 0] = w[:, 0]
        PQ[:, 0] = w2[:, 0]
        for k in range(1, NV):
            P[:, k] = (P[:, k - 1] + w[:, k]).astype(f32)
            PQ[:, k] = (PQ[:, k - 1] + w2[:, k]).astype(f32)
        lev_S, lev_Q = {}, {}
        snap_now = (t % period) == ((-R + lead) % period)
        for h in levels:
            s = np.empty((nl, NC), f32)
            q = np.empty((nl, NC), f32)
            for o in range(NC):
                hi, lo = HL + o + h, HL + o - h - 1
                if h == 0:
                    s[:, o] = w[:, HL + o]
                    q[:, o] = w2[:, HL + o]
                elif lo >= 0:
                    s[:, o] = (P[:, hi] - P[:, lo]).astype(f32)
                    q[:, o] = (PQ[:, hi] - PQ[:, lo]).astype(f32)
                else:
                    s[:, o] = P[:, hi]
                    q[:, o] = PQ[:, hi]
            lev_S[h], lev_Q[h] = s, q
        if snap_now:
            snap = lev_S[R][:, 0].copy()
        snap_sum = (snap_sum + lev_S[R][:, 0]).astype(f32)
        # ring: slot for output row yo is yo % K; this row contributes to yo = t - dy
        for dy in range(-R, R + 1):
            yo = t - dy
            if yo < 0 or yo >= H:
                continue
            j = yo % K
            h = hws[dy + R]
            if dy == -R:
                accS[j] = 0
                accQ[j] = 0
                nacc[j] = 0
            accS[j] = (accS[j] + lev_S[h]).astype(f32)
            if qring64:
                accQ[j] = accQ[j] + lev_Q[h].astype(np.float64)
            else:
                accQ[j] = (accQ[j] + lev_Q[h]).astype(f32)
            nacc[j] += 2 * h + 1
        yo = t - R
        if yo >= R:
            j = yo % K
            assert nacc[j] == ntaps
            S = accS[j]
            Q = accQ[j]
            inv = f32(1.0 / ntaps)
            ms = (S * inv).astype(f32)
            m = (c[:, None] + ms).astype(f32)
            if snap_now:
                snap_ms = ms[:, 0].copy()
            if qring64:
                v0 = ((Q - S.astype(np.float64) * ms.astype(np.float64)) / ntaps).astype(f32)
            else:
                v0 = (((Q - (S * ms).astype(f32)).astype(f32)) * inv).astype(f32)
            for o in range(NC):
                mean[yo, x0 + o] = m[:, o]
                var[yo, x0 + o] = v0[:, o]
    return mean, var


def reference(z, R=12):
    H, W = z.shape
    zz = z.astype(np.float64)
    mean = np.full((H, W), np.nan)
    var = np.full((H, W), np.nan)
    offs = [(dy, dx) for dy in range(-R, R + 1) for dx in range(-hw_circle(R, abs(dy)), hw_circle(R, abs(dy)) + 1)]
    n = len(offs)
    core = (slice(R, H - R), slice(R, W - R))
    s = np.zeros((H - 2 * R, W - 2 * R))
    for dy, dx in offs:
        s += zz[R + dy:H - R + dy, R + dx:W - R + dx]
    m = s / n
    q = np.zeros_like(s)
    for dy, dx in offs:
        q += (zz[R + dy:H - R + dy, R + dx:W - R + dx] - m) ** 2
    mean[core] = m
    var[core] = q / n
    return mean, var


def report(name, z, **kw):
    m, v = emulate(z, **kw)
    mr, vr = reference(z)
    ok = np.isfinite(m) & np.isfinite(mr)
    em = np.abs(m[ok] - mr[ok]) / np.abs(mr[ok])
    ev = np.abs(v[ok] - vr[ok]) / np.abs(vr[ok])
    es = np.abs(np.sqrt(np.maximum(v[ok], 0).astype(np.float64)) - np.sqrt(vr[ok])) / np.sqrt(vr[ok])
    print(f"{name:28s} {kw}: mean max-rel {em.max():.2e}  var max-rel {ev.max():.2e} (p99 {np.quantile(ev, 0.99):.2e}, "
          f"median {np.median(ev):.2e})  std max-rel {es.max():.2e}   [{ok.sum()} cells]", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=220)
    ap.add_argument("--cols", type=int, default=420)
    args = ap.parse_args()
    shape = (args.rows, args.cols)
    dems = {
        "smooth_dem": synth.smooth_dem(shape, seed=12),
        "asv_dem": synth.asv_dem(*shape),
        "plane+noise(0.01)": (1500 + 3.0 * np.arange(shape[1])[None, :] + 7.0 * np.arange(shape[0])[:, None]
                             + np.random.default_rng(1).normal(0, 0.01, shape)).astype(np.float32),
        "bands(500+-100)": synth.bands(shape, 5),
    }
    for name, z in dems.items():
        for kw in (dict(period=10), dict(period=10, shift_mode="rowmean"), dict(period=10, shift_mode="rowmean", lead=3),
                   dict(period=10, shift_mode="rowmean", qring64=True), dict(period=10, shift_mode="rowmean", NC=1),
                   dict(period=5, shift_mode="rowmean"), dict(period=25, shift_mode="rowmean")):
            report(name, z, **kw)
