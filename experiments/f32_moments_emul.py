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


def emulate_carry(z, nanmask, fill, R=12, NC=2, period=5):
    """Round 5: the CARRYING walk of mom_impl.h (MomWalk<.., CARRY>), float32 operation by operation.  A nodata cell is
    overwritten with `fill` before anybody reads the row and is an ordinary cell from then on (prefix sums, ring,
    re-centring by the mean of the round's widest runs with COMPILE-TIME counts, one-term history of the re-centrings);
    an output row takes its L lost cells out again about the lane's current shift: S -= L (fill - c), Q -= L (fill - c)^2,
    n = ntaps - L, and is guarded on the Q that was summed.  Returns mean, var and the guard's verdict per cell."""
    H, W = z.shape
    K = 2 * R + 1
    HL = NC * ((R + NC - 1) // NC)
    NV = NC + 2 * HL
    hws = [hw_circle(R, abs(dy)) for dy in range(-R, R + 1)]
    ntaps = sum(2 * h + 1 for h in hws)
    levels = sorted(set(hws))
    nl = (W - 2 * HL) // NC
    x0 = HL + NC * np.arange(nl)
    zf = np.where(nanmask, f32(fill), z).astype(f32)
    lostmap = nanmask.astype(np.int64)
    accS = np.zeros((K, nl, NC), f32)
    accQ = np.zeros((K, nl, NC), f32)
    nacc = np.zeros(K, np.int64)
    mean = np.full((H, W), np.nan, f32)
    var = np.full((H, W), np.nan, f32)
    good = np.zeros((H, W), bool)
    c = (f32(0.25) * (zf[0, x0] + zf[0, x0 + NC - 1] + zf[1, x0] + zf[1, x0 + NC - 1])).astype(f32)
    snap_sum = np.zeros(nl, f32)
    dqn = np.zeros(nl, f32)
    idx_cols = x0[:, None] - HL + np.arange(NV)[None, :]
    for t in range(H):
        if t % period == 0 and t > 0:
            cn = (c + (snap_sum * f32(1.0 / ((2 * R + 1) * period))).astype(f32)).astype(f32)
            snap_sum = np.zeros(nl, f32)
            delta = (cn - c).astype(f32)
            for j in range(K):
                if nacc[j] == 0:
                    continue
                N = f32(nacc[j])
                for o in range(NC):
                    S = accS[j, :, o]
                    S2 = (S - (N * delta).astype(f32)).astype(f32)
                    accQ[j, :, o] = (accQ[j, :, o] - (delta * (S + S2).astype(f32)).astype(f32)).astype(f32)
                    accS[j, :, o] = S2
            c = cn
            dqn = np.maximum((f32(ntaps) * (delta * delta).astype(f32)).astype(f32), (dqn * f32(0.85)).astype(f32))
        w = (zf[t][idx_cols] - c[:, None]).astype(f32)
        w2 = (w * w).astype(f32)
        P = np.empty_like(w)
        PQ = np.empty_like(w)
        P[:, 0], PQ[:, 0] = w[:, 0], w2[:, 0]
        for k in range(1, NV):
            P[:, k] = (P[:, k - 1] + w[:, k]).astype(f32)
            PQ[:, k] = (PQ[:, k - 1] + w2[:, k]).astype(f32)
        lev_S, lev_Q = {}, {}
        for h in levels:
            s_ = np.empty((nl, NC), f32)
            q_ = np.empty((nl, NC), f32)
            for o in range(NC):
                hi, lo = HL + o + h, HL + o - h - 1
                if h == 0:
                    s_[:, o], q_[:, o] = w[:, HL + o], w2[:, HL + o]
                elif lo >= 0:
                    s_[:, o], q_[:, o] = (P[:, hi] - P[:, lo]).astype(f32), (PQ[:, hi] - PQ[:, lo]).astype(f32)
                else:
                    s_[:, o], q_[:, o] = P[:, hi], PQ[:, hi]
            lev_S[h], lev_Q[h] = s_, q_
        snap_sum = (snap_sum + lev_S[R][:, 0]).astype(f32)
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
            accQ[j] = (accQ[j] + lev_Q[h]).astype(f32)
            nacc[j] += 2 * h + 1
        yo = t - R
        if yo >= R:
            j = yo % K
            L = np.zeros((nl, NC), f32)                       # the lost ring's entry: NaN cells under each window (exact)
            for dy in range(-R, R + 1):
                h = hws[dy + R]
                for o in range(NC):
                    cs = np.cumsum(np.concatenate([[0], lostmap[yo + dy]]))
                    L[:, o] += cs[x0 + o + h + 1] - cs[x0 + o - h]
            dl = (f32(fill) - c).astype(f32)[:, None]
            dl2 = (dl * dl).astype(f32)
            n = (f32(ntaps) - L).astype(f32)
            rn = (f32(1.0) / n).astype(f32)
            Qa = accQ[j]
            S = (accS[j] - (L * dl).astype(f32)).astype(f32)
            Q = (Qa - (L * dl2).astype(f32)).astype(f32)
            ms = (S * rn).astype(f32)
            m = (c[:, None] + ms).astype(f32)
            e = (Q - (S * ms).astype(f32)).astype(f32)
            B = (Qa + dqn[:, None]).astype(f32)
            ok = (e >= f32(0.2) * B) & ((m * m * n) >= f32(0.04) * B) & (L <= f32(0.5 * ntaps))
            v0 = (e * rn).astype(f32)
            for o in range(NC):
                mean[yo, x0 + o] = m[:, o]
                var[yo, x0 + o] = v0[:, o]
                good[yo, x0 + o] = ok[:, o]
    return mean, var, good


def reference_nan(z, nanmask, R=12):
    """float64 two-pass mean / variance of the valid cells under every window."""
    H, W = z.shape
    zz = np.where(nanmask, 0.0, z.astype(np.float64))
    vv = (~nanmask).astype(np.float64)
    offs = [(dy, dx) for dy in range(-R, R + 1) for dx in range(-hw_circle(R, abs(dy)), hw_circle(R, abs(dy)) + 1)]
    core = (slice(R, H - R), slice(R, W - R))
    s = np.zeros((H - 2 * R, W - 2 * R))
    n = np.zeros_like(s)
    for dy, dx in offs:
        s += zz[R + dy:H - R + dy, R + dx:W - R + dx]
        n += vv[R + dy:H - R + dy, R + dx:W - R + dx]
    m = s / n
    q = np.zeros_like(s)
    for dy, dx in offs:
        q += vv[R + dy:H - R + dy, R + dx:W - R + dx] * (zz[R + dy:H - R + dy, R + dx:W - R + dx] - m) ** 2
    mean = np.full((H, W), np.nan)
    var = np.full((H, W), np.nan)
    mean[core] = m
    var[core] = q / n
    return mean, var


def report_carry(name, z, frac, seed=3, R=12, tile_rows=124, tile_cols=128):
    """The kernel's own granularity: wave tiles of 128 columns x 124 output rows, each walked with ITS fill value (the cell
    at the tile centre), and a tile's results are kept only if EVERY window of it passes the guard (one failure hands the
    whole tile to the NaN-aware walker).  Prints how many tiles are kept and the worst error on kept tiles."""
    nanmask = np.random.default_rng(seed).random(z.shape) < frac
    H, W = z.shape
    kept = total = 0
    worst_m = worst_v = 0.0
    worst_dropped = 0.0
    for y0 in range(0, H - 2 * R - tile_rows + 1, tile_rows):
        for x0 in range(0, W - 2 * R - tile_cols + 1, tile_cols):
            sub = z[y0:y0 + tile_rows + 2 * R, x0:x0 + tile_cols + 2 * R]
            msk = nanmask[y0:y0 + tile_rows + 2 * R, x0:x0 + tile_cols + 2 * R]
            cy, cx = sub.shape[0] // 2, sub.shape[1] // 2
            cand = sub[cy, cx::-1][~msk[cy, cx::-1]]                   # the centre cell, or the first finite one to its left
            fill = float(cand[0]) if cand.size else 0.0
            m, v, good = emulate_carry(sub, msk, fill, R=R)
            mr, vr = reference_nan(sub, msk, R=R)
            seen = np.isfinite(m) & np.isfinite(mr) & (vr > 0)
            total += 1
            ev = np.abs(v[seen] - vr[seen]) / np.abs(vr[seen])
            em = np.abs(m[seen] - mr[seen]) / np.abs(mr[seen])
            if good[seen].all():
                kept += 1
                worst_m, worst_v = max(worst_m, em.max()), max(worst_v, ev.max())
            else:
                worst_dropped = max(worst_dropped, ev[good[seen]].max() if good[seen].any() else 0.0)
    print(f"{name:22s} NaN {frac:.1%}: {kept} of {total} tiles kept; on kept tiles mean max-rel {worst_m:.2e}  var max-rel {worst_v:.2e}"
          f"   (worst var error among the PASSING windows of dropped tiles: {worst_dropped:.2e})", flush=True)


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
    ap.add_argument("--carry", action="store_true", help="round 5: the carrying walk on rasters with nodata (profiles/r05/carry_emul.log)")
    args = ap.parse_args()
    shape = (args.rows, args.cols)
    dems = {
        "smooth_dem": synth.smooth_dem(shape, seed=12),
        "asv_dem": synth.asv_dem(*shape),
        "plane+noise(0.01)": (1500 + 3.0 * np.arange(shape[1])[None, :] + 7.0 * np.arange(shape[0])[:, None]
                             + np.random.default_rng(1).normal(0, 0.01, shape)).astype(np.float32),
        "bands(500+-100)": synth.bands(shape, 5),
    }
    if args.carry:
        shape = (2 * 124 + 24, 4 * 128 + 24)
        rng = np.random.default_rng(2)
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        carry_dems = {
            "asv_dem": synth.asv_dem(*shape),
            "smooth_dem (steep)": synth.smooth_dem(shape, seed=12),
            "plane 3/7 per cell": (1500 + 3.0 * xx + 7.0 * yy + rng.normal(0, 0.5, shape)).astype(np.float32),
            "cliffs 5000 / -9000": (1000 + rng.normal(0, 2.0, shape) + 5000.0 * (xx > 200) - 9000.0 * (xx > 390)).astype(np.float32),
            "flat next to relief": (1000 + rng.normal(0, 0.05, shape) + (xx > 300) * 300.0 * np.sin(xx / 7.0)).astype(np.float32),
            "spikes": (1000 + rng.normal(0, 1.0, shape) + 4000.0 * (rng.random(shape) < 2e-3)).astype(np.float32),
        }
        for name, z in carry_dems.items():
            for frac in (0.001, 0.01):
                report_carry(name, z, frac)
        sys.exit(0)
    for name, z in dems.items():
        for kw in (dict(period=10), dict(period=10, shift_mode="rowmean"), dict(period=10, shift_mode="rowmean", lead=3),
                   dict(period=10, shift_mode="rowmean", qring64=True), dict(period=10, shift_mode="rowmean", NC=1),
                   dict(period=5, shift_mode="rowmean"), dict(period=25, shift_mode="rowmean")):
            report(name, z, **kw)
