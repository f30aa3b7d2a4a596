// Register-resident strip layout shared by the compile-time-shape window kernels (kxk.hip) and the fused
// raster pass (pass.hip): a wave owns 256 columns x RB rows, a lane owns 4 adjacent columns and keeps the
// (RB + KH - 1) x (4 + 2*(KW/2)) cells its windows cover in VGPRs.
// `Args` needs: in, rows, cols, ld_in, halo_top, halo_bot.
#pragma once
#include "xrs_common.h"

namespace xrs {

// Fast reciprocal of a small positive count in float64: v_rcp_f64 + one Newton step (error ~1e-16,
// invisible after the float32 rounding of the result).  0 -> NaN after the multiply, like 0/0.
__device__ __forceinline__ double rcp_count(int n) {
    const double c = (double)n;
    double r = __builtin_amdgcn_rcp(c);
    r = fma(fma(-c, r, 1.0), r, r);
    return n ? r : nan("");
}

// s / n from a reciprocal, with one residual correction: exact whenever the quotient is representable, so the
// mean of a flat window is the cell value itself and its variance exactly 0, as with the reference's true division.
__device__ __forceinline__ double div_refined(double s, double n, double inv) {
    const double q = s * inv;
    const double q2 = fma(fma(-q, n, s), inv, q);
    return isfinite(q2) ? q2 : q;          // (+-inf sums, empty windows: keep the inf / NaN)
}

typedef xrs_f4u F4U;                                                   // 16 bytes at dword alignment
struct __attribute__((packed, aligned(4))) F2U { float x, y; };

// results of a lane's 4 columns: one 16-byte store, or the first `n` cells for the last lane of a ragged row
__device__ __forceinline__ void store_cols(float *p, float x, float y, float z, float w, int n) {
    if (n >= 4) { store_f4u(p, x, y, z, w); return; }
    p[0] = x;
    if (n > 1) p[1] = y;
    if (n > 2) p[2] = z;
}

// Strip loader shared by the register-resident kernels: v[r][0..NV) = columns x0-RX .. x0+3+RX of input
// row y0 - RY + r (NaN outside the raster / the shard's halo rows).  INTERIOR: no predicates at all.
template <int KH, int KW, int RB, bool INTERIOR, typename Args>
__device__ __forceinline__ void load_strip(const Args &a, long x_tile, long y0, int lane,
                                           float (&v)[RB + KH - 1][4 + 2 * (KW / 2)]) {
    constexpr int RX = KW / 2, RY = KH / 2, NV = 4 + 2 * RX, NR = RB + KH - 1;
    const long x0 = x_tile + lane * 4;
    const unsigned loff = (unsigned)lane * 4u;
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    const float qnan = nan_f32();
    const bool has_l = INTERIOR || x0 >= 4;          // x0 is a multiple of 4 and RX <= 3
    const bool has_r = INTERIOR || x0 + 8 <= a.cols;
    // (a lane whose own 4 columns or a halo block hang over the row's end -- widths that are not multiples of 4 --
    //  loads cell by cell; every other lane uses the 16-byte forms, which only need dword alignment)
    const bool ragged = !INTERIOR && x0 + 4 > a.cols;
    const bool ragged_r = !INTERIOR && !ragged && x0 + 8 > a.cols && x0 + 4 < a.cols;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const long y = y0 - RY + r;
        const bool ok = INTERIOR || (y >= y_lo && y < y_hi);
        const float *p = (a.in + y * a.ld_in + x_tile) + loff;       // scalar row base + lane offset
        if (!INTERIOR) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[r][i] = qnan;
        }
        if (ok && (ragged || ragged_r)) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const long xc = x0 - RX + i;
                if (xc >= 0 && xc < a.cols) v[r][i] = p[i - RX];
            }
        } else if (ok) {
            if (!(RX == 2 && INTERIOR)) {
                const F4U c4 = *reinterpret_cast<const F4U *>(p);       // (default cache policy: the halo rows / columns
                                                                        //  are re-read by the neighbouring strips from L2)
                v[r][RX] = c4.x; v[r][RX + 1] = c4.y; v[r][RX + 2] = c4.z; v[r][RX + 3] = c4.w;
            }
            if (RX == 1) {
                if (has_l) v[r][0] = p[-1];
                if (has_r) v[r][NV - 1] = p[4];
            } else if (RX == 2 && INTERIOR) {
                // the 8 cells x0-2 .. x0+5 as two 16-byte loads at 8-byte alignment (global loads only need
                // dword alignment): one instruction fewer per row than float2 + float4 + float2
                const F4U lo = *reinterpret_cast<const F4U *>(p - 2), hi = *reinterpret_cast<const F4U *>(p + 2);
                v[r][0] = lo.x; v[r][1] = lo.y; v[r][2] = lo.z; v[r][3] = lo.w;
                v[r][4] = hi.x; v[r][5] = hi.y; v[r][6] = hi.z; v[r][7] = hi.w;
            } else if (RX == 2) {
                if (has_l) { const F2U l2 = *reinterpret_cast<const F2U *>(p - 2); v[r][0] = l2.x; v[r][1] = l2.y; }
                if (has_r) { const F2U r2 = *reinterpret_cast<const F2U *>(p + 4); v[r][NV - 2] = r2.x; v[r][NV - 1] = r2.y; }
            } else {                                           // RX == 3 (7-wide): aligned float4 each side, 3 used
                if (has_l) { const F4U l4 = *reinterpret_cast<const F4U *>(p - 4); v[r][0] = l4.y; v[r][1] = l4.z; v[r][2] = l4.w; }
                if (has_r) { const F4U r4 = *reinterpret_cast<const F4U *>(p + 4); v[r][NV - 3] = r4.x; v[r][NV - 2] = r4.y; v[r][NV - 1] = r4.z; }
            }
        }
    }
}

// ---- The NaN-ignoring focal mean of a register strip (focal.py:305-326 with _calc_mean, :226-228; numba's nanmean: a
// float64 sum of the window's non-NaN cells over their count) -- ONE body for clean strips, strips with nodata and strips
// on the raster's edge (cells outside the raster arrive as NaN from load_strip).
//
// Every loaded row is converted to float64 once and added into the output rows whose windows cover it.  With the mask a
// compile-time constant the row's distinct tap patterns are summed once per input row (RowPlan below) and each goes into the
// output rows that see it with one addition.  Nodata costs a clean row five instructions per lane, BEFORE its conversion: a
// chain of fused multiply-adds over the lane's cells of the row (a NaN in any of the three operands comes out: 4 instructions
// for 8 cells) is non-finite when the lane holds a NaN (or +-inf) there; a wave-wide vote on that (one v_cmp_class_f32, scalar
// from there on) sends only the ROWS that hold one through the repair: NaN cells become 0 in the registers, their positions go
// into a per-lane bit mask (8 bits per row).  The repair sits between the loads and the sums and touches nothing but that row
// and two mask registers.  (Voting on the float64 row sums themselves costs two instructions, but the sums then have to be
// formed again behind the vote: a second copy of them cost the fused hillshade + 5x5 mean 170 spilled registers, a two-trip
// loop around one copy 86.)  +-inf stays in the sums and flows into the result like in the reference (inf, or NaN for
// inf - inf).  At the end an output row none of whose KH input rows was repaired is sum / ntaps; the others
// take their count as ntaps - popcount(mask bits under the window's taps) -- again only for the repaired rows -- and a
// window that lost no tap still multiplies by the same 1 / ntaps, so a cell's value does not depend on what else its strip
// holds (round 4 ran the strip a second time through a per-tap counting body as soon as one sum came out non-finite: 1.7x
// on the fused hillshade + 5x5 mean and 2.3x on the 5x5 mean alone at 0.1 % nodata, profiles/r04/r04z_nan_probe.log).
//
// Compile-time plan for the window sums of a mask known at compile time: the mask rows in order of increasing tap count,
// and for each the already summed row it can be built from (the largest subset).  The circular 5x5 mask has the row
// patterns {2}, {1,2,3}, {0..4}: per input row and output column the three row sums cost 0 + 2 + 2 additions and each
// goes into the output rows that see it with ONE addition -- 9 float64 additions per cell instead of 13 taps.  (Sums of
// float32 cells are exact in float64 as long as the window's cells are within 2^29 of each other in magnitude, so the
// association does not show.)
struct RowPlan { int order[8]; int base[8]; };
template <unsigned CMASK, int KH, int KW>
constexpr RowPlan make_row_plan() {
    RowPlan p = {};
    auto bits = [](int ky) { return (CMASK >> (ky * KW)) & ((1u << KW) - 1u); };
    auto pop = [](unsigned b) { int n = 0; for (; b; b &= b - 1) ++n; return n; };
    int cnt = 0;
    for (int c = 0; c <= KW; ++c)
        for (int ky = 0; ky < KH; ++ky)
            if (pop(bits(ky)) == c) p.order[cnt++] = ky;
    for (int r = 0; r < KH; ++r) {
        const unsigned b = bits(p.order[r]);
        int best = -1, bp = 0;
        for (int r2 = 0; r2 < r; ++r2) {
            const unsigned c = bits(p.order[r2]);
            if ((c & ~b) == 0u && pop(c) > bp) { best = r2; bp = pop(c); }
        }
        p.base[r] = best;
    }
    return p;
}

// rs[r][o] = sum of d under mask row plan.order[r] for output column o
template <unsigned CMASK, int KH, int KW>
__device__ __forceinline__ void row_pattern_sums(const double (&d)[4 + 2 * (KW / 2)], double (&rs)[KH][4]) {
    constexpr RowPlan plan = make_row_plan<CMASK ? CMASK : 1u, KH, KW>();
#pragma unroll
    for (int r = 0; r < KH; ++r) {
        const int ky = plan.order[r];
        const unsigned bits = (CMASK >> (ky * KW)) & ((1u << KW) - 1u);
        const unsigned have = plan.base[r] >= 0 ? (CMASK >> (plan.order[plan.base[r] >= 0 ? plan.base[r] : 0] * KW)) & ((1u << KW) - 1u) : 0u;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            double t = 0.0;
            bool first = true;
            if (plan.base[r] >= 0 && have) { t = rs[plan.base[r] >= 0 ? plan.base[r] : 0][o]; first = false; }
#pragma unroll
            for (int kx = 0; kx < KW; ++kx)
                if ((bits & ~have) >> kx & 1u) {
                    t = first ? d[kx + o] : t + d[kx + o];
                    first = false;
                }
            rs[r][o] = t;
        }
    }
}

// Which loaded rows of a register strip hold a NaN (or +-inf) somewhere in the wave: bit r of the result, wave-uniform.
// Fused multiply-adds carry a NaN from any of their three operands: 4 instructions for a lane's 8 cells of a row, one
// v_cmp_class_f32 and scalar bookkeeping per row, no branch.  (Cells beyond 1.8e19 overflow the products and are reported
// too; whoever repairs such a row finds nothing to do.)  0 = every cell of the strip is finite.
template <int NR, int NV>
__device__ __forceinline__ unsigned strip_probe_rows(const float (&v)[NR][NV]) {
    unsigned rows_hit = 0u;
#pragma unroll
    for (int ir = 0; ir < NR; ++ir) {
        float probe = fmaf(v[ir][0], v[ir][1], v[ir][2]);
#pragma unroll
        for (int i = 3; i + 2 < NV; i += 3) probe = fmaf(v[ir][i], v[ir][i + 1], probe) + v[ir][i + 2];
        if (NV % 3 == 2) probe = fmaf(v[ir][NV - 2], v[ir][NV - 1], probe);
        if (NV % 3 == 1) probe += v[ir][NV - 1];
        rows_hit |= __any(!isfinite(probe)) ? 1u << ir : 0u;
    }
    return rows_hit;
}

// CMASK: the window's taps as a compile-time constant (bit ky * KW + kx), 0 = read them from mask_rows at run time.
// SHARED: the row-pattern sums (compile-time masks only).  ROWWISE: a scheduling barrier per input row (the focal mean
// alone: without other work in front, the scheduler hoists the conversions of ALL rows to the top -- 128 registers of
// float64 images -- and spills 100 of them).  rows_hit: strip_probe_rows(v).  emit(r, m): the 4 means of output row r (rows
// in order).  WAVE_TABLE: all 64 lanes of the wave are here (interior strips) and ntaps < 64.
template <int KH, int KW, int RB, unsigned CMASK, bool SHARED, bool ROWWISE, bool WAVE_TABLE, typename MaskT, typename Emit>
__device__ __forceinline__ void strip_focal_mean(float (&v)[RB + KH - 1][4 + 2 * (KW / 2)], const MaskT *mask_rows,
                                                 int ntaps, double inv_ntaps, const unsigned rows_hit, Emit &&emit) {
    constexpr int NV = 4 + 2 * (KW / 2), NR = RB + KH - 1;
    constexpr unsigned ROWBITS = (1u << KW) - 1u;
    constexpr bool PLAN = SHARED && CMASK != 0u && KH <= 8;
    static_assert(NV <= 8 && NR <= 32, "strip_focal_mean: 8 mask bits per row");
    double acc[RB][4];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[r][o] = 0.0;
    unsigned nanbits[(NR + 3) / 4];           // per lane: bit 8 * (ir & 3) + i of entry ir >> 2 = cell i of loaded row ir is NaN
#pragma unroll
    for (int i = 0; i < (NR + 3) / 4; ++i) nanbits[i] = 0u;
    // All rows were voted on first (strip_probe_rows: no branch), so a clean strip meets ONE branch here and one at the end, and
    // the sums below stay one basic block (a vote-and-branch per row in front of each row's sums cost the clean raster 3-5 % in
    // every instantiation: profiles/r05/ab_row_votes.log).
    if (rows_hit) {
#pragma unroll
        for (int ir = 0; ir < NR; ++ir) {
            if (!(rows_hit >> ir & 1u)) continue;
            // NaN cells of this row -> 0, their positions into nanbits
            unsigned m = 0u;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool isn = isnan(v[ir][i]);
                m |= isn ? 1u << (8 * (ir & 3) + i) : 0u;
                v[ir][i] = isn ? 0.0f : v[ir][i];
            }
            nanbits[ir >> 2] |= m;
        }
    }

#pragma unroll
    for (int ir = 0; ir < NR; ++ir) {
        if (ROWWISE) __builtin_amdgcn_sched_barrier(0);
        double d[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) d[i] = (double)v[ir][i];
        if (PLAN) {
            constexpr RowPlan plan = make_row_plan<CMASK ? CMASK : 1u, KH, KW>();
            double rs[KH][4];
            row_pattern_sums<CMASK, KH, KW>(d, rs);
#pragma unroll
            for (int r = 0; r < KH; ++r) {
                const int ky = plan.order[r];
                const int orow = ir - ky;
                if (orow < 0 || orow >= RB || ((CMASK >> (ky * KW)) & ROWBITS) == 0u) continue;
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[orow][o] += rs[r][o];
            }
        } else {
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                const int orow = ir - ky;
                if (orow < 0 || orow >= RB) continue;
                const unsigned bits = CMASK ? ((CMASK >> (ky * KW)) & ROWBITS) : (unsigned)mask_rows[ky];
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
                    if (bits >> kx & 1u) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) acc[orow][o] += d[kx + o];
                    }
            }
        }
    }

    if (rows_hit == 0u) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            float m[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) m[o] = (float)(acc[r][o] * inv_ntaps);
            emit(r, m);
        }
        return;
    }
    // 1 / count as a table ACROSS the wave (interior strips: all 64 lanes are there): lane j holds 1 / (ntaps - j), lane 0 the
    // very 1 / ntaps of the clean path, lane ntaps NaN (0 / 0, like the reference); a window fetches its entry with two
    // ds_bpermute instead of v_rcp_f64 + a Newton step + selects per output (12 issue slots; 0.1 % nodata: 5x5 mean 0.48 ->
    // 0.44 ms).  Edge strips (lanes beyond the raster's last column are gone) compute theirs.
    const int lane = (int)(threadIdx.x & 63u);
    const double my_inv = lane == 0 ? inv_ntaps : rcp_count(ntaps - lane);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const unsigned hit = (rows_hit >> r) & ((1u << KH) - 1u);          // wave-uniform
        float m[4];
        if (hit == 0u) {
#pragma unroll
            for (int o = 0; o < 4; ++o) m[o] = (float)(acc[r][o] * inv_ntaps);
        } else {
            int lost[4] = {0, 0, 0, 0};
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                if (!(hit >> ky & 1u)) continue;                            // (scalar branch)
                const unsigned bits = CMASK ? ((CMASK >> (ky * KW)) & ROWBITS) : (unsigned)mask_rows[ky];
                const unsigned rowm = nanbits[(r + ky) >> 2] >> (8 * ((r + ky) & 3));
#pragma unroll
                for (int o = 0; o < 4; ++o) lost[o] += __popc((rowm >> o) & bits);
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const double inv = WAVE_TABLE ? __shfl(my_inv, lost[o]) : (lost[o] ? rcp_count(ntaps - lost[o]) : inv_ntaps);
                m[o] = (float)(acc[r][o] * inv);
            }
        }
        emit(r, m);
    }
}

template <int KH, int KW, int RB, typename Args>
__device__ __forceinline__ bool strip_is_interior(const Args &a, long x_tile, long y0) {
    return x_tile >= 4 && x_tile + 256 + 4 <= a.cols && y0 - KH / 2 >= -(long)a.halo_top &&
           y0 + RB + KH / 2 <= a.rows + a.halo_bot && y0 + RB <= a.rows;
}

}  // namespace xrs
