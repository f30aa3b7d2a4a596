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

template <int KH, int KW, int RB, typename Args>
__device__ __forceinline__ bool strip_is_interior(const Args &a, long x_tile, long y0) {
    return x_tile >= 4 && x_tile + 256 + 4 <= a.cols && y0 - KH / 2 >= -(long)a.halo_top &&
           y0 + RB + KH / 2 <= a.rows + a.halo_bot && y0 + RB <= a.rows;
}

}  // namespace xrs
