// Float32 statistics (row-major sum, max, min, range) of focal_stats / focal.apply for CIRCULAR masks of
// radius 4..12 cells (9x9 .. 25x25, `circle_kernel` on square cells) -- the large-mask case where the
// tap-by-tap walk of kxk.hip spends 6 VALU + 6 SALU instructions per tap (profiles/r01/pmc_focal25_sum.json).
//
// Reference semantics (xrspatial/focal.py:268-326 with the numba reducers :226-258): over the cells under
// `kernel == 1`, in row-major order, NaN cells skipped, window clipped at the raster edge:
//   sum    float32 accumulator, sequential adds (numba nansum keeps the array dtype) -- the rounding of every
//          partial sum is part of the result, so the 441 adds per cell cannot be shared or re-associated;
//   max / min / range   order-free.
//
// Column walker.  A lane owns ONE column and walks down the input rows of its tile.  Every input row y' is
// read once per column (2R+1 neighbouring cells, L1 hits) and contributes to the 2R+1 output rows y' - dy:
//   * min / max: the running minimum over the centred run of half-width h, m_h = min(m_{h-1}, v[-h], v[+h]),
//     costs R `v_min3` per row and is exactly what output row y' - dy needs for h = hw(dy) -- 2R+1 more
//     `v_min` instead of one per tap (441 -> 37 per cell for R = 12);
//   * sum: output row y' - dy appends v[-hw(dy)] .. v[+hw(dy)] left to right to its float32 accumulator; the
//     rows dy and -dy append the SAME values in the SAME order, so their two accumulators share one packed
//     `v_pk_add_f32` per value (441 -> 233 instructions per cell).
// The 2R+1 partial results per statistic live in a register ring that shifts by one slot per input row (the
// loop body is the same for every row; the circle's half-widths are compile-time constants, which is what
// makes every register index static).  NaN cells: added as +0.0 (exact: a float32 accumulator that starts at
// +0.0 is never -0.0) and skipped by IEEE minNum / maxNum; a window without a valid cell gives sum 0 and NaN
// for max / min / range, like the reference.  Rows / columns outside the raster (or the shard's halo) read as NaN.
//
// 256-thread workgroup = 4 waves side by side = 256 columns x TH rows; no LDS, no barriers.
#include "xrs_common.h"

#include <cmath>

using namespace xrs;

namespace {

constexpr int CTH = 128;     // output rows per tile (input rows walked: CTH + 2R)

struct CircleArgs {
    const float *in;
    float *out_sum, *out_max, *out_min, *out_range;    // any may be NULL
    long rows, cols, ld_in, ld_out;
    int halo_top, halo_bot;
    long tiles_x, n_tiles;
};

// half-width of the circle's row dy: largest dx with dx^2 + dy^2 <= R^2 (the division-free test of
// convolution.py:144 on square cells)
constexpr int half_width(int R, int dy) {
    int h = 0;
    while ((h + 1) * (h + 1) + dy * dy <= R * R) ++h;
    return h;
}

typedef float v2f __attribute__((ext_vector_type(2)));

template <int R, bool EDGE, bool WANT_SUM, bool WANT_MM>
__global__ void __launch_bounds__(256) focal_circle_f32_kernel(const CircleArgs a) {
    constexpr int K = 2 * R + 1;
    const long t = xcd_tile(blockIdx.x, a.n_tiles);
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long xw = tx * 256 + wv * 64;                 // first column of this wave (scalar)
    const long x = xw + lane;
    const long y0 = ty * CTH;
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    const long y_end = (y0 + CTH < a.rows ? y0 + CTH : a.rows);        // output rows [y0, y_end)
    if (xw >= a.cols) return;
    const float qnan = nan_f32();

    // ring slot j holds the partial results of output row (current input row) - (j - R)
    v2f sp[R];                   // (slot j, slot 2R - j) for j < R: rows dy = j - R and -dy append identical values
    float sc = 0.0f;             // slot R (dy = 0)
    float mn[K], mx[K];
#pragma unroll
    for (int j = 0; j < R; ++j) sp[j] = (v2f)(0.0f);
#pragma unroll
    for (int j = 0; j < K; ++j) { mn[j] = INFINITY; mx[j] = -INFINITY; }

    for (long yy = y0 - R; yy < y_end + R; ++yy) {
        // ---- the 2R+1 cells of input row yy around this lane's column
        float v[K];
        const bool row_ok = yy >= y_lo && yy < y_hi;                    // wave-uniform
        if (row_ok) {
            const float *p = a.in + yy * a.ld_in + xw + lane;           // scalar row base + lane
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (EDGE) {
                    const long xc = x + k - R;
                    v[k] = (xc >= 0 && xc < a.cols) ? p[k - R] : qnan;
                } else {
                    v[k] = p[k - R];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = qnan;
        }

        // ---- max / min: running extrema over centred runs, handed to the slots whose row has that half-width
        if (WANT_MM) {
            float lo = v[R], hi = v[R];
#pragma unroll
            for (int h = 0; h <= R; ++h) {
                if (h > 0) {
                    lo = fminf(fminf(lo, v[R - h]), v[R + h]);
                    hi = fmaxf(fmaxf(hi, v[R - h]), v[R + h]);
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int dy = j - R;
                    if (half_width(R, dy < 0 ? -dy : dy) == h) {
                        mn[j] = fminf(mn[j], lo);
                        mx[j] = fmaxf(mx[j], hi);
                    }
                }
            }
        }

        // ---- sum: every slot appends its run of this row, left to right, in float32
        if (WANT_SUM) {
            float z[K];
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] = isnan(v[k]) ? 0.0f : v[k];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int h = half_width(R, R - j);
#pragma unroll
                for (int k = R - h; k <= R + h; ++k) sp[j] += (v2f)(z[k]);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) sc += z[k];
        }

        // ---- output row yy - R is complete (slot 2R); then the ring advances one slot
        const long yo = yy - R;
        if (yo >= y0 && (!EDGE || x < a.cols)) {
            const long off = yo * a.ld_out + x;
            if (WANT_SUM && a.out_sum) a.out_sum[off] = sp[0].y;
            if (WANT_MM) {
                const bool none = mn[2 * R] > mx[2 * R];                // no valid cell under the window
                if (a.out_max) a.out_max[off] = none ? qnan : mx[2 * R];
                if (a.out_min) a.out_min[off] = none ? qnan : mn[2 * R];
                if (a.out_range) a.out_range[off] = none ? qnan : mx[2 * R] - mn[2 * R];
            }
        }
        if (WANT_SUM) {
            // slots 0..R-1 are the .x halves (moving up), slots R+1..2R the .y halves of sp[2R - slot] (moving "down" the array)
#pragma unroll
            for (int j = 0; j + 1 < R; ++j) sp[j].y = sp[j + 1].y;      // slot 2R-j <- slot 2R-j-1
            const float old_c = sc;
            sc = sp[R - 1].x;                                           // slot R <- slot R-1
            sp[R - 1].y = old_c;                                        // slot R+1 <- slot R
#pragma unroll
            for (int j = R - 1; j > 0; --j) sp[j].x = sp[j - 1].x;      // slot j <- slot j-1
            sp[0].x = 0.0f;
        }
        if (WANT_MM) {
#pragma unroll
            for (int j = K - 1; j > 0; --j) { mn[j] = mn[j - 1]; mx[j] = mx[j - 1]; }
            mn[0] = INFINITY; mx[0] = -INFINITY;
        }
    }
}

template <int R>
bool is_circle(const double *kernel) {
    constexpr int K = 2 * R + 1;
    for (int ky = 0; ky < K; ++ky) {
        const int dy = ky < R ? R - ky : ky - R, h = half_width(R, dy);
        for (int kx = 0; kx < K; ++kx) {
            const int dx = kx < R ? R - kx : kx - R;
            if ((kernel[ky * K + kx] == 1.0) != (dx <= h)) return false;
        }
    }
    return true;
}

template <int R, bool WANT_SUM, bool WANT_MM>
int launch_circle(CircleArgs &a, hipStream_t s) {
    a.tiles_x = (a.cols + 255) / 256;
    a.n_tiles = a.tiles_x * ((a.rows + CTH - 1) / CTH);
    const long grid = xcd_grid(a.n_tiles);
    if (grid > 0x7fffffffL) return fail("focal circle: raster too large for one launch");
    // (one instantiation with column predicates for every tile keeps the code small; the predicates are cheap
    //  next to the 2R+1 dependent float32 adds per slot -- interior-only specialisation measured no gain)
    hipLaunchKernelGGL((focal_circle_f32_kernel<R, true, WANT_SUM, WANT_MM>), dim3((unsigned)grid), dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <int R>
int dispatch_circle(CircleArgs &a, hipStream_t s) {
    const bool ws = a.out_sum, wm = a.out_max || a.out_min || a.out_range;
    if (ws && wm) return launch_circle<R, true, true>(a, s);
    if (ws) return launch_circle<R, true, false>(a, s);
    return launch_circle<R, false, true>(a, s);
}

}  // namespace

namespace xrs {

// 0 = launched, -1 = not a circle this file is instantiated for (caller walks the taps), > 0 = error
int try_launch_focal_circle_f32(const float *in, float *out_sum, float *out_max, float *out_min, float *out_range,
                                long rows, long cols, long ld_in, long ld_out, const double *kernel, int krows,
                                int kcols, int halo_top, int halo_bot, hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    if (!out_sum && !out_max && !out_min && !out_range) return 0;
    CircleArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.out_sum = out_sum; a.out_max = out_max; a.out_min = out_min; a.out_range = out_range;
    a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out;
    a.halo_top = halo_top; a.halo_bot = halo_bot;
#define XRS_CIRCLE(R_) \
    case R_: return is_circle<R_>(kernel) ? dispatch_circle<R_>(a, s) : -1;
    switch (krows / 2) {
        XRS_CIRCLE(4) XRS_CIRCLE(5) XRS_CIRCLE(6) XRS_CIRCLE(7) XRS_CIRCLE(8) XRS_CIRCLE(9) XRS_CIRCLE(10)
        XRS_CIRCLE(11) XRS_CIRCLE(12)
        default: return -1;
    }
#undef XRS_CIRCLE
}

}  // namespace xrs
