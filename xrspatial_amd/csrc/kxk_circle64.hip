// Float64 statistics (mean, var, std) of focal_stats / focal.apply for CIRCULAR masks of radius 4..12 cells:
// the column walker of kxk_circle.hip applied to the moments.
//
// Reference semantics (xrspatial/focal.py:226-258, 268-326): numba nanmean / nanvar / nanstd over the cells under
// `kernel == 1` -- float64 accumulation, NaN cells skipped, var = two-pass mean of squared deviations -- rounded
// to float32 at the store.
//
// Per input row y' a lane (one column) forms, from the centre outwards, the float64 sum S_h and sum of squares
// Q_h of the SHIFTED values d = v - c over the centred run of half-width h, plus the count C_h of valid cells
// (2 adds + 2 fma + 1 integer add per level); output row y' - dy adds (S, Q, C) of h = hw(dy) to its ring slot:
// 25 triple-adds per row and column instead of 441 taps (or 25 prefix differences through LDS, kxk_runs.hip).
// c is the lane's own column value at the middle row of the tile: every output is produced by one lane, so the
// shift may differ per lane, and a nearby value keeps d small.  At the end
//     mean = c + S/n,   var = (Q - S^2/n)/n
// The one-pass variance is guarded exactly like kxk_runs.hip: if the result is not comfortably above the
// rounding noise of its operands (flat patches inside high-relief tiles, or +-inf under the window) the output
// is recomputed tap by tap with the reference's two-pass loops.  Sums are float64 and re-associated relative
// to the reference's row-major order: invisible after the float32 rounding of the result (tests: rtol 1e-6).
#include "xrs_common.h"

#include <cmath>

using namespace xrs;

namespace {

constexpr int CTH = 128;

struct Circle64Args {
    const float *in;
    float *out_mean, *out_var, *out_std;     // any may be NULL
    long rows, cols, ld_in, ld_out;
    int halo_top, halo_bot;
    long tiles_x, n_tiles;
};

constexpr int half_width(int R, int dy) {
    int h = 0;
    while ((h + 1) * (h + 1) + dy * dy <= R * R) ++h;
    return h;
}

constexpr int circle_taps(int R) {
    int n = 0;
    for (int dy = -R; dy <= R; ++dy) n += 2 * half_width(R, dy < 0 ? -dy : dy) + 1;
    return n;
}

__device__ __forceinline__ double rcp_n(int n) {
    const double c = (double)n;
    double r = __builtin_amdgcn_rcp(c);
    r = fma(fma(-c, r, 1.0), r, r);
    return n ? r : nan("");
}

template <int R>
__global__ void __launch_bounds__(256) focal_circle_f64_kernel(const Circle64Args a) {
    constexpr int K = 2 * R + 1;
    const long t = xcd_tile(blockIdx.x, a.n_tiles);
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long xw = tx * 256 + wv * 64;
    const long x = xw + lane;
    const long y0 = ty * CTH;
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    const long y_end = (y0 + CTH < a.rows ? y0 + CTH : a.rows);
    if (xw >= a.cols) return;
    const float qnan = nan_f32();

    // per-lane shift: this column's value in the middle of the tile (0 if not a finite in-raster cell)
    float cf = 0.0f;
    {
        const long yc = (y0 + CTH / 2 < a.rows ? y0 + CTH / 2 : a.rows - 1);
        if (x < a.cols) {
            const float c0 = a.in[yc * a.ld_in + x];
            if (isfinite(c0)) cf = c0;
        }
    }
    const double shift = (double)cf;

    double sd[K], sq[K];          // ring: slot j belongs to output row (current input row) - (j - R)
    int cn[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { sd[j] = 0.0; sq[j] = 0.0; cn[j] = 0; }
    float amax = 0.0f;            // running max |v - c| over everything this lane has read (guard scale)

    for (long yy = y0 - R; yy < y_end + R; ++yy) {
        float v[K];
        const bool row_ok = yy >= y_lo && yy < y_hi;
        if (row_ok) {
            const float *p = a.in + yy * a.ld_in + xw + lane;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const long xc = x + k - R;
                v[k] = (xc >= 0 && xc < a.cols) ? p[k - R] : qnan;
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = qnan;
        }

        double S = 0.0, Q = 0.0;
        int C = 0;
#pragma unroll
        for (int h = 0; h <= R; ++h) {
#pragma unroll
            for (int side = 0; side < (h == 0 ? 1 : 2); ++side) {
                const float val = v[side == 0 ? R - h : R + h];
                const bool ok = !isnan(val);
                const double d = ok ? (double)val - shift : 0.0;
                S += d;
                Q = fma(d, d, Q);
                C += ok ? 1 : 0;
                amax = fmaxf(amax, isfinite(val) ? fabsf(val - cf) : 0.0f);   // (+-inf: the sums go non-finite -> exact path)
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int dy = j - R;
                if (half_width(R, dy < 0 ? -dy : dy) == h) { sd[j] += S; sq[j] += Q; cn[j] += C; }
            }
        }

        const long yo = yy - R;
        if (yo >= y0 && x < a.cols) {
            const int n = cn[2 * R];
            const double inv = rcp_n(n);
            const double ms = sd[2 * R] * inv;                          // mean of the shifted values
            const double ssd = sq[2 * R] - sd[2 * R] * ms;
            double mean = shift + ms;
            double var = (ssd > 0.0 ? ssd : 0.0) * inv;
            // rounding noise of Q and S^2/n is ~ ntaps * eps * max(d^2); 1e6 of headroom as in kxk_runs.hip
            const double guard = 1e-9 * (double)circle_taps(R) * ((double)amax * (double)amax);
            if (n != 0 && !(ssd >= guard)) {
                // ill-conditioned / exactly flat window, or +-inf under it: the reference's two-pass loops
                double s = 0.0;
                int m = 0;
                for (int ky = 0; ky < K; ++ky) {
                    const long yr = yo - R + ky;
                    if (yr < y_lo || yr >= y_hi) continue;
                    const int h = half_width(R, ky < R ? R - ky : ky - R);
                    for (int kx = R - h; kx <= R + h; ++kx) {
                        const long xr = x - R + kx;
                        if (xr < 0 || xr >= a.cols) continue;
                        const float val = a.in[yr * a.ld_in + xr];
                        if (!isnan(val)) { s += (double)val; ++m; }
                    }
                }
                mean = m ? s / (double)m : nan("");          // true division: a flat window must give its value exactly
                double dev = 0.0;
                for (int ky = 0; ky < K; ++ky) {
                    const long yr = yo - R + ky;
                    if (yr < y_lo || yr >= y_hi) continue;
                    const int h = half_width(R, ky < R ? R - ky : ky - R);
                    for (int kx = R - h; kx <= R + h; ++kx) {
                        const long xr = x - R + kx;
                        if (xr < 0 || xr >= a.cols) continue;
                        const float val = a.in[yr * a.ld_in + xr];
                        if (!isnan(val)) { const double d = (double)val - mean; dev += d * d; }
                    }
                }
                var = m ? dev / (double)m : nan("");
            }
            const long off = yo * a.ld_out + x;
            if (a.out_mean) a.out_mean[off] = (float)mean;
            if (a.out_var) a.out_var[off] = (float)var;
            if (a.out_std) a.out_std[off] = (float)sqrt(var);
        }
#pragma unroll
        for (int j = K - 1; j > 0; --j) { sd[j] = sd[j - 1]; sq[j] = sq[j - 1]; cn[j] = cn[j - 1]; }
        sd[0] = 0.0; sq[0] = 0.0; cn[0] = 0;
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

template <int R>
int launch_circle64(Circle64Args &a, const double *kernel, hipStream_t s) {
    if (!is_circle<R>(kernel)) return -1;
    a.tiles_x = (a.cols + 255) / 256;
    a.n_tiles = a.tiles_x * ((a.rows + CTH - 1) / CTH);
    const long grid = xcd_grid(a.n_tiles);
    if (grid > 0x7fffffffL) return fail("focal circle: raster too large for one launch");
    hipLaunchKernelGGL((focal_circle_f64_kernel<R>), dim3((unsigned)grid), dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace xrs {

// 0 = launched, -1 = not a circle this file is instantiated for, > 0 = error
int try_launch_focal_circle_f64(const float *in, float *out_mean, float *out_var, float *out_std, long rows, long cols,
                                long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top,
                                int halo_bot, hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    if (!out_mean && !out_var && !out_std) return 0;
    Circle64Args a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.out_mean = out_mean; a.out_var = out_var; a.out_std = out_std;
    a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out;
    a.halo_top = halo_top; a.halo_bot = halo_bot;
    switch (krows / 2) {
        case 4: return launch_circle64<4>(a, kernel, s);
        case 5: return launch_circle64<5>(a, kernel, s);
        case 6: return launch_circle64<6>(a, kernel, s);
        case 7: return launch_circle64<7>(a, kernel, s);
        case 8: return launch_circle64<8>(a, kernel, s);
        case 9: return launch_circle64<9>(a, kernel, s);
        case 10: return launch_circle64<10>(a, kernel, s);
        case 11: return launch_circle64<11>(a, kernel, s);
        case 12: return launch_circle64<12>(a, kernel, s);
        default: return -1;
    }
}

}  // namespace xrs
