// Focal mean for LARGE 0/1 masks whose rows are made of a few contiguous runs (circles, boxes, annuli):
// per-tile row prefix sums turn every run into two LDS reads, so a 25x25 circle costs 25 run
// differences per cell instead of 441 taps (SURVEY.md §7 "large masks are not bandwidth-bound if done
// naively").  Same results as the tap-by-tap kernels of kxk.hip (reference: _apply_numpy + _calc_mean,
// xrspatial/focal.py:268-326): float64 sums, NaN cells skipped and counted out, window clipped at the
// raster edge.
//
// Workgroup = 1024 threads (16 waves), output tile 16 x 128.
//   scan phase   wave w owns tile rows w, w+16, ...: each lane loads up to 3 consecutive cells of the
//                (128 + 2*rx)-wide input row straight from global memory, turns NaN / out-of-raster
//                cells into (0, count 0), forms its local prefix, and one wave64 shuffle scan of the
//                lane totals completes the row prefix; float64 sums P and int32 counts C go to LDS.
//   run phase    wave w produces output row w; lane l the columns l and l+64 (consecutive lanes ->
//                consecutive 8-byte LDS words: conflict-free ds_read_b64):
//                sum += P[row][x + e] - P[row][x + s] for every run (row, s, e) of the mask.
// Tiles that contain +-inf (prefix differences would give inf - inf) fall back to a direct tap loop
// over global memory for that tile only.
#include "xrs_common.h"

#include <cmath>

#include "wave_reduce.h"

using namespace xrs;

namespace {

constexpr int RTW = 128, RTH = 16, MAX_RUNS = 128, MAX_KR = 61;

struct RunArgs {
    const float *in;
    float *out;                  // mean (may be NULL in the mean+var kernel)
    float *out_var, *out_std;    // mean+var kernel only
    long rows, cols, ld_in, ld_out;
    int halo_top, halo_bot;
    int krows, kcols;
    int pp;                      // LDS row pitch of the prefix arrays (entries) = RTW + 2*rx + 2
    int nruns, ntaps;
    double inv_ntaps;
    long tiles_x, n_tiles;
    unsigned run_off_s[MAX_RUNS], run_off_e[MAX_RUNS];   // (ky * pp + s) and (ky * pp + e): taps [s, e) of kernel row ky
    unsigned long long mask_rows[MAX_KR];
};

__device__ __forceinline__ double rcp_count(int n) {
    const double c = (double)n;
    double r = __builtin_amdgcn_rcp(c);
    r = fma(fma(-c, r, 1.0), r, r);
    return n ? r : nan("");
}

// 1024 threads = 16 waves share one tile (the occupancy lever: two such workgroups fill a CU's 32 wave
// slots while the 74 KiB tile is paid once per 16 waves); wave w produces output row w.
__global__ void __launch_bounds__(1024, 8) focal_mean_runs_kernel(const RunArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const long t = xcd_tile(blockIdx.x, a.n_tiles, XCD_UNIT(XRS_XCD_LDS, a.tiles_x));
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const long X0 = tx * RTW, Y0 = ty * RTH;
    const int ry = a.krows / 2, rx = a.kcols / 2;
    const int trows = RTH + a.krows - 1, twl = RTW + 2 * rx;
    double *P = reinterpret_cast<double *>(smem);                      // [trows][pp], P[r][c] = sum of cells < c
    int *C = reinterpret_cast<int *>(P + (size_t)trows * a.pp);        // [trows][pp]
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..15

    // ---- scan phase: wave w owns tile rows w, w+16, ... (all loads of a wave issued before its first scan)
    bool saw_inf = false, saw_gap = false;
    const int per = (twl + 63) >> 6;                                   // cells per lane (<= 3)
    constexpr int CH = 5;                                              // rows per wave: covers RTH + 60 tile rows
    float v[CH][3];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int r = wv + 16 * i;
        const long y = Y0 - ry + r;
        const bool yok = r < trows && y >= y_lo && y < y_hi;
        const float *grow = a.in + y * a.ld_in;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = lane * per + k;
            const long x = X0 - rx + c;
            v[i][k] = nan_f32();
            if (yok && k < per && c < twl && x >= 0 && x < a.cols) v[i][k] = grow[x];
        }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int r = wv + 16 * i;
        if (r < trows) {                                               // wave-uniform
            double m[3] = {0.0, 0.0, 0.0};
            int cn[3] = {0, 0, 0};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c = lane * per + k;
                if (k < per && c < twl) {
                    const float x = v[i][k];
                    if (isnan(x)) { saw_gap = true; }
                    else if (isinf(x)) { saw_inf = true; }
                    else { m[k] = (double)x; cn[k] = 1; }
                }
            }
            const double l1 = m[0] + m[1], l2 = l1 + m[2];
            const int c1 = cn[0] + cn[1], c2 = c1 + cn[2];
            double sc[1] = {l2};                                       // inclusive wave64 scans of the lane totals
            wave_scan_f64<1>(sc);                                      // (wave_reduce.h: DPP cross-lane moves, no LDS traffic)
            const double tot = sc[0];
            const int ctot = wave_scan_i32(c2);
            const double base = tot - l2;                              // exclusive prefix of this lane
            const int cbase = ctot - c2;
            double *prow = P + (size_t)r * a.pp;
            int *crow = C + (size_t)r * a.pp;
            if (lane == 0) { prow[0] = 0.0; crow[0] = 0; }
            const double pk[3] = {base + m[0], base + l1, base + l2};
            const int ck[3] = {cbase + cn[0], cbase + c1, cbase + c2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c = lane * per + k;
                if (k < per && c < twl) { prow[c + 1] = pk[k]; crow[c + 1] = ck[k]; }
            }
        }
    }
    // (__syncthreads_or is a logical OR of predicates, hence one call per flag)
    const int flags = (__syncthreads_or(saw_inf) ? 2 : 0) | (__syncthreads_or(saw_gap) ? 1 : 0);

    const long y = Y0 + wv;                                            // this wave's output row
    if (y >= a.rows) return;
    if (flags & 2) {
        // ---- rare: the tile holds +-inf.  Direct tap loop from global memory (the reference's loop).
        for (int o = 0; o < 2; ++o) {
            const long x = X0 + lane + 64 * o;
            if (x >= a.cols) break;
            double s = 0.0;
            int n = 0;
            for (int ky = 0; ky < a.krows; ++ky) {
                const long yy = y - ry + ky;
                if (yy < y_lo || yy >= y_hi) continue;
                const unsigned long long bits = a.mask_rows[ky];
                for (int kx = 0; kx < a.kcols; ++kx) {
                    const long xx = x - rx + kx;
                    if (!(bits >> kx & 1ull) || xx < 0 || xx >= a.cols) continue;
                    const float val = a.in[yy * a.ld_in + xx];
                    if (!isnan(val)) { s += (double)val; ++n; }
                }
            }
            st_stream(&a.out[y * a.ld_out + x], (float)(s * rcp_count(n)));
        }
        return;
    }

    // ---- run phase: lane l -> columns l and l+64 of row wv (consecutive lanes, consecutive 8-byte words)
    const bool counted = flags & 1;                                   // some cell of the tile is NaN / outside
    double acc0 = 0.0, acc1 = 0.0;
    int cnt0 = 0, cnt1 = 0;
    const double *pbase = P + (size_t)wv * a.pp + lane;
    const int *cbase_ = C + (size_t)wv * a.pp + lane;
#pragma unroll 5
    for (int q = 0; q < a.nruns; ++q) {
        const unsigned os = a.run_off_s[q], oe = a.run_off_e[q];       // wave-uniform, precomputed on the host
        acc0 += pbase[oe] - pbase[os];
        acc1 += pbase[64 + oe] - pbase[64 + os];
        if (counted) {
            cnt0 += cbase_[oe] - cbase_[os];
            cnt1 += cbase_[64 + oe] - cbase_[64 + os];
        }
    }
    const long x0 = X0 + lane;
    if (x0 < a.cols) st_stream(&a.out[y * a.ld_out + x0], (float)(acc0 * (counted ? rcp_count(cnt0) : a.inv_ntaps)));
    if (x0 + 64 < a.cols) st_stream(&a.out[y * a.ld_out + x0 + 64], (float)(acc1 * (counted ? rcp_count(cnt1) : a.inv_ntaps)));
}

// Mean + variance + standard deviation from prefix sums (the all-statistics path for large masks).
// Same tile / scan / run structure as focal_mean_runs_kernel with a second prefix array of squares.
// Numerics: cells are shifted by the tile's centre value s before they are summed, so the one-pass
// ssd = sum(x'^2) - (sum x')^2 / n only cancels against the tile's LOCAL relief R = max|x - s|; its error is
// bounded by ~1e-15 * max(n, row length) * R^2, and any output whose ssd is not at least 1e6 times that bound
// (flat patches inside high-relief tiles, e.g. lakes) recomputes its squared deviations tap by tap from global
// memory -- the reference's two-pass nanvar (numba arraymath.py:1020-1040) -- so results keep <= 1e-6 parity.
__global__ void __launch_bounds__(1024, 4) focal_meanvar_runs_kernel(const RunArgs a) {   // one WG per CU (LDS): 4 waves/SIMD
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned r2max_bits;
    const long t = xcd_tile(blockIdx.x, a.n_tiles, XCD_UNIT(XRS_XCD_LDS, a.tiles_x));
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const long X0 = tx * RTW, Y0 = ty * RTH;
    const int ry = a.krows / 2, rx = a.kcols / 2;
    const int trows = RTH + a.krows - 1, twl = RTW + 2 * rx;
    double *P = reinterpret_cast<double *>(smem);                      // prefix of x'
    double *P2 = P + (size_t)trows * a.pp;                             // prefix of x'^2
    int *C = reinterpret_cast<int *>(P2 + (size_t)trows * a.pp);       // prefix of valid counts
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..15
    if (threadIdx.x == 0) r2max_bits = 0u;

    // shift: the tile's centre cell when it is a finite in-raster value, else 0
    double shift = 0.0;
    {
        const long yc = Y0 + RTH / 2, xc = X0 + RTW / 2;
        if (yc < a.rows && xc < a.cols) {
            const float c = a.in[yc * a.ld_in + xc];
            if (isfinite(c)) shift = (double)c;
        }
    }
    __syncthreads();

    bool saw_inf = false, saw_gap = false;
    float r2 = 0.f;
    const int per = (twl + 63) >> 6;
    constexpr int CH = 5;
    float v[CH][3];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int r = wv + 16 * i;
        const long y = Y0 - ry + r;
        const bool yok = r < trows && y >= y_lo && y < y_hi;
        const float *grow = a.in + y * a.ld_in;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = lane * per + k;
            const long x = X0 - rx + c;
            v[i][k] = nan_f32();
            if (yok && k < per && c < twl && x >= 0 && x < a.cols) v[i][k] = grow[x];
        }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int r = wv + 16 * i;
        if (r < trows) {
            double m[3] = {0.0, 0.0, 0.0}, q[3] = {0.0, 0.0, 0.0};
            int cn[3] = {0, 0, 0};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c = lane * per + k;
                if (k < per && c < twl) {
                    const float x = v[i][k];
                    if (isnan(x)) { saw_gap = true; }
                    else if (isinf(x)) { saw_inf = true; }
                    else {
                        m[k] = (double)x - shift; q[k] = m[k] * m[k]; cn[k] = 1;
                        r2 = fmaxf(r2, (float)q[k]);
                    }
                }
            }
            const double l1 = m[0] + m[1], l2 = l1 + m[2];
            const double q1 = q[0] + q[1], q2 = q1 + q[2];
            const int c1 = cn[0] + cn[1], c2 = c1 + cn[2];
            double sc[2] = {l2, q2};
            wave_scan_f64<2>(sc);
            const double tot = sc[0], qtot = sc[1];
            const int ctot = wave_scan_i32(c2);
            const double base = tot - l2, qbase = qtot - q2;
            const int cbase = ctot - c2;
            double *prow = P + (size_t)r * a.pp, *qrow = P2 + (size_t)r * a.pp;
            int *crow = C + (size_t)r * a.pp;
            if (lane == 0) { prow[0] = 0.0; qrow[0] = 0.0; crow[0] = 0; }
            const double pk[3] = {base + m[0], base + l1, base + l2};
            const double qk[3] = {qbase + q[0], qbase + q1, qbase + q2};
            const int ck[3] = {cbase + cn[0], cbase + c1, cbase + c2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c = lane * per + k;
                if (k < per && c < twl) { prow[c + 1] = pk[k]; qrow[c + 1] = qk[k]; crow[c + 1] = ck[k]; }
            }
        }
    }
    atomicMax(&r2max_bits, __float_as_uint(r2));                       // non-negative floats order like their bit patterns
    const bool tile_inf = __syncthreads_or(saw_inf);
    const bool counted = __syncthreads_or(saw_gap);
    const double r2max = (double)__uint_as_float(r2max_bits);

    const long y = Y0 + wv;
    if (y >= a.rows) return;
    const int longest = a.ntaps > twl ? a.ntaps : twl;
    const double guard = 1e-15 * (double)longest * r2max * 1e6;

    // exact statistics of one output straight from global memory (the reference's loops)
    auto exact = [&](long x, double &mean, double &var) {
        double s = 0.0;
        int n = 0;
        for (int ky = 0; ky < a.krows; ++ky) {
            const long yy = y - ry + ky;
            if (yy < y_lo || yy >= y_hi) continue;
            const unsigned long long bits = a.mask_rows[ky];
            for (int kx = 0; kx < a.kcols; ++kx) {
                const long xx = x - rx + kx;
                if (!(bits >> kx & 1ull) || xx < 0 || xx >= a.cols) continue;
                const float val = a.in[yy * a.ld_in + xx];
                if (!isnan(val)) { s += (double)val; ++n; }
            }
        }
        mean = n ? s / (double)n : nan("");               // true division: a flat window must give its value exactly
        double ssd = 0.0;
        for (int ky = 0; ky < a.krows; ++ky) {
            const long yy = y - ry + ky;
            if (yy < y_lo || yy >= y_hi) continue;
            const unsigned long long bits = a.mask_rows[ky];
            for (int kx = 0; kx < a.kcols; ++kx) {
                const long xx = x - rx + kx;
                if (!(bits >> kx & 1ull) || xx < 0 || xx >= a.cols) continue;
                const float val = a.in[yy * a.ld_in + xx];
                if (!isnan(val)) { const double d = (double)val - mean; ssd += d * d; }
            }
        }
        var = n ? ssd / (double)n : nan("");
    };

    double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
    int cnt[2] = {0, 0};
    if (!tile_inf) {
        const double *pbase = P + (size_t)wv * a.pp + lane;
        const double *qbase_ = P2 + (size_t)wv * a.pp + lane;
        const int *cbase_ = C + (size_t)wv * a.pp + lane;
#pragma unroll 5
        for (int q = 0; q < a.nruns; ++q) {
            const unsigned os = a.run_off_s[q], oe = a.run_off_e[q];
            s1[0] += pbase[oe] - pbase[os];
            s1[1] += pbase[64 + oe] - pbase[64 + os];
            s2[0] += qbase_[oe] - qbase_[os];
            s2[1] += qbase_[64 + oe] - qbase_[64 + os];
            if (counted) {
                cnt[0] += cbase_[oe] - cbase_[os];
                cnt[1] += cbase_[64 + oe] - cbase_[64 + os];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const long x = X0 + lane + 64 * o;
        if (x >= a.cols) continue;
        double mean, var;
        bool need_exact = tile_inf;
        if (!tile_inf) {
            const double inv = counted ? rcp_count(cnt[o]) : a.inv_ntaps;
            const double mshift = s1[o] * inv;                          // mean of the shifted values
            const double ssd = s2[o] - s1[o] * mshift;
            mean = shift + mshift;
            var = (ssd > 0.0 ? ssd : 0.0) * inv;
            if (counted && cnt[o] == 0) var = nan("");
            need_exact = !(ssd >= guard) && !(counted && cnt[o] == 0);  // ill-conditioned (or exactly flat) window
        }
        if (need_exact) exact(x, mean, var);
        const long off = y * a.ld_out + x;
        if (a.out) a.out[off] = (float)mean;
        if (a.out_var) a.out_var[off] = (float)var;
        if (a.out_std) a.out_std[off] = (float)sqrt(var);
    }
}

}  // namespace

namespace {

// Parse the 0/1 mask into runs; false if it does not suit the prefix-sum kernels.
bool parse_runs(RunArgs &a, const double *kernel, int krows, int kcols) {
    if (krows > MAX_KR || kcols > MAX_KR) return false;
    for (int ky = 0; ky < krows; ++ky) {
        int kx = 0;
        while (kx < kcols) {
            while (kx < kcols && kernel[ky * kcols + kx] != 1.0) ++kx;
            if (kx >= kcols) break;
            const int s0 = kx;
            while (kx < kcols && kernel[ky * kcols + kx] == 1.0) ++kx;
            if (a.nruns >= MAX_RUNS) return false;
            a.run_off_s[a.nruns] = (unsigned)s0 | (unsigned)ky << 16;     // pitch applied below
            a.run_off_e[a.nruns] = (unsigned)kx | (unsigned)ky << 16;
            a.ntaps += kx - s0;
            a.mask_rows[ky] |= ((kx - s0) >= 64 ? ~0ull : ((1ull << (kx - s0)) - 1)) << s0;
            ++a.nruns;
        }
    }
    if (a.nruns == 0 || a.nruns * 6 > a.ntaps) return false;           // runs only pay off for long rows
    a.krows = krows; a.kcols = kcols;
    a.pp = RTW + 2 * (kcols / 2) + 2;
    for (int q = 0; q < a.nruns; ++q) {
        a.run_off_s[q] = (a.run_off_s[q] >> 16) * a.pp + (a.run_off_s[q] & 0xffff);
        a.run_off_e[q] = (a.run_off_e[q] >> 16) * a.pp + (a.run_off_e[q] & 0xffff);
    }
    a.inv_ntaps = 1.0 / a.ntaps;
    return true;
}

template <typename K>
int launch_runs(K kernel_fn, RunArgs &a, size_t bytes_per_cell, hipStream_t s) {
    const size_t lds = (size_t)(RTH + a.krows - 1) * a.pp * bytes_per_cell;
    if (lds > 150 * 1024) return -1;
    a.tiles_x = (a.cols + RTW - 1) / RTW;
    a.n_tiles = a.tiles_x * ((a.rows + RTH - 1) / RTH);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel_fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail("hipFuncSetAttribute(max dynamic LDS %zu) failed: %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel_fn, dim3((unsigned)xcd_grid(a.n_tiles, XCD_UNIT(XRS_XCD_LDS, a.tiles_x))), dim3(1024), lds, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace xrs {

// Both return 0 if launched, -1 if the mask does not fit the kernel (caller falls back), > 0 on error.
int try_launch_focal_mean_runs(const float *in, float *out, long rows, long cols, long ld_in, long ld_out,
                               const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                               hipStream_t s) {
    RunArgs a;
    memset(&a, 0, sizeof(a));
    if (!parse_runs(a, kernel, krows, kcols)) return -1;
    a.in = in; a.out = out; a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out;
    a.halo_top = halo_top; a.halo_bot = halo_bot;
    return launch_runs(focal_mean_runs_kernel, a, sizeof(double) + sizeof(int), s);
}

int try_launch_focal_meanvar_runs(const float *in, float *out_mean, float *out_var, float *out_std, long rows,
                                  long cols, long ld_in, long ld_out, const double *kernel, int krows, int kcols,
                                  int halo_top, int halo_bot, hipStream_t s) {
    RunArgs a;
    memset(&a, 0, sizeof(a));
    if (!parse_runs(a, kernel, krows, kcols)) return -1;
    a.in = in; a.out = out_mean; a.out_var = out_var; a.out_std = out_std;
    a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out;
    a.halo_top = halo_top; a.halo_bot = halo_bot;
    return launch_runs(focal_meanvar_runs_kernel, a, 2 * sizeof(double) + sizeof(int), s);
}

}  // namespace xrs
