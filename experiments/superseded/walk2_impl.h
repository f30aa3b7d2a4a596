// All seven focal statistics over a large circular / box mask in ONE pass: focal_stats(agg, circle_kernel(...)) with
// windows 9x9 .. 25x25 (xrspatial/focal.py:782-797 runs seven apply() passes, each gathering the window per cell and
// calling a numba reducer, :268-302: nanmean / nanvar / nanstd with float64 accumulators, nansum, nanmin, nanmax).
//
// Second generation of the column walker of circle_walk.h (which stays as the exact NaN-skipping path):
//   * a lane owns ONE column of a 64-column x W2TH-row wave tile and walks down; every input row is loaded once per
//     wave (one dword per lane + 2R halo lanes, prefetched 3-5 rows ahead), staged through LDS, and each lane reads the
//     2R+1 cells around its column back (ds_read2_b32 pairs).
//   * the 2R+1 output rows in flight live in register rings indexed (row - dy) mod (2R+1); the row loop is unrolled
//     2R+1 times so every index is a compile-time constant and the rings never move (the first generation shifts them
//     every row: ~150 of its ~400 instructions per cell and row are v_mov).
//   * no per-cell NaN bookkeeping: the fast path assumes finite cells and a constant count; a non-finite window sum
//     at emit time hands the whole tile to the exact walker.  Raster edges stay on the fast path (out-of-raster cells
//     contribute nothing, the divisor is the geometric count of in-raster cells).
//   * moments: float64 sums S and Q of d = v - c (c = the cell at the wave tile's centre) over centred runs, built
//     from the centre outwards; the LOADING lane forms d once per cell and stages it as float64 next to the raw
//     float32 value, so the 2R+1 readers of a cell neither convert nor subtract.  mean = c + S/n,
//     var = (Q - S^2/n)/n, guarded against cancellation like the first generation (a tile with a flat or
//     ill-conditioned window has its moments redone by the exact walker, which runs the reference's two-pass loops).
//   * sum = n*c + S rounded once to float32.  The reference adds the taps sequentially in float32 (numba nansum keeps
//     the array dtype), so its own result carries a rounding error of up to (n-1) * 2^-24 * sum|v|; this one is the
//     exactly rounded sum, always within that bound of the reference and within 1e-5 relative whenever the window does
//     not cancel.  XRS_FOCAL_SUM=sequential selects the bit-exact sequential kernel (kxk_circle.hip) instead.
//   * max / min / range: running extrema over centred runs (v_min3 / v_max3), float32, exact.
//   * subsets: the extrema pass and the moments pass are compile-time options (XRS_WALK2_MM / XRS_WALK2_MOM), so
//     apply(func=_calc_max) or focal_stats(['mean', 'std']) on a 25x25 window pay for one pass only; the extrema-only
//     variant skips NaN cells like nanmin / nanmax do and only leaves tiles that hold a NaN to the exact walker (an
//     all-NaN window must come out NaN).
// Included by kxk_circle2*.hip and kxk_box2*.hip, which define XRS_WALK_SHAPE / XRS_WALK_KERNEL / XRS_WALK_ENTRY and the
// variant.
#include "circle_walk.h"

#include <utility>

using namespace xrs;

namespace {

constexpr int W2TH = 128;         // output rows per tile
#ifndef XRS_WALK2_MM
#define XRS_WALK2_MM 1            // max / min / range
#endif
#ifndef XRS_WALK2_MOM
#define XRS_WALK2_MOM 1           // mean / var / std / sum
#endif
static_assert(XRS_WALK2_MM || XRS_WALK2_MOM, "nothing to compute");
#ifndef XRS_WALK2_WAVES
#define XRS_WALK2_WAVES 2         // workgroups per CU = waves per SIMD
#endif

template <int R, typename Shape>
struct Walk2Cfg {
    static constexpr int K = 2 * R + 1;
    static constexpr int STG = 64 + 2 * R;
    static constexpr int NTAPS = shape_taps<Shape>(R);
#ifndef XRS_WALK2_U
#define XRS_WALK2_U (2 * R + 1)
#endif
    // rows per unrolled round (rings rotated by U after each).  U = 2R+1: no rotation at all.  Measured for R = 12, all
    // seven statistics (profiles/r02): U = 5 6.6 ms (the rotation's parallel copy spills at the 256-register budget),
    // U = 10 4.3 ms, U = 25 3.4 ms.
    static constexpr int U = XRS_WALK2_U;
    static constexpr int PFN = (U % 5 == 0) ? 5 : (U % 4 == 0) ? 4 : 3;       // rows prefetched into registers
    static constexpr bool ROT = U % PFN == 0;              // prefetch slot = phase mod PFN (otherwise the slots shift)
    static constexpr bool level_used(int h) {
        for (int dy = 0; dy <= R; ++dy)
            if (Shape::hw(R, dy) == h) return true;
        return false;
    }
};

// v_min / v_max without the canonicalising v_max x, x the compiler adds in front of fminf / fmaxf: quiet NaNs (the
// out-of-raster fill) are skipped by the hardware instructions as they are, and a tile with a NaN of its own is redone
// by the exact walker anyway.
__device__ __forceinline__ float raw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float raw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float raw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float raw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// per wave and row buffer: STG float64 shifted cells, then STG raw float32 cells
template <int R, typename Shape, bool EDGE>
struct Walk2 {
    using C = Walk2Cfg<R, Shape>;
    static constexpr int K = C::K, PFN = C::PFN;
    static constexpr bool MM = XRS_WALK2_MM != 0, MOM = XRS_WALK2_MOM != 0;

    double sd[MOM ? K : 1], sq[MOM ? K : 1];
    float mn[MM ? K : 1], mx[MM ? K : 1];
    float pf_own[PFN], pf_halo[PFN];
    float dmax;                    // largest |v - c| among the cells this lane loaded (guard scale, reduced over the wave at the end)
    double vmin;                   // smallest unclamped variance this lane emitted
    bool bad;
    int i;

    const WalkGeom &g;
    const WalkOuts &o;
    char *lds;                     // this wave's row buffer
    long xw, x, y0, y_end, y_first;
    int n_in, lane;
    float cf;
    double cd;
    int kmin, kmax;                // EDGE: in-raster part of the lane's row window, v[kmin .. kmax]
    int n_full;                    // EDGE: cell count of a window whose rows are all inside

    __device__ __forceinline__ Walk2(const WalkGeom &g_, const WalkOuts &o_, char *lds_, long xw_, long y0_, long ye, int lane_)
        : g(g_), o(o_), lds(lds_), xw(xw_), x(xw_ + lane_), y0(y0_), y_end(ye), lane(lane_) {}

    __device__ __forceinline__ void load_row(int il, float &own, float &halo) const {
        // staged cell s <-> raster column xw - R + s; lane loads s = lane, lanes < 2R also s = 64 + lane
        const long yy = y_first + il;
        const float *p = g.in + yy * g.ld_in + (xw - R);
        if (!EDGE) {
            own = p[lane];
            halo = cf;                                          // (lanes without a halo cell: d = 0 for the guard scale)
            if (lane < 2 * R) halo = p[64 + lane];
            return;
        }
        own = halo = cf;                                    // (out-of-raster cells are masked by kmin / kmax, never used)
        const bool row_ok = il < n_in && yy >= -(long)g.halo_top && yy < g.rows + g.halo_bot;     // wave-uniform
        if (!row_ok) return;
        const long xa = xw - R + lane, xb = xa + 64;
        if (xa >= 0 && xa < g.cols) own = p[lane];
        if (lane < 2 * R && xb >= 0 && xb < g.cols) halo = p[64 + lane];
    }

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < (MOM ? K : 1); ++j) { sd[j] = 0.0; sq[j] = 0.0; }
#pragma unroll
        for (int j = 0; j < (MM ? K : 1); ++j) { mn[j] = INFINITY; mx[j] = -INFINITY; }
        dmax = 0.0f;
        vmin = (double)INFINITY;
        bad = false;
        i = 0;
        y_first = y0 - R;
        n_in = (int)(y_end - y0) + 2 * R;
        // the shift: the cell at the centre of the wave tile (any finite value works; a nearby one keeps |d| small)
        const long yc = y0 + (y_end - y0) / 2, xc = xw + 32 < g.cols ? xw + 32 : g.cols - 1;
        const float c0 = g.in[yc * g.ld_in + xc];
        cf = isfinite(c0) ? c0 : 0.0f;
        cd = (double)cf;
        kmin = 0; kmax = 2 * R; n_full = C::NTAPS;
        if (EDGE) {
            kmin = x < R ? (int)(R - x) : 0;
            kmax = x + R > g.cols - 1 ? (int)(g.cols - 1 - x + R) : 2 * R;
            n_full = count(0, -(long)R, (long)R + 1);
        }
#pragma unroll
        for (int s = 0; s < PFN; ++s) load_row(s, pf_own[s], pf_halo[s]);
    }

    // in-raster cells under the window centred on (yo, x), rows [y_lo, y_hi)
    __device__ __forceinline__ int count(long yo, long y_lo, long y_hi) const {
        int n = 0;
        for (int dy = -R; dy <= R; ++dy) {
            const long yr = yo + dy;
            if (yr < y_lo || yr >= y_hi) continue;
            const int h = Shape::hw(R, dy < 0 ? -dy : dy);
            const long a = x - h < 0 ? 0 : x - h, b = x + h > g.cols - 1 ? g.cols - 1 : x + h;
            n += b >= a ? (int)(b - a + 1) : 0;
        }
        return n;
    }

    template <int PHASE>
    __device__ __forceinline__ void step() {
        if (i < n_in) step_body<PHASE>();
        ++i;
    }

    template <int PHASE>
    __device__ __forceinline__ void step_body() {
        constexpr int SLOT = C::ROT ? PHASE % PFN : 0;
        const float q = pf_own[SLOT], hq = pf_halo[SLOT];
        if (!C::ROT) {
#pragma unroll
            for (int s = 0; s + 1 < PFN; ++s) { pf_own[s] = pf_own[s + 1]; pf_halo[s] = pf_halo[s + 1]; }
        }
        constexpr int REFILL = C::ROT ? SLOT : PFN - 1;
        if (EDGE || i + PFN < n_in) load_row(i + PFN, pf_own[REFILL], pf_halo[REFILL]);

        const long yy = y_first + i;
        const bool row_in = !EDGE || (yy >= -(long)g.halo_top && yy < g.rows + g.halo_bot);      // wave-uniform
        if (row_in) {
            char *buf = lds;       // ONE row buffer: LDS serves a wave's instructions in order, so the next row's writes
                                   // (issued after this row's reads) cannot overtake them
            double *rowd = reinterpret_cast<double *>(buf);
            float *rowf = reinterpret_cast<float *>(buf + C::STG * 8);
            if (MM) rowf[lane] = q;
            if (MOM) rowd[lane] = (double)q - cd;
            if (lane < 2 * R) {
                if (MM) rowf[64 + lane] = hq;
                if (MOM) rowd[64 + lane] = (double)hq - cd;
            }
            if (MOM) dmax = amax3(dmax, q - cf, hq - cf);
            else bad |= __builtin_isunordered(q, hq);        // a NaN cell: nanmin / nanmax of an all-NaN window is NaN
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                 // (LDS serves one wave's instructions in order)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- extrema, then float64 moments, over centred runs from the centre outwards (two passes over the row
            // buffer: the 2R+1 raw cells and the 2R+1 shifted float64 cells are not live at the same time).  EDGE: the
            // cells left of column 0 / right of the last column (k outside [kmin, kmax]) are not part of any window: NaN
            // for min / max (skipped), 0 for the moments.  A NaN INSIDE the raster poisons S and sends the tile to the
            // exact walker.
            const float qnan = nan_f32();
            if (MM) {
                float lo, hi;
                float v[K];
#pragma unroll
                for (int k = 0; k < K; ++k) v[k] = rowf[lane + k];
                const bool in_c = !EDGE || (R >= kmin && R <= kmax);
                lo = hi = in_c ? v[R] : qnan;
#pragma unroll
                for (int h = 0; h <= R; ++h) {
                    if (h > 0) {
                        const bool in_a = !EDGE || R - h >= kmin, in_b = !EDGE || R + h <= kmax;
                        const float va = in_a ? v[R - h] : qnan, vb = in_b ? v[R + h] : qnan;
                        lo = raw_min3(lo, va, vb);
                        hi = raw_max3(hi, va, vb);
                    }
                    if (!C::level_used(h)) continue;
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const int dy = j - R;
                        if (Shape::hw(R, dy < 0 ? -dy : dy) != h) continue;
                        const int idx = ((PHASE - dy) % K + K) % K;
                        mn[idx] = raw_min(mn[idx], lo);
                        mx[idx] = raw_max(mx[idx], hi);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MOM) {
                double dv[K];
#pragma unroll
                for (int k = 0; k < K; ++k) dv[k] = rowd[lane + k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                double S, Q;
                {
                    const bool in_c = !EDGE || (R >= kmin && R <= kmax);
                    const double d = in_c ? dv[R] : 0.0;
                    S = d; Q = d * d;
                }
                (void)qnan;
#pragma unroll
                for (int h = 0; h <= R; ++h) {
                    if (h > 0) {
                        const bool in_a = !EDGE || R - h >= kmin, in_b = !EDGE || R + h <= kmax;
                        const double a = in_a ? dv[R - h] : 0.0;
                        const double b = in_b ? dv[R + h] : 0.0;
                        S += a + b;
                        Q = fma(a, a, fma(b, b, Q));
                    }
                    if (!C::level_used(h)) continue;
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const int dy = j - R;
                        if (Shape::hw(R, dy < 0 ? -dy : dy) != h) continue;
                        const int idx = ((PHASE - dy) % K + K) % K;
                        sd[idx] += S;
                        sq[idx] += Q;
                    }
                }
            }
            if (!MOM) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- the output row R rows up is complete
        constexpr int DONE = ((PHASE - R) % K + K) % K;
        constexpr int DM = MOM ? DONE : 0, DX = MM ? DONE : 0;
        if (i >= 2 * R && (!EDGE || x < g.cols)) emit(y0 + (i - 2 * R), sd[DM], sq[DM], mn[DX], mx[DX]);
        if (MOM) { sd[DM] = 0.0; sq[DM] = 0.0; }
        if (MM) { mn[DX] = INFINITY; mx[DX] = -INFINITY; }
    }

    __device__ __forceinline__ void emit(long yo, double S, double Q, float lo, float hi) {
        const long off = yo * g.ld_out + x;
        if (MM) {
            if (o.max) st_stream(&o.max[off], hi);
            if (o.min) st_stream(&o.min[off], lo);
            if (o.range) st_stream(&o.range[off], hi - lo);
        }
        if (!MOM) return;
        int n = C::NTAPS;
        if (EDGE) {
            const bool rows_in = yo - R >= -(long)g.halo_top && yo + R < g.rows + g.halo_bot;    // wave-uniform
            n = rows_in ? n_full : count(yo, -(long)g.halo_top, g.rows + g.halo_bot);
        }
        bad |= !isfinite(S) || !isfinite(Q);
        const double dn = (double)n;
        const double inv = EDGE ? walk_rcp(n) : 1.0 / (double)C::NTAPS;
        const double ms = S * inv;
        const double mean = cd + ms;
        const double v0 = (Q - S * ms) * inv;                // the one-pass variance before clamping
        const double var = v0 > 0.0 ? v0 : 0.0;
        // cancellation guard, as in circle_walk.h, evaluated once per tile in run(): the rounding noise of Q and S^2/n is
        // ~ n * eps * max(d^2), so the smallest variance of the tile must stand clear of 1e-9 * max(d^2)
        vmin = v0 < vmin ? v0 : vmin;
        if (o.mean) st_stream(&o.mean[off], (float)mean);
        if (o.var) st_stream(&o.var[off], (float)var);
        if (o.std) {
            // float32 square root of the float64 variance (1 ulp of the float32 result); variances below the float32
            // range are scaled first
            const bool tiny = var < 0x1p-100;
            st_stream(&o.std[off], sqrtf((float)(tiny ? var * 0x1p+200 : var)) * (tiny ? 0x1p-100f : 1.0f));
        }
        if (o.sum) st_stream(&o.sum[off], (float)fma(dn, cd, S));
    }

    template <int... P>
    __device__ __forceinline__ void round(std::integer_sequence<int, P...>) {
        (step<P>(), ...);
        // the round started at a row i0 with ring slot (j - i0) mod K for output row j; the next one starts at i0 + U
        if (MOM) { ring_rotate<MOM ? K : 1, MOM ? C::U : 1>(sd); ring_rotate<MOM ? K : 1, MOM ? C::U : 1>(sq); }
        if (MM) { ring_rotate<MM ? K : 1, MM ? C::U : 1>(mn); ring_rotate<MM ? K : 1, MM ? C::U : 1>(mx); }
    }

    // 0: every result of the tile is good; 1: the exact walker redoes the moments; 2: it redoes everything
    __device__ __forceinline__ int run() {
        init();
        while (i < n_in) {
            round(std::make_integer_sequence<int, C::U>{});
            if (__any(bad)) return 2;
        }
        if (!MOM || !(o.var || o.std)) return 0;
        float am = dmax;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) am = fmaxf(am, __shfl_xor(am, sft));
        const bool redo = !(vmin >= 1e-9 * ((double)am * (double)am));      // flat / ill-conditioned window: the exact walker
        return __any(redo) ? 1 : 0;                                          // redoes the tile's moments
    }
};

template <int R>
__global__ void __launch_bounds__(256, XRS_WALK2_WAVES) XRS_WALK_KERNEL(const WalkGeom g, const WalkOuts o) {
    using C = Walk2Cfg<R, XRS_WALK_SHAPE>;
    __shared__ __attribute__((aligned(16))) char lds_rows[4][C::STG * 12];
    const long t = xcd_tile(blockIdx.x, g.n_tiles, XCD_UNIT(XRS_XCD_WALK, g.tiles_x));
    if (t < 0) return;
    const long ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long xw = tx * 256 + wv * 64;
    const long y0 = ty * W2TH;
    if (xw >= g.cols) return;
    const long y_end = y0 + W2TH < g.rows ? y0 + W2TH : g.rows;
    const bool interior = xw - R >= 0 && xw + 64 + R <= g.cols && y0 - R >= -(long)g.halo_top &&
                          y_end + R <= g.rows + g.halo_bot;
    int rc;
    if (interior) {
        Walk2<R, XRS_WALK_SHAPE, false> w(g, o, lds_rows[wv], xw, y0, y_end, lane);
        rc = w.run();
    } else {
        Walk2<R, XRS_WALK_SHAPE, true> w(g, o, lds_rows[wv], xw, y0, y_end, lane);
        rc = w.run();
    }
    if (rc == 0) return;
    // a non-finite cell under one of the tile's windows (2), or a flat / ill-conditioned window (1): the exact
    // NaN-skipping walkers (moments; for 2 also the float32 statistics with the reference's sequential sum)
    if (XRS_WALK2_MOM && (o.mean || o.var || o.std))
        walk_columns<R, XRS_WALK_SHAPE, false, false, false, true, true>(g, o, xw, lane, y0, y_end);
    if (rc == 2 && (o.sum || o.max || o.min || o.range))
        walk_columns<R, XRS_WALK_SHAPE, true, XRS_WALK2_MOM != 0, XRS_WALK2_MM != 0, false, false>(g, o, xw, lane, y0, y_end);
}

template <int R>
int launch2(WalkGeom &g, const WalkOuts &o, const double *kernel, hipStream_t s) {
    if (!is_shape<R, XRS_WALK_SHAPE>(kernel)) return -1;
    g.tiles_x = (g.cols + 255) / 256;
    g.n_tiles = g.tiles_x * ((g.rows + W2TH - 1) / W2TH);
    const long grid = xcd_grid(g.n_tiles, XCD_UNIT(XRS_XCD_WALK, g.tiles_x));
    if (grid > 0x7fffffffL) return fail("focal statistics: raster too large for one launch");
    hipLaunchKernelGGL((XRS_WALK_KERNEL<R>), dim3((unsigned)grid), dim3(256), 0, s, g, o);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace xrs {

// 0 = launched, -1 = not this shape / a radius this file is instantiated for, > 0 = error.  Null outputs are skipped.
int XRS_WALK_ENTRY(const float *in, float *out_sum, float *out_max, float *out_min, float *out_range, float *out_mean,
                   float *out_var, float *out_std, long rows, long cols, long ld_in, long ld_out, const double *kernel,
                   int krows, int kcols, int halo_top, int halo_bot, hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    if (!out_sum && !out_max && !out_min && !out_range && !out_mean && !out_var && !out_std) return 0;
    WalkGeom g;
    memset(&g, 0, sizeof(g));
    g.in = in; g.rows = rows; g.cols = cols; g.ld_in = ld_in; g.ld_out = ld_out;
    g.halo_top = halo_top; g.halo_bot = halo_bot;
    if ((!XRS_WALK2_MM && (out_max || out_min || out_range)) || (!XRS_WALK2_MOM && (out_sum || out_mean || out_var || out_std)))
        return fail("focal statistics: walker variant without the requested pass");
    const WalkOuts o = {out_sum, out_max, out_min, out_range, out_mean, out_var, out_std};
    switch (krows / 2) {
#ifndef XRS_WALK2_PROBE
        case 4: return launch2<4>(g, o, kernel, s);
        case 5: return launch2<5>(g, o, kernel, s);
        case 6: return launch2<6>(g, o, kernel, s);
        case 7: return launch2<7>(g, o, kernel, s);
        case 8: return launch2<8>(g, o, kernel, s);
        case 9: return launch2<9>(g, o, kernel, s);
        case 10: return launch2<10>(g, o, kernel, s);
        case 11: return launch2<11>(g, o, kernel, s);
#endif
        case 12: return launch2<12>(g, o, kernel, s);
        default: return -1;
    }
}

}  // namespace xrs
