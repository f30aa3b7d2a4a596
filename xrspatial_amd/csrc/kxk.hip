// k x k window kernels: convolve_2d, focal statistics (focal.apply / focal_stats), focal.mean 3x3.
//
// Reference runners replaced:
//   _convolve_2d_numpy  xrspatial/convolution.py:285-313
//   _apply_numpy + _calc_{mean,max,min,range,std,var,sum}  xrspatial/focal.py:305-326, 268-302
//   _mean_numpy         xrspatial/focal.py:44-67
//
// Kernel shape: a 256-thread workgroup produces a TH x 256 output tile.  The input
// region (TH + k-1 rows, 256 + 2*roundup4(k//2) columns) is fetched ONCE from HBM with
// row-coalesced 16-byte loads into an LDS tile; cells outside the raster (or outside
// the shard's halo) are written to LDS as NaN, which gives both edge rules for free:
// convolve_2d propagates NaN (= the reference's NaN border), focal statistics skip NaN
// (= the reference's window clipped to the raster).  Each lane then owns 4 adjacent
// output columns and walks the window with a sliding 4-register view of every LDS row,
// so one ds_read feeds four taps.  Accumulators follow the reference's CPU arithmetic:
// float64 for mean / var / std / convolution (Numba nanmean / nanvar; `num = 0.0`),
// float32 row-major for `sum` (Numba nansum keeps the array dtype).
// No MFMA: 0/1 masks with NaN-skipping are not a dense contraction.
#include "strip.h"

#include <cmath>
#include <cstdlib>

using namespace xrs;

namespace {

constexpr int TW = 256;          // output tile width = 64 lanes x 4 columns
constexpr int TH_FAST = 16;      // output tile height of the compile-time-shape kernels (4 rows per wave)
constexpr int MAX_K = 63;        // one uint64 bit-row per kernel row

struct KxkArgs {
    const float *in;
    float *out[XRS_NUM_STATS];    // convolve uses out[0]
    long rows, cols, ld_in, ld_out;
    int halo_top, halo_bot;
    int krows, kcols;             // runtime sizes (also valid for the compile-time variants)
    int th;                       // output rows per tile (multiple of 4)
    int lpad;                     // roundup4(kcols/2): 16-byte aligned global column where a tile row starts
    int pitch;                    // LDS row pitch in floats = 256 + roundup4(2*(kcols/2))
    int ntaps;                    // number of kernel == 1 taps (focal)
    double inv_ntaps;
    const double *weights;        // device: krows*kcols float64 weights (convolve)
    long tiles_x, n_tiles;
    unsigned long long mask_rows[MAX_K];   // bit kx of entry ky: tap (ky, kx) has kernel == 1 (focal)
};

// LDS tile: element (r, c) holds raster cell (Y0 - ry + r, X0 - rx + c), so the window of the lane
// that owns output columns X0+4*lane .. +3 starts at the 16-byte aligned LDS column 4*lane and is read
// with ds_read_b128.  The shift by rx happens once, at the LDS write.  Cells outside the raster (or
// outside the shard's halo rows) are stored as NaN.  Returns true if this thread stored any
// non-finite value (so the workgroup can pick the NaN-free fast path).
__device__ __forceinline__ bool put4(float *trow, int c0, int pitch, const float4 v) {
    // c0 = LDS column of v.x (may be -3..-1 at the left edge; columns >= pitch are dropped)
    if ((c0 & 3) == 0) {
        if (c0 >= 0 && c0 + 3 < pitch) *reinterpret_cast<float4 *>(trow + c0) = v;
    } else if ((c0 & 1) == 0) {
        if (c0 >= 0 && c0 + 1 < pitch) *reinterpret_cast<float2 *>(trow + c0) = make_float2(v.x, v.y);
        if (c0 + 2 >= 0 && c0 + 3 < pitch) *reinterpret_cast<float2 *>(trow + c0 + 2) = make_float2(v.z, v.w);
    } else {
        if (c0 >= 0 && c0 < pitch) trow[c0] = v.x;
        if (c0 + 1 >= 0 && c0 + 1 < pitch) trow[c0 + 1] = v.y;
        if (c0 + 2 >= 0 && c0 + 2 < pitch) trow[c0 + 2] = v.z;
        if (c0 + 3 >= 0 && c0 + 3 < pitch) trow[c0 + 3] = v.w;
    }
    return !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
}

template <bool VEC>
__device__ __forceinline__ bool load_tile(const KxkArgs &a, float *tile, long X0, long Y0) {
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    const int ry = a.krows / 2, rx = a.kcols / 2;
    const int trows = a.th + a.krows - 1;
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const float qnan = nan_f32();
    bool bad = false;
    if (VEC) {
        // wave wy stages tile rows wy, wy+4, ...: one full-wave float4 load per row + a short tail
        const int nj = (TW + 2 * a.lpad) >> 2;
        const int sh = a.lpad - rx;
        for (int r = wy; r < trows; r += 4) {
            const long y = Y0 - ry + r;
            const bool yok = y >= y_lo && y < y_hi;
            const float *grow = a.in + y * a.ld_in + (X0 - a.lpad);
            float *trow = tile + r * a.pitch;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = lane + 64 * jj;
                if (j >= nj) break;
                const long x = X0 - a.lpad + 4L * j;
                float4 v = make_float4(qnan, qnan, qnan, qnan);
                if (yok && x >= 0 && x < a.cols) v = *reinterpret_cast<const float4 *>(grow + 4 * j);
                bad |= put4(trow, 4 * j - sh, a.pitch, v);
            }
        }
    } else {
        const int total = trows * a.pitch;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int r = i / a.pitch, c = i - r * a.pitch;
            const long y = Y0 - ry + r;
            const long x = X0 - rx + c;
            float v = qnan;
            if (y >= y_lo && y < y_hi && x >= 0 && x < a.cols) v = a.in[y * a.ld_in + x];
            tile[i] = v;
            bad |= !isfinite(v);
        }
    }
    return bad;
}

template <bool VEC>
__device__ __forceinline__ void store_row(float *out, long ld, long y, long x0, long cols, const float (&v)[4]) {
    if (!out) return;
    float *p = out + y * ld + x0;
    if (VEC) {
        stg_stream(reinterpret_cast<float4 *>(p), make_float4(v[0], v[1], v[2], v[3]));
    } else {
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (x0 + o < cols) p[o] = v[o];
    }
}

// Visit every selected tap of the kernel for the 4 adjacent outputs a lane owns:
//   f(ky, kx, v0, v1, v2, v3)  with v_o = input cell under tap (ky, kx) of output column o.
// `rowmask` (bit kx of entry ky selects tap (ky, kx); wave-uniform) or nullptr for "every tap" (convolution).
// KH/KW > 0: compile-time shape, fully unrolled, register-indexed.  0: runtime shape, an 8-register sliding
// view advanced one 16-byte slot per 4 taps; a slot whose 4 mask bits are all set runs without per-tap
// tests, an empty one is skipped (the slot after the last needed one is read and ignored; the tile
// allocation carries 64 bytes of slack for it).
__device__ __forceinline__ void keep4(float4 &q) {
    asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w));
}

template <int KH, int KW, typename F>
__device__ __forceinline__ void walk_window(const KxkArgs &a, const float *tile, int orow, int lane,
                                            const unsigned long long *rowmask, F &&f) {
    if constexpr (KH > 0) {
        constexpr int NS = (4 + 2 * (KW / 2) + 3) / 4;     // 16-byte slots per lane per row
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
            const float4 *r4 = reinterpret_cast<const float4 *>(tile + (orow + ky) * a.pitch) + lane;
            const unsigned long long bits = rowmask ? rowmask[ky] : ~0ull;
            float w[4 * NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const float4 q = r4[i];
                w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
            }
#pragma unroll
            for (int kx = 0; kx < KW; ++kx)
                if (bits >> kx & 1ull) f(ky, kx, w[kx], w[kx + 1], w[kx + 2], w[kx + 3]);
        }
    } else {
        const int kh = a.krows, kw = a.kcols;
        const int nchunks = (kw + 3) >> 2;
        for (int ky = 0; ky < kh; ++ky) {
            unsigned long long bits = rowmask ? rowmask[ky] : ~0ull;
            if (kw < 64) bits &= (1ull << kw) - 1;
            if (!bits) continue;
            const float4 *r4 = reinterpret_cast<const float4 *>(tile + (orow + ky) * a.pitch) + lane;
            // slots are fetched four at a time (ds_read_b128 x4 in flight) and consumed from registers
            // Every slot is loaded whole: `keep4` makes all four components live, otherwise the compiler narrows the
            // 16-byte reads to the dwords a path happens to use (ds_read_b96 / b64 / read2_b32), and at a lane
            // stride of 16 bytes those are 2- to 4-way bank conflicts where ds_read_b128 has none (PMC on the 25x25
            // float32 walk: 65 % of the LDS cycles were conflicts, LDS 93 % busy).
            float4 slot[5];
            slot[0] = r4[0];
            keep4(slot[0]);
            for (int jg = 0; jg < nchunks; jg += 4) {
#pragma unroll
                for (int c = 0; c < 4; ++c) slot[c + 1] = r4[jg + c + 1];      // (reads past the row's last slot land in
#pragma unroll                                                                //  the next row / the allocation's slack)
                for (int c = 0; c < 4; ++c) keep4(slot[c + 1]);                // (after all four are in flight)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int jc = jg + c;
                    const unsigned b4 = jc < nchunks ? (unsigned)(bits >> (4 * jc)) & 15u : 0u;
                    const float w[8] = {slot[c].x, slot[c].y, slot[c].z, slot[c].w,
                                        slot[c + 1].x, slot[c + 1].y, slot[c + 1].z, slot[c + 1].w};
                    if (b4 == 15u) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) f(ky, jc * 4 + jj, w[jj], w[jj + 1], w[jj + 2], w[jj + 3]);
                    } else if (b4) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            if (b4 >> jj & 1u) f(ky, jc * 4 + jj, w[jj], w[jj + 1], w[jj + 2], w[jj + 3]);
                    }
                }
                slot[0] = slot[4];
            }
        }
    }
}

// ------------------------------------------------------------------ focal statistics
// All statistics / any kernel shape.  Per output row: pass 1 walks the window row-major (the order the
// reference's reducers visit their scratch array), pass 2 (std / var only) accumulates squared deviations.
// MODE 0: mean only.  1: every requested statistic.  2: the float32 statistics only (sum, max, min, range) --
// used for large masks whose mean / var / std come from the prefix-sum kernel of kxk_runs.hip; 3: of those only
// the row-major float32 sum; 4: only max / min / range.
template <int KH, int KW, int MODE, bool VEC>
__device__ __forceinline__ void focal_rows_general(const KxkArgs &a, const float *tile, long X0, long Y0,
                                                   bool nan_free = false) {
    constexpr bool MEAN_ONLY = MODE == 0;
    constexpr bool F32_ONLY = MODE >= 2;                 // no float64 statistics
    constexpr bool WANT_SUM = MODE != 4, WANT_MM = MODE != 3;
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const long x0 = X0 + lane * 4;
    if (x0 >= a.cols) return;
    const int rpw = a.th >> 2;                       // output rows per wave

    for (int rr = 0; rr < rpw; ++rr) {
        const int orow = wy * rpw + rr;
        const long y = Y0 + orow;
        if (y >= a.rows) break;

        double sum64[4] = {0, 0, 0, 0};
        int cnt[4] = {0, 0, 0, 0};
        float sum32[4] = {0, 0, 0, 0};
        float mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};

        if (F32_ONLY && nan_free) {
            // float32 statistics of a tile without NaN / out-of-raster cells: three VALU ops per tap
            walk_window<KH, KW>(a, tile, orow, lane, a.mask_rows, [&](int, int, float v0, float v1, float v2, float v3) {
                const float v[4] = {v0, v1, v2, v3};
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    if (WANT_SUM) sum32[o] += v[o];
                    if (WANT_MM) {
                        mn[o] = fminf(mn[o], v[o]);
                        mx[o] = fmaxf(mx[o], v[o]);
                    }
                }
            });
#pragma unroll
            for (int o = 0; o < 4; ++o) cnt[o] = a.ntaps;
        } else {
            walk_window<KH, KW>(a, tile, orow, lane, a.mask_rows, [&](int, int, float v0, float v1, float v2, float v3) {
                const float v[4] = {v0, v1, v2, v3};
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const bool ok = !isnan(v[o]);
                    if (!F32_ONLY) sum64[o] += ok ? (double)v[o] : 0.0;
                    cnt[o] += ok ? 1 : 0;
                    if (!MEAN_ONLY) {
                        if (WANT_SUM) sum32[o] = ok ? sum32[o] + v[o] : sum32[o];
                        if (WANT_MM) {
                            mn[o] = fminf(mn[o], v[o]);
                            mx[o] = fmaxf(mx[o], v[o]);
                        }
                    }
                }
            });
        }

        double mean[4];
        float o_tmp[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            mean[o] = div_refined(sum64[o], (double)cnt[o], rcp_count(cnt[o]));
            o_tmp[o] = (float)mean[o];
        }
        if (!F32_ONLY) store_row<VEC>(a.out[XRS_STAT_MEAN], a.ld_out, y, x0, a.cols, o_tmp);
        if (MEAN_ONLY) continue;

        if (a.out[XRS_STAT_MAX]) {
#pragma unroll
            for (int o = 0; o < 4; ++o) o_tmp[o] = cnt[o] ? mx[o] : nan_f32();
            store_row<VEC>(a.out[XRS_STAT_MAX], a.ld_out, y, x0, a.cols, o_tmp);
        }
        if (a.out[XRS_STAT_MIN]) {
#pragma unroll
            for (int o = 0; o < 4; ++o) o_tmp[o] = cnt[o] ? mn[o] : nan_f32();
            store_row<VEC>(a.out[XRS_STAT_MIN], a.ld_out, y, x0, a.cols, o_tmp);
        }
        if (a.out[XRS_STAT_RANGE]) {
#pragma unroll
            for (int o = 0; o < 4; ++o) o_tmp[o] = cnt[o] ? mx[o] - mn[o] : nan_f32();
            store_row<VEC>(a.out[XRS_STAT_RANGE], a.ld_out, y, x0, a.cols, o_tmp);
        }
        if (WANT_SUM) store_row<VEC>(a.out[XRS_STAT_SUM], a.ld_out, y, x0, a.cols, sum32);

        if (!F32_ONLY && (a.out[XRS_STAT_STD] || a.out[XRS_STAT_VAR])) {
            double ssd[4] = {0, 0, 0, 0};
            walk_window<KH, KW>(a, tile, orow, lane, a.mask_rows, [&](int, int, float v0, float v1, float v2, float v3) {
                const float v[4] = {v0, v1, v2, v3};
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const double d = (double)v[o] - mean[o];
                    ssd[o] += isnan(v[o]) ? 0.0 : d * d;
                }
            });
            float o_var[4], o_std[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const double var = ssd[o] * rcp_count(cnt[o]);
                o_var[o] = (float)var;
                o_std[o] = (float)sqrt(var);
            }
            store_row<VEC>(a.out[XRS_STAT_VAR], a.ld_out, y, x0, a.cols, o_var);
            store_row<VEC>(a.out[XRS_STAT_STD], a.ld_out, y, x0, a.cols, o_std);
        }
    }
}

template <int KH, int KW, int MODE, bool VEC>
__global__ void __launch_bounds__(256) focal_stats_kernel(const KxkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const long t = xcd_tile(blockIdx.x, a.n_tiles, XCD_UNIT(XRS_XCD_LDS, a.tiles_x));
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const long X0 = tx * TW, Y0 = ty * a.th;
    const bool bad = load_tile<VEC>(a, tile, X0, Y0);
    const bool nan_free = !__syncthreads_or(bad);           // (also the barrier between staging and the walk)
    focal_rows_general<KH, KW, MODE, VEC>(a, tile, X0, Y0, nan_free);
}

// Mean only, compile-time kernel shape, 16-byte friendly raster: the headline kernel.
// Each wave owns 4 output rows x 256 columns.  If the staged tile holds only finite in-raster cells
// (the overwhelmingly common case) the window walk is inverted: every LDS row is read ONCE
// (ds_read_b128), converted to float64 ONCE, and added into the accumulators of all the output rows
// whose window covers it; the count is the constant number of taps.  Tiles that touch a raster edge
// or hold NaN/inf take the general per-output-row path with NaN skipping and counting.
template <int KH, int KW>
__global__ void __launch_bounds__(256) focal_mean_fast_kernel(const KxkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    constexpr int RPW = TH_FAST / 4, RX = KW / 2, NV = 4 + 2 * RX, NS = (NV + 3) / 4;
    const long t = xcd_tile(blockIdx.x, a.n_tiles, XCD_UNIT(XRS_XCD_LDS, a.tiles_x));
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const long X0 = tx * TW, Y0 = ty * TH_FAST;
    const bool bad = load_tile<true>(a, tile, X0, Y0);
    if (__syncthreads_or(bad)) {
        focal_rows_general<KH, KW, 0, true>(a, tile, X0, Y0);
        return;
    }
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const long x0 = X0 + lane * 4;          // (finite tile => whole tile inside the raster)

    double acc[RPW][4];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[r][o] = 0.0;

#pragma unroll
    for (int ir = 0; ir < RPW + KH - 1; ++ir) {            // tile row wy*RPW + ir
        const float4 *r4 = reinterpret_cast<const float4 *>(tile + (wy * RPW + ir) * a.pitch) + lane;
        double d[4 * NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const float4 q = r4[i];
            d[4 * i] = (double)q.x; d[4 * i + 1] = (double)q.y; d[4 * i + 2] = (double)q.z; d[4 * i + 3] = (double)q.w;
        }
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
            const int orow = ir - ky;                      // compile-time after unrolling
            if (orow < 0 || orow >= RPW) continue;
            const unsigned bits = (unsigned)a.mask_rows[ky];
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                if (bits >> kx & 1u) {                     // wave-uniform
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[orow][o] += d[kx + o];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long y = Y0 + wy * RPW + r;
        float *p = a.out[XRS_STAT_MEAN] + y * a.ld_out + x0;
        stg_stream(reinterpret_cast<float4 *>(p), make_float4((float)(acc[r][0] * a.inv_ntaps), (float)(acc[r][1] * a.inv_ntaps),
                                                     (float)(acc[r][2] * a.inv_ntaps), (float)(acc[r][3] * a.inv_ntaps)));
    }
}

// Mean only, compile-time shape, register-resident variant (no LDS): the 3x3 terrain kernels' strip
// layout applied to a k-wide window.  Each wave owns 256 columns x RB rows; a lane loads, per input
// row, its own float4 plus the RX cells left and right of it (L1/L2 hits: they are the neighbouring
// lanes' float4s), converts each value to float64 ONCE, and adds it into every output row it belongs
// to.  All (RB+KH-1) row loads of a lane are independent and issued up front.
// Interior waves (window entirely inside the raster: wave-uniform) load unconditionally from a scalar
// row base.  The sums and the nodata handling are strip.h's strip_focal_mean: one body for clean strips, strips
// with NaN cells and strips on the raster's edge.
// CMASK: the mask as a compile-time constant (bit ky * KW + kx; 0 = a.mask_rows at run time) -- straight-line tap walk for
// np.ones((3, 3)), as in the fused pass.
template <int KH, int KW, int RB, bool INTERIOR, unsigned CMASK = 0u>
__device__ __forceinline__ void focal_mean_direct_body(const KxkArgs &a, long x_tile, long y0, int lane) {
    constexpr int RX = KW / 2, NV = 4 + 2 * RX, NR = RB + KH - 1;
    const unsigned loff = (unsigned)lane * 4u;
    float v[NR][NV];
    load_strip<KH, KW, RB, INTERIOR>(a, x_tile, y0, lane, v);

    float *out = a.out[XRS_STAT_MEAN] + y0 * a.ld_out + x_tile;      // scalar
    const int nown = INTERIOR ? 4 : (int)(a.cols - (x_tile + loff) < 4 ? a.cols - (x_tile + loff) : 4);
    strip_focal_mean<KH, KW, RB, CMASK, true, false, INTERIOR>(v, a.mask_rows, a.ntaps, a.inv_ntaps, strip_probe_rows(v), [&](int r, const float (&m)[4]) {
        if (!INTERIOR && y0 + r >= a.rows) return;
        store_cols(out + r * a.ld_out + loff, m[0], m[1], m[2], m[3], nown);
    });
}

template <int KH, int KW, int RB, unsigned CMASK = 0u>
#ifndef XRS_LB_MEAN
#define XRS_LB_MEAN 4
#endif
// At most 4 waves per SIMD: the body needs 62-86 VGPRs and would run 5-7, and this streaming kernel is then 3 % slower (same-box
// A/B, profiles/r05/ab_wave_caps.log: 3x3 mean 0.349 -> 0.339 ms, the 152-register round-4 kernel ran 3 and took 0.340).
#ifndef XRS_MEAN_WAVES
#define XRS_MEAN_WAVES 4
#endif
__attribute__((amdgpu_waves_per_eu(1, XRS_MEAN_WAVES)))
__global__ void __launch_bounds__(256, XRS_LB_MEAN) focal_mean_direct_kernel(const KxkArgs a) {
    const long t = xcd_tile(blockIdx.x, a.n_tiles, a.tiles_x);
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int lane = threadIdx.x & 63;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long x_tile = tx * TW;
    const long y0 = ty * (4 * RB) + (long)wy * RB;
    if (y0 >= a.rows) return;
    if (strip_is_interior<KH, KW, RB>(a, x_tile, y0)) {
        focal_mean_direct_body<KH, KW, RB, true, CMASK>(a, x_tile, y0, lane);
        return;
    }
    if (x_tile + lane * 4 >= a.cols) return;
    focal_mean_direct_body<KH, KW, RB, false, CMASK>(a, x_tile, y0, lane);
}

// All seven statistics, compile-time 3x3 / 5x5 shape, register-resident strip (same layout as the mean
// kernel).  Per output row the window is walked row-major straight out of the lane's registers: float64
// sum, float32 row-major sum, min, max in pass 1, squared deviations from the float64 mean in pass 2 (the
// reference's two-pass nanvar).  CAREFUL = the strip touches a raster edge or holds a non-finite cell:
// NaN cells are skipped and counted; otherwise the count is the constant number of taps.
template <int KH, int KW, int RB, bool INTERIOR, bool CAREFUL>
__device__ __forceinline__ void focal_stats_direct_rows(const KxkArgs &a, long x_tile, long y0, int lane,
                                                        const float (&v)[RB + KH - 1][4 + 2 * (KW / 2)]) {
    const unsigned loff = (unsigned)lane * 4u;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        if (!INTERIOR && y0 + r >= a.rows) break;
        double sum64[4] = {0, 0, 0, 0}, ssd[4] = {0, 0, 0, 0}, mean[4];
        int cnt[4] = {0, 0, 0, 0};
        float sum32[4] = {0, 0, 0, 0};
        float mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
            const unsigned bits = (unsigned)a.mask_rows[ky];
#pragma unroll
            for (int kx = 0; kx < KW; ++kx)
                if (bits >> kx & 1u) {
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const float x = v[r + ky][kx + o];
                        if (CAREFUL) {
                            const bool ok = !isnan(x);
                            sum64[o] += ok ? (double)x : 0.0;
                            cnt[o] += ok ? 1 : 0;
                            sum32[o] = ok ? sum32[o] + x : sum32[o];
                        } else {
                            sum64[o] += (double)x;
                            sum32[o] += x;
                        }
                        mn[o] = fminf(mn[o], x);
                        mx[o] = fmaxf(mx[o], x);
                    }
                }
        }
        double inv[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            inv[o] = CAREFUL ? rcp_count(cnt[o]) : a.inv_ntaps;
            mean[o] = div_refined(sum64[o], CAREFUL ? (double)cnt[o] : (double)a.ntaps, inv[o]);
        }
        if (a.out[XRS_STAT_STD] || a.out[XRS_STAT_VAR]) {
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                const unsigned bits = (unsigned)a.mask_rows[ky];
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
                    if (bits >> kx & 1u) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            const float x = v[r + ky][kx + o];
                            const double d = (double)x - mean[o];
                            ssd[o] += (CAREFUL && isnan(x)) ? 0.0 : d * d;
                        }
                    }
            }
        }
        const long off = (y0 + r) * a.ld_out + x_tile + loff;
        const bool none[4] = {CAREFUL && !cnt[0], CAREFUL && !cnt[1], CAREFUL && !cnt[2], CAREFUL && !cnt[3]};
        const float qn = nan_f32();
        const int nown = INTERIOR ? 4 : (int)(a.cols - (x_tile + loff) < 4 ? a.cols - (x_tile + loff) : 4);
        auto put = [&](int stat, float x0, float x1, float x2, float x3) {
            if (a.out[stat]) store_cols(a.out[stat] + off, x0, x1, x2, x3, nown);
        };
        put(XRS_STAT_MEAN, (float)mean[0], (float)mean[1], (float)mean[2], (float)mean[3]);
        put(XRS_STAT_MAX, none[0] ? qn : mx[0], none[1] ? qn : mx[1], none[2] ? qn : mx[2], none[3] ? qn : mx[3]);
        put(XRS_STAT_MIN, none[0] ? qn : mn[0], none[1] ? qn : mn[1], none[2] ? qn : mn[2], none[3] ? qn : mn[3]);
        put(XRS_STAT_RANGE, none[0] ? qn : mx[0] - mn[0], none[1] ? qn : mx[1] - mn[1], none[2] ? qn : mx[2] - mn[2],
            none[3] ? qn : mx[3] - mn[3]);
        put(XRS_STAT_SUM, sum32[0], sum32[1], sum32[2], sum32[3]);
        double var[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) var[o] = ssd[o] * inv[o];
        put(XRS_STAT_VAR, (float)var[0], (float)var[1], (float)var[2], (float)var[3]);
        put(XRS_STAT_STD, (float)sqrt(var[0]), (float)sqrt(var[1]), (float)sqrt(var[2]), (float)sqrt(var[3]));
    }
}

template <int KH, int KW, int RB>
__global__ void __launch_bounds__(256) focal_stats_direct_kernel(const KxkArgs a) {
    constexpr int NV = 4 + 2 * (KW / 2), NR = RB + KH - 1;
    const long t = xcd_tile(blockIdx.x, a.n_tiles, a.tiles_x);
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int lane = threadIdx.x & 63;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long x_tile = tx * TW;
    const long y0 = ty * (4 * RB) + (long)wy * RB;
    if (y0 >= a.rows) return;
    float v[NR][NV];
    if (strip_is_interior<KH, KW, RB>(a, x_tile, y0)) {
        load_strip<KH, KW, RB, true>(a, x_tile, y0, lane, v);
        bool bad = false;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int i = 0; i < NV; ++i) bad |= !isfinite(v[r][i]);
        if (!__any(bad)) focal_stats_direct_rows<KH, KW, RB, true, false>(a, x_tile, y0, lane, v);
        else focal_stats_direct_rows<KH, KW, RB, true, true>(a, x_tile, y0, lane, v);
    } else {
        if (x_tile + lane * 4 >= a.cols) return;
        load_strip<KH, KW, RB, false>(a, x_tile, y0, lane, v);
        focal_stats_direct_rows<KH, KW, RB, false, true>(a, x_tile, y0, lane, v);
    }
}

// convolve_2d, compile-time 3x3 / 5x5 shape, register-resident strip.  NaN (and the out-of-raster NaN
// fill of edge strips) propagates through the float64 multiply-adds by itself: no special cases.
template <int KH, int KW, int RB>
__global__ void __launch_bounds__(256) convolve_direct_kernel(const KxkArgs a) {
    constexpr int NV = 4 + 2 * (KW / 2), NR = RB + KH - 1;
    const long t = xcd_tile(blockIdx.x, a.n_tiles, a.tiles_x);
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int lane = threadIdx.x & 63;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long x_tile = tx * TW;
    const long y0 = ty * (4 * RB) + (long)wy * RB;
    if (y0 >= a.rows) return;
    float v[NR][NV];
    const bool interior = strip_is_interior<KH, KW, RB>(a, x_tile, y0);
    if (interior) {
        load_strip<KH, KW, RB, true>(a, x_tile, y0, lane, v);
    } else {
        if (x_tile + lane * 4 >= a.cols) return;
        load_strip<KH, KW, RB, false>(a, x_tile, y0, lane, v);
    }
    double acc[RB][4];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[r][o] = 0.0;
#pragma unroll
    for (int ir = 0; ir < NR; ++ir) {
        double d[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) d[i] = (double)v[ir][i];
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
            const int orow = ir - ky;
            if (orow < 0 || orow >= RB) continue;
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const double wt = a.weights[ky * KW + kx];           // wave-uniform: scalar load
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[orow][o] += wt * d[kx + o];
            }
        }
    }
    const unsigned loff = (unsigned)lane * 4u;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        if (!interior && y0 + r >= a.rows) break;
        store_cols(a.out[0] + (y0 + r) * a.ld_out + x_tile + loff, (float)acc[r][0], (float)acc[r][1], (float)acc[r][2],
                   (float)acc[r][3], interior ? 4 : (int)(a.cols - (x_tile + loff) < 4 ? a.cols - (x_tile + loff) : 4));
    }
}

// ------------------------------------------------------------------------ convolve_2d
template <int KH, int KW, bool VEC>
__global__ void __launch_bounds__(256) convolve_kernel(const KxkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const long t = xcd_tile(blockIdx.x, a.n_tiles, XCD_UNIT(XRS_XCD_LDS, a.tiles_x));
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const long X0 = tx * TW, Y0 = ty * a.th;
    load_tile<VEC>(a, tile, X0, Y0);
    __syncthreads();

    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const long x0 = X0 + lane * 4;
    if (x0 >= a.cols) return;
    const int rpw = a.th >> 2;
    const int kw = KW ? KW : a.kcols;
    for (int rr = 0; rr < rpw; ++rr) {
        const int orow = wy * rpw + rr;
        const long y = Y0 + orow;
        if (y >= a.rows) break;
        double acc[4] = {0, 0, 0, 0};
        walk_window<KH, KW>(a, tile, orow, lane, nullptr, [&](int ky, int kx, float v0, float v1, float v2, float v3) {
            const double wt = a.weights[ky * kw + kx];     // wave-uniform address: scalar load
            acc[0] += wt * (double)v0;
            acc[1] += wt * (double)v1;
            acc[2] += wt * (double)v2;
            acc[3] += wt * (double)v3;
        });
        const float o[4] = {(float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]};
        store_row<VEC>(a.out[0], a.ld_out, y, x0, a.cols, o);
    }
}

// -------------------------------------------------------------------- focal.mean 3x3
struct Mean3Args {
    const void *in;
    double *out;
    long rows, cols, ld_in, ld_out;
    int halo_top, halo_bot;
    int n_excl;
    int only_nan_excl;            // every exclude value is NaN (or there is none): no per-cell exclude test in the strip path
    int excl_nan;                 // some exclude value is NaN: NaN cells pass through
    double excl[8];
};

template <typename InT>
__global__ void __launch_bounds__(256) focal_mean3_kernel(const Mean3Args a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.rows * a.cols) return;
    const long y = idx / a.cols, x = idx - y * a.cols;
    const InT *in = static_cast<const InT *>(a.in);
    const double c = (double)in[y * a.ld_in + x];
    bool excl = false;
    for (int e = 0; e < a.n_excl; ++e)
        excl = excl || c == a.excl[e] || (isnan(c) && isnan(a.excl[e]));
    double res = c;
    if (!excl) {
        const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
        double s = 0.0;
        int n = 0;
        for (long yy = y - 1; yy <= y + 1; ++yy) {
            if (yy < y_lo || yy >= y_hi) continue;
            for (long xx = x - 1; xx <= x + 1; ++xx) {
                if (xx < 0 || xx >= a.cols) continue;
                const double v = (double)in[yy * a.ld_in + xx];
                if (!isnan(v)) { s += v; ++n; }
            }
        }
        res = s / (double)n;
    }
    st_stream(&a.out[y * a.ld_out + x], res);
}

// Strip version of focal.mean for 16-byte friendly rasters: a wave owns 256 columns x 4 rows, each lane
// keeps its 6 x 6 neighbourhood in registers (float32 or float64 input), interior waves sum the nine
// cells in float64 row-major (the reference's nanmean order) and divide by the constant 9 (by the count of non-NaN
// cells under a window with nodata); a strip that touches a raster edge is done cell by cell with the counting
// body.  Excluded centre values are passed through in both paths.
template <typename InT>
__device__ __forceinline__ void load6(const InT *p, bool has_l, bool has_r, double (&d)[6]);
template <>
__device__ __forceinline__ void load6<float>(const float *p, bool has_l, bool has_r, double (&d)[6]) {
    const xrs_f4u c = load_f4u(p);
    d[1] = c.x; d[2] = c.y; d[3] = c.z; d[4] = c.w;
    d[0] = has_l ? (double)p[-1] : nan("");
    d[5] = has_r ? (double)p[4] : nan("");
}
template <>
__device__ __forceinline__ void load6<double>(const double *p, bool has_l, bool has_r, double (&d)[6]) {
    const xrs_d2u a = reinterpret_cast<const xrs_d2u *>(p)[0], b = reinterpret_cast<const xrs_d2u *>(p)[1];
    d[1] = a.x; d[2] = a.y; d[3] = b.x; d[4] = b.y;
    d[0] = has_l ? p[-1] : nan("");
    d[5] = has_r ? p[4] : nan("");
}

__device__ __forceinline__ bool is_excluded(const Mean3Args &a, double c) {
    bool ex = false;
    for (int e = 0; e < a.n_excl; ++e) ex = ex || c == a.excl[e] || (isnan(c) && isnan(a.excl[e]));
    return ex;
}

// s / 9 correctly rounded without the division sequence (v_div_scale / v_rcp / ~8 fma / v_div_fmas / v_div_fixup): with
// y = RN(1/9) exact, two residual corrections q <- q + (s - 9 q) y (each residual exact in one fma) give the correctly
// rounded quotient (Markstein; the divisor's significand is not all ones).  Finite s only (the caller guarantees it).
// Results that would be subnormal or overflow the scaling-free form take the true division.
__device__ __forceinline__ double div9_exact(double s) {
    const double y = 0x1.c71c71c71c71cp-4;                 // RN(1/9)
    const double as = fabs(s);
    if (__builtin_expect(!(as > 0x1p-900 && as < 0x1p900), 0)) return s / 9.0;
    double q = s * y;
    q = fma(fma(-9.0, q, s), y, q);
    q = fma(fma(-9.0, q, s), y, q);
    return q;
}

#ifndef XRS_LB_MEAN3
#define XRS_LB_MEAN3 3
#endif
// EXCL: some exclude value is not NaN (a nodata VALUE: excludes=[nan, -9999]) -- a centre cell that equals one passes
// through.  Its own instantiation: the test per cell costs the default one (NaN only) 46 spilled registers.
template <typename InT, bool EXCL>
__global__ void __launch_bounds__(256, XRS_LB_MEAN3) focal_mean3_strip_kernel(const Mean3Args a, const long tiles_x, const long n_tiles) {
    constexpr int RB = 4;
    const long t = xcd_tile(blockIdx.x, n_tiles, tiles_x);
    if (t < 0) return;
    const long ty = t / tiles_x, tx = t - ty * tiles_x;
    const int lane = threadIdx.x & 63;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long x_tile = tx * TW, y0 = ty * (4 * RB) + (long)wy * RB;
    if (y0 >= a.rows) return;
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    const InT *in = static_cast<const InT *>(a.in);
    const unsigned loff = (unsigned)lane * 4u;
    const long x0 = x_tile + loff;
    const bool interior = x_tile >= 4 && x_tile + TW + 4 <= a.cols && y0 - 1 >= y_lo && y0 + RB + 1 <= y_hi &&
                          y0 + RB <= a.rows && (EXCL || a.only_nan_excl);
    if (interior) {
        double d[RB + 2][6];
#pragma unroll
        for (int r = 0; r < RB + 2; ++r) load6<InT>(in + (y0 - 1 + r) * a.ld_in + x_tile + loff, true, true, d[r]);
        // Nodata like strip.h's strip_focal_mean: a chain of fused multiply-adds over the lane's 6 cells of a row is non-finite
        // when one of them is NaN / inf (3 instructions; float64 products of float32 or sane float64 cells do not overflow, and a
        // row that does only takes the repair for nothing); the rows are voted on wave-wide.  A clean strip then runs straight
        // through: nine additions and a division by 9 per cell, no test per cell (no cell is NaN, and NaN is the only exclude
        // value this path is taken with).  Otherwise the ROWS that hold one are repaired (NaN -> 0 in the registers: adding 0.0
        // where the reference skips the cell leaves its row-major float64 sum as it is; positions into a per-lane bit mask),
        // and only output rows under a repaired row count their cells and divide by the count.  (Round 4 redid the whole strip
        // cell by cell from re-loaded rows: 0.66 -> 1.00 ms at 0.1 % nodata.)
        unsigned rows_hit = 0u;                   // wave-uniform
#pragma unroll
        for (int r = 0; r < RB + 2; ++r) {
            const double probe = fma(d[r][3], d[r][4], fma(d[r][0], d[r][1], d[r][2])) + d[r][5];
            rows_hit |= __any(!isfinite(probe)) ? 1u << r : 0u;
        }
        // s / 9 for a finite sum of float32 cells: 0 or 2^-149 <= |s| <= 2^132, where div9_exact's corrections need no guard
        auto ninth = [](double s) {
            if (sizeof(InT) == 8) return div9_exact(s);
            const double y = 0x1.c71c71c71c71cp-4;
            double q = s * y;
            q = fma(fma(-9.0, q, s), y, q);
            return fma(fma(-9.0, q, s), y, q);
        };
        if (rows_hit == 0u) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                __builtin_amdgcn_sched_barrier(0);   // (one output row at a time: interleaved, the rows' temporaries spill)
                double res[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    double s = 0.0;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) s += d[r + ky][o + kx];
                    res[o] = ninth(s);
                    // (a nodata VALUE among the excludes -- excludes=[nan, -9999] --: such a centre cell passes through; until
                    // round 6 any non-NaN exclude sent the whole raster down the cell-by-cell body)
                    if (EXCL && is_excluded(a, d[r + 1][o + 1])) res[o] = d[r + 1][o + 1];
                }
                store_wave_row_d4(a.out + (y0 + r) * a.ld_out + x_tile, lane, res[0], res[1], res[2], res[3]);
            }
            return;
        }
        unsigned nanbits[2] = {0u, 0u};           // bit 8 * (r & 3) + i of entry r >> 2: cell i of loaded row r was NaN
#pragma unroll
        for (int r = 0; r < RB + 2; ++r) {
            if (!(rows_hit >> r & 1u)) continue;
            unsigned m = 0u;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool isn = isnan(d[r][i]);
                m |= isn ? 1u << (8 * (r & 3) + i) : 0u;
                d[r][i] = isn ? 0.0 : d[r][i];
            }
            nanbits[r >> 2] |= m;
        }
        // s / n for the n = 9 - lost cells that count, correctly rounded like the reference's division, without the division
        // sequence: Markstein's two residual corrections (div9_exact) with y = RN(1 / n) -- checked against exact rational
        // arithmetic for n = 1 .. 9 on 1.8e6 sums -- and y fetched from a table ACROSS the wave (lane j holds 1 / (9 - j); lane 9:
        // 1 / 0 = inf, and 0 * inf = NaN is the reference's 0 / 0).
        const double my_y = 1.0 / (double)(9 - (lane < 9 ? lane : 9));
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const unsigned hit = (rows_hit >> r) & 7u;
            double res[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                double s = 0.0;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) s += d[r + ky][o + kx];
                if (hit == 0u) {
                    res[o] = ninth(s);
                } else {
                    int lost = 0;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
                        if (hit >> ky & 1u) lost += __popc((nanbits[(r + ky) >> 2] >> (8 * ((r + ky) & 3) + o)) & 7u);
                    const double n = (double)(9 - lost), y = __shfl(my_y, lost);
                    double q = s * y;
                    q = fma(fma(-n, q, s), y, q);
                    q = fma(fma(-n, q, s), y, q);
                    // (an infinite or, from float64 cells, extreme sum: the true division; 0 * inf above is already the NaN of 0 / 0)
                    const double as = fabs(s);
                    if (__builtin_expect(!(as < 0x1p900 && (sizeof(InT) == 4 || as > 0x1p-900 || as == 0.0)) && lost < 9, 0)) q = s / n;
                    const bool c_nan = (hit & 2u) && (nanbits[(r + 1) >> 2] >> (8 * ((r + 1) & 3) + o + 1) & 1u);
                    res[o] = (c_nan && a.excl_nan) ? nan("") : q;
                    if (EXCL && !c_nan && is_excluded(a, d[r + 1][o + 1])) res[o] = d[r + 1][o + 1];
                }
                if (EXCL && hit == 0u && is_excluded(a, d[r + 1][o + 1])) res[o] = d[r + 1][o + 1];
            }
            store_wave_row_d4(a.out + (y0 + r) * a.ld_out + x_tile, lane, res[0], res[1], res[2], res[3]);
        }
        return;
    }
    // edge / NaN strip: the reference's loop, cell by cell (clamped window, NaN skipped, 0/0 -> NaN)
    if (x0 >= a.cols) return;
    for (int r = 0; r < RB; ++r) {
        const long y = y0 + r;
        if (y >= a.rows) break;
        for (int o = 0; o < 4; ++o) {
            const long x = x0 + o;
            if (x >= a.cols) break;
            const double c = (double)in[y * a.ld_in + x];
            double resv = c;
            if (!is_excluded(a, c)) {
                double s = 0.0;
                int n = 0;
                for (long yy = y - 1; yy <= y + 1; ++yy) {
                    if (yy < y_lo || yy >= y_hi) continue;
                    for (long xx = x - 1; xx <= x + 1; ++xx) {
                        if (xx < 0 || xx >= a.cols) continue;
                        const double v = (double)in[yy * a.ld_in + xx];
                        if (!isnan(v)) { s += v; ++n; }
                    }
                }
                resv = s / (double)n;
            }
            st_stream(&a.out[y * a.ld_out + x], resv);
        }
    }
}

// --------------------------------------------------------------------------- host side
int plan_tile(KxkArgs &a, size_t *lds_bytes) {
    const int rx = a.kcols / 2;
    a.lpad = (rx + 3) & ~3;
    a.pitch = TW + ((2 * rx + 3) & ~3);
    for (int th = 16; th >= 4; th >>= 1) {
        const size_t bytes = ((size_t)(th + a.krows - 1) * a.pitch + 32) * sizeof(float);   // + read slack (<= 5 slots past the last row)
        // (CDNA4: 160 KiB of LDS per CU.  Up to 64 KiB several workgroups share a CU; the tiles of 51x51 .. 63x63 windows
        //  take up to 84 KiB = one workgroup per CU, raised per kernel with hipFuncSetAttribute: allow_big_lds)
        if (bytes <= 156 * 1024) {
            a.th = th;
            *lds_bytes = bytes;
            return 0;
        }
    }
    return fail("kernel %dx%d needs more than 156 KiB of LDS per tile", a.krows, a.kcols);
}

int check_common(const char *who, const float *in, long rows, long cols, long ld_in, long ld_out,
                 const double *kernel, int krows, int kcols, int ht, int hb) {
    if (!in) return fail("%s: null input", who);
    if (rows < 0 || cols < 0 || ld_in < cols || ld_out < cols) return fail("%s: bad shape", who);
    if (!kernel || krows <= 0 || kcols <= 0 || !(krows & 1) || !(kcols & 1))
        return fail("%s: kernel must be odd x odd, got %dx%d", who, krows, kcols);
    // (windows beyond MAX_K in either direction: kxk_big.hip -- the callers test for that right after this check)
    if (ht < 0 || hb < 0) return fail("%s: negative halo", who);
    return 0;
}

bool vec_ok(const KxkArgs &a, unsigned out_mask) {
    bool v = (a.cols % 4 == 0) && (a.ld_in % 4 == 0) && (a.ld_out % 4 == 0) && aligned16(a.in);
    for (int i = 0; i < XRS_NUM_STATS; ++i)
        if ((out_mask >> i & 1) && a.out[i]) v = v && aligned16(a.out[i]);
    return v;
}

// dynamic LDS beyond 64 KiB has to be allowed per kernel and device (the call synchronises, so it is issued once per
// thread, kernel instantiation and device: every instantiation of one kernel template has the same pointer TYPE, so the
// cache is keyed on the pointer VALUE, with one bit per device as in zonal.hip)
template <typename K>
int allow_big_lds(K kernel_fn, size_t lds) {
    if (lds <= 64 * 1024) return 0;
    struct Seen { const void *fn; unsigned long long devices; };
    static thread_local Seen seen[8];
    static thread_local int n_seen = 0;
    const void *fn = reinterpret_cast<const void *>(kernel_fn);
    int dev = 0;
    XRS_HIP(hipGetDevice(&dev));
    const unsigned long long bit = dev < 64 ? 1ull << dev : 0ull;          // (device >= 64: never cached)
    Seen *e = nullptr;
    for (int i = 0; i < n_seen; ++i) if (seen[i].fn == fn) e = &seen[i];
    if (e && (e->devices & bit)) return 0;
    XRS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    if (!e) {
        e = &seen[n_seen < 8 ? n_seen++ : 7];                               // (table full: the last entry is recycled)
        e->fn = fn; e->devices = 0;
    }
    e->devices |= bit;
    return 0;
}

template <int KH, int KW, int MODE>
int launch_focal(const KxkArgs &a, bool vec, size_t lds, hipStream_t s) {
    const unsigned grid = (unsigned)xcd_grid(a.n_tiles, XCD_UNIT(XRS_XCD_LDS, a.tiles_x));
    if (int rc = vec ? allow_big_lds(&focal_stats_kernel<KH, KW, MODE, true>, lds) : allow_big_lds(&focal_stats_kernel<KH, KW, MODE, false>, lds)) return rc;
    if (vec)
        hipLaunchKernelGGL((focal_stats_kernel<KH, KW, MODE, true>), dim3(grid), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((focal_stats_kernel<KH, KW, MODE, false>), dim3(grid), dim3(256), lds, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <int KH, int KW>
int launch_mean_fast(const KxkArgs &a, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((focal_mean_fast_kernel<KH, KW>), dim3((unsigned)xcd_grid(a.n_tiles, XCD_UNIT(XRS_XCD_LDS, a.tiles_x))), dim3(256), lds, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <int KH, int KW, int RB>
int launch_mean_direct_rb(KxkArgs a, hipStream_t s) {
    a.n_tiles = a.tiles_x * ((a.rows + 4 * RB - 1) / (4 * RB));
    // np.ones((3, 3)) as a compile-time mask: 0.41 -> 0.37 ms.  (The same for the 5x5 masks made the stand-alone kernel
    // spill at its 128-register budget -- 0.42 -> 1.80 ms -- so they keep the run-time mask; profiles/r01/cmask2_ab_r01.log.)
    constexpr bool SPECIALISE = KH == 3 && KW == 3 && RB == 4;
    constexpr unsigned BOX = SPECIALISE ? (1u << (KH * KW)) - 1u : 0u;
    unsigned mask = 0;
    for (int ky = 0; ky < KH; ++ky) mask |= (unsigned)a.mask_rows[ky] << (ky * KW);
    const dim3 grid((unsigned)xcd_grid(a.n_tiles, a.tiles_x));
    if (SPECIALISE && mask == BOX)
        hipLaunchKernelGGL((focal_mean_direct_kernel<KH, KW, RB, BOX>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((focal_mean_direct_kernel<KH, KW, RB>), grid, dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <int KH, int KW>
int launch_mean_direct(const KxkArgs &a, hipStream_t s) {
    const char *e = ab_env("XRS_FOCAL_RB");          // A/B knob: rows per wave (default 4)
    if (e && e[0] == '2') return launch_mean_direct_rb<KH, KW, 2>(a, s);
    return launch_mean_direct_rb<KH, KW, 4>(a, s);
}

template <int KH, int KW>
int launch_stats_direct(KxkArgs a, hipStream_t s) {
    constexpr int RB = KH >= 5 ? 1 : 2;          // few rows per wave: the 7 output streams dominate traffic, and
                                                 // registers (12 per output + the window) set the occupancy
    a.n_tiles = a.tiles_x * ((a.rows + 4 * RB - 1) / (4 * RB));
    hipLaunchKernelGGL((focal_stats_direct_kernel<KH, KW, RB>), dim3((unsigned)xcd_grid(a.n_tiles, a.tiles_x)), dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <int KH, int KW>
int launch_convolve_direct(KxkArgs a, hipStream_t s) {
    constexpr int RB = 4;
    a.n_tiles = a.tiles_x * ((a.rows + 4 * RB - 1) / (4 * RB));
    hipLaunchKernelGGL((convolve_direct_kernel<KH, KW, RB>), dim3((unsigned)xcd_grid(a.n_tiles, a.tiles_x)), dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

// XRS_FOCAL_VARIANT=lds forces the LDS-tile kernels for small masks (A/B measurements; default: direct)
bool prefer_lds() {
    const char *e = ab_env("XRS_FOCAL_VARIANT");
    return e && strcmp(e, "lds") == 0;
}

// Column walkers (kxk_circle*.hip, kxk_box*.hip) for masks that are circles or boxes: -1 if neither.
int try_walk_f32(const float *in, float *const *out, bool with_moments, long rows, long cols, long ld_in, long ld_out,
                 const double *kernel, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s) {
    float *mean = with_moments ? out[XRS_STAT_MEAN] : nullptr, *var = with_moments ? out[XRS_STAT_VAR] : nullptr,
          *sd = with_moments ? out[XRS_STAT_STD] : nullptr;
    int rc = try_launch_focal_circle_f32(in, out[XRS_STAT_SUM], out[XRS_STAT_MAX], out[XRS_STAT_MIN], out[XRS_STAT_RANGE],
                                         mean, var, sd, rows, cols, ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s);
    if (rc < 0)
        rc = try_launch_focal_box_f32(in, out[XRS_STAT_SUM], out[XRS_STAT_MAX], out[XRS_STAT_MIN], out[XRS_STAT_RANGE], mean,
                                      var, sd, rows, cols, ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s);
    return rc;
}
int try_walk_f64(const float *in, float *mean, float *var, float *sd, long rows, long cols, long ld_in, long ld_out,
                 const double *kernel, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s) {
    int rc = try_launch_focal_circle_f64(in, mean, var, sd, rows, cols, ld_in, ld_out, kernel, krows, kcols, halo_top,
                                         halo_bot, s);
    if (rc < 0)
        rc = try_launch_focal_box_f64(in, mean, var, sd, rows, cols, ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s);
    return rc;
}

// XRS_FOCAL_VARIANT=strip keeps small circular masks on the register-strip all-statistics kernel (A/B; default: walker)
bool prefer_strip() {
    const char *e = ab_env("XRS_FOCAL_VARIANT");
    return e && (strcmp(e, "strip") == 0 || strcmp(e, "lds") == 0);
}

template <bool MEAN_ONLY>
int dispatch_focal(const KxkArgs &a, bool vec, size_t lds, hipStream_t s) {
    // (the register-strip kernels take any width / pitch / base address; `vec` only gates the LDS-tile kernels)
    if (MEAN_ONLY && !prefer_lds()) {
        if (a.krows == 3 && a.kcols == 3) return launch_mean_direct<3, 3>(a, s);
        if (a.krows == 5 && a.kcols == 5) return launch_mean_direct<5, 5>(a, s);
        // (7x7 would need 256 VGPRs in registers: it stays on the LDS-tile kernel)
    }
    if (MEAN_ONLY && vec && a.th == TH_FAST) {
        if (a.krows == 3 && a.kcols == 3) return launch_mean_fast<3, 3>(a, lds, s);
        if (a.krows == 5 && a.kcols == 5) return launch_mean_fast<5, 5>(a, lds, s);
        if (a.krows == 7 && a.kcols == 7) return launch_mean_fast<7, 7>(a, lds, s);
    }
    if (!MEAN_ONLY && !prefer_lds()) {
        if (a.krows == 3 && a.kcols == 3) return launch_stats_direct<3, 3>(a, s);
        if (a.krows == 5 && a.kcols == 5) return launch_stats_direct<5, 5>(a, s);
    }
    // (the unrolled all-statistics bodies need > 170 VGPRs beyond 3x3; the runtime walk needs ~72)
    if (a.krows == 3 && a.kcols == 3) return launch_focal<3, 3, MEAN_ONLY ? 0 : 1>(a, vec, lds, s);
    return launch_focal<0, 0, MEAN_ONLY ? 0 : 1>(a, vec, lds, s);
}

// mean / var / std / sum over annulus_kernel(1, 1, R, RI): the moments walker's translation unit for the outer radius
int launch_mom_annulus(const float *in, float *o_sum, float *o_mean, float *o_var, float *o_std, long rows, long cols, long ld_in,
                       long ld_out, const double *kernel, int krows, int kcols, int ht, int hb, hipStream_t s) {
    if (krows != kcols) return -1;
    switch (krows / 2) {
#define XRS_ANN(RR) case RR: return try_launch_focal_mom_annulus##RR(in, o_sum, o_mean, o_var, o_std, rows, cols, ld_in, ld_out, kernel, krows, kcols, ht, hb, s);
        XRS_ANN(4) XRS_ANN(5) XRS_ANN(6) XRS_ANN(7) XRS_ANN(8) XRS_ANN(9) XRS_ANN(10) XRS_ANN(11) XRS_ANN(12)
#undef XRS_ANN
        default: return -1;
    }
}

// the mean or the uniform-weight convolution over annulus_kernel(1, 1, R, RI): the wide walker's translation unit for the outer radius
int launch_wide_annulus(const float *in, float *o_mean, float *o_conv, long rows, long cols, long ld_in, long ld_out,
                        const double *kernel, const double *weights_dev, int krows, int kcols, int ht, int hb, hipStream_t s) {
    if (krows != kcols) return -1;
    switch (krows / 2) {
#define XRS_ANN(RR) case RR: return try_launch_wide_annulus##RR(in, o_mean, o_conv, rows, cols, ld_in, ld_out, kernel, weights_dev, krows, kcols, ht, hb, s);
        XRS_ANN(4) XRS_ANN(5) XRS_ANN(6) XRS_ANN(7) XRS_ANN(8) XRS_ANN(9) XRS_ANN(10) XRS_ANN(11) XRS_ANN(12)
#undef XRS_ANN
        default: return -1;
    }
}

}  // namespace

extern "C" {

size_t xrs_kxk_workspace_bytes(int krows, int kcols) {
    if (krows <= 0 || kcols <= 0) return 0;
    return (size_t)krows * kcols * sizeof(double);       // float64 weights of convolve2d
}

// the kernel copy (256-byte aligned) + the tile map of the separable box walk (boxsep.hip) + the work-list of the moments
// kernels' slow tiles (mom_impl.h: focal_mom_rescue_kernel)
static size_t weights_span(int krows, int kcols) { return ((size_t)krows * kcols * sizeof(double) + 255) & ~(size_t)255; }
static size_t todo_span(long rows, long cols) { return (box_todo_bytes(rows, cols) + 255) & ~(size_t)255; }
size_t xrs_focal_workspace_bytes(int64_t rows, int64_t cols, int krows, int kcols) {
    if (krows <= 0 || kcols <= 0 || rows < 0 || cols < 0) return 0;
    return weights_span(krows, kcols) + todo_span(rows, cols) + mom_rescue_bytes(rows, cols);
}

int xrs_convolve2d_f32(const float *in_dev, float *out_dev, int64_t rows, int64_t cols, int64_t ld_in,
                       int64_t ld_out, const double *kernel, int krows, int kcols, void *work_dev,
                       int halo_top, int halo_bot, void *stream) {
    if (int rc = check_common("xrs_convolve2d_f32", in_dev, rows, cols, ld_in, ld_out, kernel, krows, kcols,
                              halo_top, halo_bot)) return rc;
    if (!out_dev || !work_dev) return fail("xrs_convolve2d_f32: null output/workspace");
    if (rows == 0 || cols == 0) return 0;
    if (krows > MAX_K || kcols > MAX_K) {
        float *outs1[1] = {out_dev};
        return launch_window_any_size(true, in_dev, outs1, rows, cols, ld_in, ld_out, kernel, krows, kcols, work_dev, halo_top,
                                      halo_bot, as_stream(stream));
    }
    KxkArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in_dev; a.out[0] = out_dev;
    a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out;
    a.halo_top = halo_top; a.halo_bot = halo_bot; a.krows = krows; a.kcols = kcols;
    size_t lds;
    if (int rc = plan_tile(a, &lds)) return rc;
    hipStream_t s = as_stream(stream);
    XRS_HIP(hipMemcpyAsync(work_dev, kernel, (size_t)krows * kcols * sizeof(double), hipMemcpyHostToDevice, s));
    a.weights = static_cast<const double *>(work_dev);
    if (krows >= 7 && !ab_env("XRS_CONV_TAPS")) {
        // one weight value on a circle / box (normalised circle_kernel, np.ones / k^2): the wide row walker (wide_impl.h,
        // float32 on shifted values, guarded).  (Round 1's float64 column walker: experiments/superseded/.)
        int rc = -1;
        rc = try_launch_conv_wide_circle(in_dev, out_dev, rows, cols, ld_in, ld_out, kernel, a.weights, krows, kcols, halo_top, halo_bot, s);
        if (rc < 0)
            rc = try_launch_conv_wide_box(in_dev, out_dev, rows, cols, ld_in, ld_out, kernel, a.weights, krows, kcols, halo_top, halo_bot, s);
        if (rc < 0)             // a normalised annulus_kernel (focal.hotspots' other documented mask)
            rc = launch_wide_annulus(in_dev, nullptr, out_dev, rows, cols, ld_in, ld_out, kernel, a.weights, krows, kcols, halo_top, halo_bot, s);
        if (rc >= 0) return rc;
    }
    a.tiles_x = (cols + TW - 1) / TW;
    a.n_tiles = a.tiles_x * ((rows + a.th - 1) / a.th);
    const unsigned grid = (unsigned)xcd_grid(a.n_tiles, XCD_UNIT(XRS_XCD_LDS, a.tiles_x));
    const bool vec = vec_ok(a, 1u);
#define XRS_CONV(KH, KW)                                                                              \
    do {                                                                                              \
        if (int rc_ = vec ? allow_big_lds(&convolve_kernel<KH, KW, true>, lds) : allow_big_lds(&convolve_kernel<KH, KW, false>, lds)) return rc_; \
        if (vec) hipLaunchKernelGGL((convolve_kernel<KH, KW, true>), dim3(grid), dim3(256), lds, s, a);  \
        else hipLaunchKernelGGL((convolve_kernel<KH, KW, false>), dim3(grid), dim3(256), lds, s, a);     \
    } while (0)
    if (!prefer_lds() && krows == 3 && kcols == 3) return launch_convolve_direct<3, 3>(a, s);
    if (!prefer_lds() && krows == 5 && kcols == 5) return launch_convolve_direct<5, 5>(a, s);
    if (krows == 3 && kcols == 3) XRS_CONV(3, 3);
    else if (krows == 5 && kcols == 5) XRS_CONV(5, 5);
    else XRS_CONV(0, 0);
#undef XRS_CONV
    XRS_LAUNCH_CHECK();
    return 0;
}

int xrs_focal_stats_f32(const float *in_dev, float *const *outs_dev, unsigned stat_mask, int64_t rows,
                        int64_t cols, int64_t ld_in, int64_t ld_out, const double *kernel, int krows,
                        int kcols, void *work_dev, int halo_top, int halo_bot, void *stream) {
    return xrs_focal_stats_f32_ex(in_dev, outs_dev, stat_mask, rows, cols, ld_in, ld_out, kernel, krows, kcols, work_dev,
                                  work_dev ? xrs_kxk_workspace_bytes(krows, kcols) : 0, halo_top, halo_bot, 0u, stream);
}

int xrs_focal_stats_f32_ex(const float *in_dev, float *const *outs_dev, unsigned stat_mask, int64_t rows,
                           int64_t cols, int64_t ld_in, int64_t ld_out, const double *kernel, int krows,
                           int kcols, void *work_dev, size_t work_bytes, int halo_top, int halo_bot, unsigned flags,
                           void *stream) {
    if (int rc = check_common("xrs_focal_stats_f32", in_dev, rows, cols, ld_in, ld_out, kernel, krows, kcols,
                              halo_top, halo_bot)) return rc;
    if (!outs_dev) return fail("xrs_focal_stats_f32: null outputs");
    stat_mask &= (1u << XRS_NUM_STATS) - 1;
    if (!stat_mask) return 0;
    KxkArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in_dev;
    for (int i = 0; i < XRS_NUM_STATS; ++i) {
        if (stat_mask >> i & 1) {
            if (!outs_dev[i]) return fail("xrs_focal_stats_f32: statistic %d selected but its output is NULL", i);
            a.out[i] = outs_dev[i];
        }
    }
    if (rows == 0 || cols == 0) return 0;
    hipStream_t s = as_stream(stream);
    if (krows > MAX_K || kcols > MAX_K) {
        if (!work_dev || work_bytes < xrs_kxk_workspace_bytes(krows, kcols)) return fail("xrs_focal_stats_f32: windows beyond 63x63 need a workspace of xrs_kxk_workspace_bytes()");
        return launch_window_any_size(false, in_dev, a.out, rows, cols, ld_in, ld_out, kernel, krows, kcols, work_dev, halo_top,
                                      halo_bot, s);
    }
    // np.ones((k, k)): scratch for the tile map of the separable walk (boxsep.hip), if the caller brought enough
    const bool have_work = work_dev && work_bytes >= xrs_focal_workspace_bytes(rows, cols, krows, kcols);
    unsigned char *const box_todo = have_work ? static_cast<unsigned char *>(work_dev) + weights_span(krows, kcols) : nullptr;
    // ... and for the work-list of the moments kernels (mom_impl.h); without it their slow tiles are walked in place
    struct RescueScope {
        explicit RescueScope(unsigned *p) { mom_rescue_slot() = p; }
        ~RescueScope() { mom_rescue_slot() = nullptr; }
    } rescue_scope(have_work ? reinterpret_cast<unsigned *>(static_cast<unsigned char *>(work_dev) + weights_span(krows, kcols) + todo_span(rows, cols))
                             : nullptr);
    // Large circles / boxes: the float32 walkers of wide_impl.h / ext_impl.h / mom_impl.h.  XRS_FOCAL_EXACT_MOMENTS keeps
    // the float64 column walkers (mean / var / std within ~1 ulp of the reference's float64 accumulators, ~2x the time);
    // XRS_FOCAL_SEQUENTIAL_SUM keeps `sum` on the kernel that adds the taps in the reference's order in float32 (bit-exact
    // with numba's nansum) instead of rounding the exact sum once.  (`make AB=1`: XRS_FOCAL_GEN=1 selects the first generation
    // for A/B runs; the second lives in experiments/superseded/.)
    if (flags & ~(unsigned)(XRS_FOCAL_EXACT_MOMENTS | XRS_FOCAL_SEQUENTIAL_SUM)) return fail("xrs_focal_stats_f32_ex: unknown flag bits 0x%x", flags);
    const char *gen = ab_env("XRS_FOCAL_GEN");
    const bool gen1 = (flags & XRS_FOCAL_EXACT_MOMENTS) || (gen && gen[0] == '1');
    const bool seq_sum = (flags & XRS_FOCAL_SEQUENTIAL_SUM) != 0;
    const unsigned m_mean = 1u << XRS_STAT_MEAN, m_sum = 1u << XRS_STAT_SUM;
    if (!gen1 && krows == kcols && krows >= 7 && !(stat_mask & ~(m_mean | m_sum)) && !((stat_mask & m_sum) && seq_sum)) {
        // mean and / or sum only: one 16-byte load per lane and row, float32 prefix sums (wide_impl.h)
        int rc = try_launch_focal_wide_circle(in_dev, a.out[XRS_STAT_MEAN], a.out[XRS_STAT_SUM], rows, cols, ld_in, ld_out,
                                              kernel, krows, kcols, halo_top, halo_bot, s);
        if (rc < 0)
            rc = try_launch_focal_wide_box(in_dev, a.out[XRS_STAT_MEAN], a.out[XRS_STAT_SUM], rows, cols, ld_in, ld_out,
                                           kernel, krows, kcols, halo_top, halo_bot, s);
        // annulus_kernel(1, 1, R, RI), the mean alone: the wide walker's annulus instantiations (a row with a hole = two runs)
        if (rc < 0 && stat_mask == m_mean)
            rc = launch_wide_annulus(in_dev, a.out[XRS_STAT_MEAN], nullptr, rows, cols, ld_in, ld_out, kernel, nullptr, krows, kcols,
                                     halo_top, halo_bot, s);
        // ... with the sum: the moments walker with only the mean / sum planes
        if (rc < 0 && krows >= 9)
            rc = launch_mom_annulus(in_dev, a.out[XRS_STAT_SUM], a.out[XRS_STAT_MEAN], nullptr, nullptr, rows, cols, ld_in, ld_out, kernel,
                                    krows, kcols, halo_top, halo_bot, s);
        if (rc >= 0) return rc;
    }
    if (!gen1 && krows == kcols && krows >= 9 && (stat_mask & ~(m_mean | m_sum))) {
        // several statistics on a circle / box: the extrema walker (ext_impl.h: max / min / range) and the moments walker
        // (mom_impl.h: mean / var / std / sum), each one pass; a sequential `sum` comes from its own kernel.
        float *o_sum = seq_sum ? nullptr : a.out[XRS_STAT_SUM];
        const bool want_mm = a.out[XRS_STAT_MAX] || a.out[XRS_STAT_MIN] || a.out[XRS_STAT_RANGE];
        const bool want_mom = o_sum || a.out[XRS_STAT_MEAN] || a.out[XRS_STAT_VAR] || a.out[XRS_STAT_STD];
        {
            hipStream_t s_mm = s;          // (the two launches on two streams, forked / joined by events, measured no faster than back to back:
                                           //  2.39 vs 2.43 ms in round 3; 2.14 - 2.23 vs 2.19 in round 4, also with the extrema kernel at 2 waves per SIMD:
                                           //  profiles/r04/ab_two_streams.log)
            int rc = 0;
            if (want_mm) {
                rc = try_launch_focal_ext_circle(in_dev, a.out[XRS_STAT_MAX], a.out[XRS_STAT_MIN], a.out[XRS_STAT_RANGE], rows,
                                                 cols, ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s_mm);
                if (rc < 0)
                    rc = try_launch_focal_ext_box(in_dev, a.out[XRS_STAT_MAX], a.out[XRS_STAT_MIN], a.out[XRS_STAT_RANGE], rows,
                                                  cols, ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s_mm);
                // annulus_kernel(1, 1, R, RI): the same walkers, one instantiation per radius pair
                typedef int (*ExtFn)(const float *, float *, float *, float *, long, long, long, long, const double *, int, int, int, int, hipStream_t);
                const ExtFn ann[3] = {try_launch_focal_ext_annulus_a, try_launch_focal_ext_annulus_b, try_launch_focal_ext_annulus_c};
                for (int i = 0; i < 3 && rc < 0; ++i)
                    rc = ann[i](in_dev, a.out[XRS_STAT_MAX], a.out[XRS_STAT_MIN], a.out[XRS_STAT_RANGE], rows, cols, ld_in, ld_out, kernel,
                                krows, kcols, halo_top, halo_bot, s_mm);
            }
            if (rc == 0 && want_mom) {
                rc = try_launch_focal_mom_circle(in_dev, o_sum, a.out[XRS_STAT_MEAN], a.out[XRS_STAT_VAR], a.out[XRS_STAT_STD],
                                                 rows, cols, ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s);
                if (rc < 0)
                    rc = try_launch_focal_mom_box(in_dev, o_sum, a.out[XRS_STAT_MEAN], a.out[XRS_STAT_VAR], a.out[XRS_STAT_STD],
                                                  rows, cols, ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s, box_todo);
                if (rc < 0) rc = launch_mom_annulus(in_dev, o_sum, a.out[XRS_STAT_MEAN], a.out[XRS_STAT_VAR], a.out[XRS_STAT_STD], rows, cols,
                                                    ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s);
            }
            if (rc > 0) return rc;
            if (rc == 0) {
                if (!(seq_sum && a.out[XRS_STAT_SUM])) return 0;
                float *only_sum[XRS_NUM_STATS] = {nullptr};
                only_sum[XRS_STAT_SUM] = a.out[XRS_STAT_SUM];
                const int rc2 = try_walk_f32(in_dev, only_sum, false, rows, cols, ld_in, ld_out, kernel, krows, kcols, halo_top,
                                             halo_bot, s);
                if (rc2 >= 0) return rc2;
                return fail("xrs_focal_stats_f32: no sequential-sum kernel for this mask");
            }
            // (rc < 0: neither a circle nor a box of radius 4..12 -- the kernels below)
        }
    }
    if (stat_mask == (1u << XRS_STAT_MEAN) && krows * kcols >= 49 && !ab_env("XRS_FOCAL_MEAN_RUNS")) {
        // circles and boxes, 7x7 .. 25x25: column walker (running float64 sums over centred runs)
        const int rc = try_walk_f64(in_dev, a.out[XRS_STAT_MEAN], nullptr, nullptr, rows, cols, ld_in, ld_out, kernel,
                                    krows, kcols, halo_top, halo_bot, s);
        if (rc >= 0) return rc;
    }
    if (stat_mask == (1u << XRS_STAT_MEAN) && krows * kcols > 49) {
        // large run-structured masks (circles, boxes, annuli): prefix-sum kernel, O(rows of the mask) per cell
        const int rc = try_launch_focal_mean_runs(in_dev, a.out[XRS_STAT_MEAN], rows, cols, ld_in, ld_out, kernel,
                                                  krows, kcols, halo_top, halo_bot, s);
        if (rc >= 0) return rc;
    }
    a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out;
    a.halo_top = halo_top; a.halo_bot = halo_bot; a.krows = krows; a.kcols = kcols;
    size_t lds;
    if (int rc = plan_tile(a, &lds)) return rc;

    // `kernel == 1` exactly selects a tap (focal.py:323)
    for (int ky = 0; ky < krows; ++ky) {
        unsigned long long bits = 0;
        for (int kx = 0; kx < kcols; ++kx)
            if (kernel[ky * kcols + kx] == 1.0) bits |= 1ull << kx;
        a.mask_rows[ky] = bits;
        a.ntaps += __builtin_popcountll(bits);
    }
    a.inv_ntaps = a.ntaps ? 1.0 / a.ntaps : 0.0;
    (void)work_dev;   // reserved (row-run tables for large masks); the bit-rows travel as kernel arguments

    a.tiles_x = (cols + TW - 1) / TW;
    a.n_tiles = a.tiles_x * ((rows + a.th - 1) / a.th);
    const bool vec = vec_ok(a, stat_mask);
    if (stat_mask != (1u << XRS_STAT_MEAN) && krows * kcols > 49) {
        // large run-structured masks, several statistics: mean / var / std from prefix sums (kxk_runs.hip),
        // the float32 statistics (row-major sum, min, max, range) from one tap walk over the LDS tile
        const unsigned f64_stats = (1u << XRS_STAT_MEAN) | (1u << XRS_STAT_VAR) | (1u << XRS_STAT_STD);
        int rc = 0;
        if (stat_mask & f64_stats) {
            rc = try_walk_f64(in_dev, a.out[XRS_STAT_MEAN], a.out[XRS_STAT_VAR], a.out[XRS_STAT_STD], rows, cols, ld_in, ld_out,
                              kernel, krows, kcols, halo_top, halo_bot, s);
            if (rc > 0) return rc;
        }
        if ((stat_mask & f64_stats) && rc < 0)
            rc = try_launch_focal_meanvar_runs(in_dev, a.out[XRS_STAT_MEAN], a.out[XRS_STAT_VAR], a.out[XRS_STAT_STD], rows,
                                               cols, ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s);
        if (rc > 0) return rc;
        if (rc == 0) {
            if (!(stat_mask & ~f64_stats)) return 0;
            // circles and boxes: column walker (kxk_circle.hip / kxk_box.hip); any other run-structured mask: tap walk
            const int rc2 = try_walk_f32(in_dev, a.out, false, rows, cols, ld_in, ld_out, kernel, krows, kcols, halo_top,
                                         halo_bot, s);
            if (rc2 >= 0) return rc2;
            const bool want_sum = stat_mask >> XRS_STAT_SUM & 1;
            const bool want_mm = stat_mask & ((1u << XRS_STAT_MAX) | (1u << XRS_STAT_MIN) | (1u << XRS_STAT_RANGE));
            if (!want_mm) return launch_focal<0, 0, 3>(a, vec, lds, s);
            if (!want_sum) return launch_focal<0, 0, 4>(a, vec, lds, s);
            return launch_focal<0, 0, 2>(a, vec, lds, s);
        }
    }
    if (stat_mask == (1u << XRS_STAT_MEAN) && !prefer_lds() && !ab_env("XRS_FOCAL_MEAN_DIRECT") &&
        pass_has_compile_time_mask(kernel, krows, kcols)) {
        // circle_kernel(1, 1, 2) / np.ones((3, 3)): the strip kernel of pass.hip without terrain products -- compile-time
        // mask, shared row sums (XRS_FOCAL_MEAN_DIRECT=1: this file's run-time-mask kernel, A/B runs)
        return xrs_raster_pass_f32(in_dev, nullptr, nullptr, nullptr, nullptr, a.out[XRS_STAT_MEAN], kernel, krows, kcols, work_dev,
                                   rows, cols, ld_in, ld_out, 1.0, 1.0, 0.0, 0.0, halo_top, halo_bot, stream);
    }
    if (stat_mask == (1u << XRS_STAT_MEAN)) return dispatch_focal<true>(a, vec, lds, s);
    if ((krows == 5 || krows == 7) && krows == kcols && !gen1 && !(seq_sum && a.out[XRS_STAT_SUM]) && !ab_env("XRS_FOCAL_SW_OFF") &&
        (a.out[XRS_STAT_VAR] || a.out[XRS_STAT_STD] || a.out[XRS_STAT_MEAN])) {
        // small circles / boxes with moments among the statistics: everything from one pass of the strip walker (sw_impl.h)
        int rc = try_launch_focal_sw_circle(in_dev, a.out, rows, cols, ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s);
        if (rc < 0) rc = try_launch_focal_sw_box(in_dev, a.out, rows, cols, ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s);
        if (rc >= 0) return rc;
    }
    if (krows <= 7 && krows == kcols && (krows >= 5 || ab_env("XRS_FOCAL_WALK3")) && !prefer_strip()) {
        // small circles / boxes (5x5, 7x7): all requested statistics from one column-walker kernel
        const bool f32_stats = a.out[XRS_STAT_SUM] || a.out[XRS_STAT_MAX] || a.out[XRS_STAT_MIN] || a.out[XRS_STAT_RANGE];
        const int rc = f32_stats
            ? try_walk_f32(in_dev, a.out, true, rows, cols, ld_in, ld_out, kernel, krows, kcols, halo_top, halo_bot, s)
            : try_walk_f64(in_dev, a.out[XRS_STAT_MEAN], a.out[XRS_STAT_VAR], a.out[XRS_STAT_STD], rows, cols, ld_in,
                           ld_out, kernel, krows, kcols, halo_top, halo_bot, s);
        if (rc >= 0) return rc;
    }
    // the all-statistics kernel always produces the mean internally; give it somewhere to go
    return dispatch_focal<false>(a, vec, lds, s);
}

int xrs_focal_mean3x3(const void *in_dev, int in_is_f64, double *out_dev, int64_t rows, int64_t cols,
                      int64_t ld_in, int64_t ld_out, const double *excludes, int n_excludes, int halo_top,
                      int halo_bot, void *stream) {
    if (!in_dev || !out_dev) return fail("xrs_focal_mean3x3: null pointer");
    if (rows < 0 || cols < 0 || ld_in < cols || ld_out < cols) return fail("xrs_focal_mean3x3: bad shape");
    if (n_excludes < 0 || n_excludes > 8) return fail("xrs_focal_mean3x3: at most 8 exclude values");
    if (rows == 0 || cols == 0) return 0;
    Mean3Args a;
    memset(&a, 0, sizeof(a));
    a.in = in_dev; a.out = out_dev; a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out;
    a.halo_top = halo_top; a.halo_bot = halo_bot; a.n_excl = n_excludes;
    a.only_nan_excl = 1; a.excl_nan = 0;
    for (int i = 0; i < n_excludes; ++i) {
        a.excl[i] = excludes[i];
        if (std::isnan(excludes[i])) a.excl_nan = 1; else a.only_nan_excl = 0;
    }
    // (the strip kernel's 16-byte accesses only need dword / 8-byte alignment; its edge path is cell by cell)
    const bool vec = true;
    if (vec) {
        const long tiles_x = (cols + TW - 1) / TW, n_tiles = tiles_x * ((rows + 15) / 16);
        const unsigned g = (unsigned)xcd_grid(n_tiles, tiles_x);
        if (in_is_f64)
            if (a.only_nan_excl) hipLaunchKernelGGL((focal_mean3_strip_kernel<double, false>), dim3(g), dim3(256), 0, as_stream(stream), a, tiles_x, n_tiles);
            else hipLaunchKernelGGL((focal_mean3_strip_kernel<double, true>), dim3(g), dim3(256), 0, as_stream(stream), a, tiles_x, n_tiles);
        else
            if (a.only_nan_excl) hipLaunchKernelGGL((focal_mean3_strip_kernel<float, false>), dim3(g), dim3(256), 0, as_stream(stream), a, tiles_x, n_tiles);
            else hipLaunchKernelGGL((focal_mean3_strip_kernel<float, true>), dim3(g), dim3(256), 0, as_stream(stream), a, tiles_x, n_tiles);
        XRS_LAUNCH_CHECK();
        return 0;
    }
    const long grid = (rows * cols + 255) / 256;
    if (grid > 0x7fffffffL) return fail("xrs_focal_mean3x3: raster too large");
    if (in_is_f64)
        hipLaunchKernelGGL(focal_mean3_kernel<double>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL(focal_mean3_kernel<float>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    XRS_LAUNCH_CHECK();
    return 0;
}

int xrs_focal_mean3x3_passes(const void *in_dev, int in_is_f64, double *out_dev, double *scratch_dev, int passes,
                             int64_t rows, int64_t cols, const double *excludes, int n_excludes, void *stream) {
    // the reference's loop `for _ in range(passes): out = _mean(out, excludes)` (focal.py:257-259) enqueued in one call:
    // the passes ping-pong between `out_dev` and `scratch_dev` so that the last one lands in `out_dev`
    if (passes < 1) return fail("xrs_focal_mean3x3_passes: passes must be >= 1");
    if (passes > 1 && !scratch_dev) return fail("xrs_focal_mean3x3_passes: more than one pass needs a scratch plane");
    const void *src = in_dev;
    int src_f64 = in_is_f64;
    for (int p = 0; p < passes; ++p) {
        double *dst = ((passes - 1 - p) & 1) ? scratch_dev : out_dev;
        if (int rc = xrs_focal_mean3x3(src, src_f64, dst, rows, cols, cols, cols, excludes, n_excludes, 0, 0, stream)) return rc;
        src = dst;
        src_f64 = 1;
    }
    return 0;
}

}  // extern "C"
