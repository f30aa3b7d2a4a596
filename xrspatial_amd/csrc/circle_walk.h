// Column-walker building blocks for focal masks whose rows are centred runs -- circles and boxes
// (walk_f32_impl.h -> kxk_circle.hip / kxk_box.hip: float32 sum / max / min / range; walk_f64_impl.h ->
// kxk_circle64.hip / kxk_box64.hip: float64 mean / var / std; small radii run both in one kernel).
//
// A lane owns ONE column of a 256 x CTH tile and walks down its input rows.  Every input row y' is read once per
// column (2R+1 neighbouring cells, L1 hits) and contributes to the 2R+1 output rows y' - dy that see it; their
// partial results live in a register ring that shifts one slot per input row, so the loop body is the same for
// every row, and because the circle's half-widths are compile-time constants every register index is static.
// No LDS, no barriers.  Reference semantics: xrspatial/focal.py:226-258 (numba reducers) over the cells under
// `kernel == 1` in row-major order (:268-326), NaN cells skipped, window clipped at the raster edge.
#pragma once
#include "xrs_common.h"

#include <cmath>
#include <cstdlib>
#include "wave_reduce.h"
#include <type_traits>

namespace xrs {

constexpr int CTH = 128;     // output rows per tile (input rows walked: CTH + 2R)

// Mask shapes the walkers are instantiated for: every row of the (2R+1)^2 mask is ONE run centred on the kernel's
// centre column, with a compile-time half-width hw(R, |dy|).
struct CircleShape {      // circle_kernel on square cells: largest dx with dx^2 + dy^2 <= R^2 (convolution.py:144)
    static constexpr int hw(int R, int dy) {
        int h = 0;
        while ((h + 1) * (h + 1) + dy * dy <= R * R) ++h;
        return h;
    }
    static constexpr int hwi(int, int) { return -1; }      // no hole
};
struct BoxShape {         // np.ones((2R+1, 2R+1))
    static constexpr int hw(int R, int) { return R; }
    static constexpr int hwi(int, int) { return -1; }
};
// annulus_kernel(1, 1, R, RI) = circle_kernel(R) - circle_kernel(RI) (convolution.py:199-259): a row at offset dy is the
// centred run of half-width hw(dy) WITHOUT the centred run of half-width hwi(dy) (-1: no hole in this row) -- two runs,
// but every sum over them is a difference of two centred-run sums, and every extremum one over a "shell" of cell pairs.
template <int RI>
struct AnnulusShape {
    static constexpr int hw(int R, int dy) { return CircleShape::hw(R, dy); }
    static constexpr int hwi(int, int dy) { return dy <= RI ? CircleShape::hw(RI, dy) : -1; }
};
template <typename Shape>
constexpr bool shape_has_hole(int R) {
    for (int dy = 0; dy <= R; ++dy)
        if (Shape::hwi(R, dy) >= 0) return true;
    return false;
}
// rows at offsets d1, d2 (>= 0) are the same run pattern; d is the smallest offset with its pattern
template <typename Shape>
constexpr bool shape_same_row(int R, int d1, int d2) { return Shape::hw(R, d1) == Shape::hw(R, d2) && Shape::hwi(R, d1) == Shape::hwi(R, d2); }
template <typename Shape>
constexpr bool shape_first_row(int R, int d) {
    for (int e = 0; e < d; ++e)
        if (shape_same_row<Shape>(R, e, d)) return false;
    return true;
}
template <typename Shape>
constexpr int shape_row_cells(int R, int dy) {             // taps of the row at offset |dy|
    return 2 * Shape::hw(R, dy) + 1 - (Shape::hwi(R, dy) >= 0 ? 2 * Shape::hwi(R, dy) + 1 : 0);
}

// the rows of a shape as compile-time tables: hw / hwi per |dy| and pat[d] = the smallest offset with the same (hw, hwi).
// (Evaluating Shape::hw -- a loop -- per iteration of the walkers' unrolled loops is fine for one call; the row-pattern tests
// of the hole shapes needed four to six and stopped folding: the round loop came out with 2400 scalar instructions.)
template <int R, typename Shape>
struct ShapeRows {
    int hw[R + 1], hwi[R + 1], pat[R + 1];
    constexpr ShapeRows() : hw{}, hwi{}, pat{} {
        for (int d = 0; d <= R; ++d) { hw[d] = Shape::hw(R, d); hwi[d] = Shape::hwi(R, d); }
        for (int d = 0; d <= R; ++d) {
            pat[d] = d;
            for (int e = d - 1; e >= 0; --e)
                if (hw[e] == hw[d] && hwi[e] == hwi[d]) pat[d] = e;
        }
    }
};

template <typename Shape>
constexpr int shape_taps(int R) {
    int n = 0;
    for (int dy = -R; dy <= R; ++dy) n += shape_row_cells<Shape>(R, dy < 0 ? -dy : dy);
    return n;
}

// host: the inner radius of a K x K mask that could be annulus_kernel(1, 1, K / 2, RI): the zeros of the centre row run from
// the centre to dx = RI (circle_kernel(RI) has half-width RI in its middle row); -1: the centre cell is set (no hole)
inline int annulus_inner_radius(const double *kernel, int K) {
    const int R = K / 2;
    int ri = -1;
    while (ri + 1 <= R && kernel[R * K + R + ri + 1] != 1.0) ++ri;
    return ri;
}

template <int R, typename Shape>
inline bool is_shape(const double *kernel) {
    constexpr int K = 2 * R + 1;
    for (int ky = 0; ky < K; ++ky) {
        const int dy = ky < R ? R - ky : ky - R, h = Shape::hw(R, dy), hi = Shape::hwi(R, dy);
        for (int kx = 0; kx < K; ++kx) {
            const int dx = kx < R ? R - kx : kx - R;
            if ((kernel[ky * K + kx] == 1.0) != (dx <= h && dx > hi)) return false;
        }
    }
    return true;
}

// a[s] <- a[(s + U) mod K] for every s, in place: the permutation is gcd(K, U) cycles of length K / gcd, walked with
// one temporary each (static indices after unrolling, so this is K register moves and nothing else).
constexpr int walk_gcd(int a, int b) { return b == 0 ? a : walk_gcd(b, a % b); }
template <int K, int U, typename T>
__device__ __forceinline__ void ring_rotate(T (&a)[K]) {
    constexpr int G = walk_gcd(K, U % K == 0 ? K : U % K), LEN = K / G;
    if (U % K == 0) return;
#pragma unroll
    for (int c = 0; c < G; ++c) {
        const T tmp = a[c];
        int sl = c;
#pragma unroll
        for (int k = 0; k + 1 < LEN; ++k) {
            const int from = (sl + U) % K;
            a[sl] = a[from];
            sl = from;
        }
        a[sl] = tmp;
    }
}

template <int K, int U, int N>
__device__ __forceinline__ void ring_rotate(float (&a)[K][N]) {                // the same for N values per slot
    constexpr int G = walk_gcd(K, U % K == 0 ? K : U % K), LEN = K / G;
    if (U % K == 0) return;
#pragma unroll
    for (int c = 0; c < G; ++c) {
        float tmp[N];
#pragma unroll
        for (int o = 0; o < N; ++o) tmp[o] = a[c][o];
        int sl = c;
#pragma unroll
        for (int k = 0; k + 1 < LEN; ++k) {
            const int from = (sl + U) % K;
#pragma unroll
            for (int o = 0; o < N; ++o) a[sl][o] = a[from][o];
            sl = from;
        }
#pragma unroll
        for (int o = 0; o < N; ++o) a[sl][o] = tmp[o];
    }
}


// Work order of the third-generation walkers (ext_impl.h, mom_impl.h): the workgroups on the rim of the raster first,
// then the interior in one contiguous band per XCD.  A rim workgroup runs its predicated edge walk 2-3x longer than an
// interior one; left where the row-major order puts them, the bottom row of tiles is the last thing one XCD starts and the
// whole launch waits for it (measured on the 25x25 moments kernel: 1.85 ms -> see DESIGN.md), and the top row delays
// another.  Dealt out first, round-robin over the XCDs, they are spread evenly and their tails hide behind interior work.
// (gy, gx) of workgroup `block` in a gw x gh grid of workgroup tiles; false for the surplus blocks of the padded grid.
// Output rows per tile of the third-generation walkers.  Every tile pays 2R rows of run-in before its first output row
// (128-row tiles walk 1.19 input rows per output row at radius 12, 256-row tiles 1.09), but the launch also has to come
// out in whole ROUNDS of resident workgroups: 4096 workgroups on 768 slots are 5.3 rounds and cost 6.  Chosen per launch:
// the tile height whose (rounds x input rows per tile) is smallest, for `groups_x` workgroups per tile row, `wg_per_cu`
// resident workgroups per CU and rounds of `u` rows.  Measured on 16384^2, radius 12 (profiles/r03): moments 1.65 ms with
// 128-row tiles, 1.36 with 256, 2.2 with 384.  XRS_WALK_TILE_ROWS overrides (A/B runs).
// resident 256-thread workgroups per CU of a kernel, as the runtime computes it from the kernel's registers and LDS
template <typename K>
inline int walk3_wg_per_cu(K kernel_fn, int fallback) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel_fn, 256, 0) != hipSuccess || n < 1) return fallback;
    return n;
}

inline int walk3_tile_base(long rows, long groups_x, int radius, int u, int wg_per_cu) {
    const char *e = ab_env("XRS_WALK_TILE_ROWS");
    if (e && atoi(e) >= 16) return atoi(e);
    static thread_local int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }
    // (radius < 10, or a raster too short to fill the chip several times: 128 -- the lighter kernels of small windows are
    //  bound by HBM, not by rounds, and measured 10-15 % slower on 256-row tiles: box 11x11 seven statistics 2.39 vs 2.75 ms)
    if (radius < 10 || rows < 8192) return 128;
    const long slots = (long)n_cu * wg_per_cu;
    int best = 256;
    double best_cost = 1e300;
    for (int base = 192; base <= 320; base += 32) {
        const long nin = ((base + 2 * radius + u - 1) / u) * u, wth = nin - 2 * radius;
        const long tiles_y = (rows + wth - 1) / wth;
        const long rounds = (groups_x * tiles_y + slots - 1) / slots;
        // (a partly filled last round still costs a round: its workgroups run alone on their CUs, not faster)
        const double cost = (double)rounds * (double)nin;
        if (cost < best_cost) { best_cost = cost; best = base; }
    }
    return best;
}

struct RimFirst {
    long gw, gh, n_rim, n_all;
    // mode 1: rim first; 0: row-major tiles in one contiguous band per XCD (rounds 1-2: XRS_RIM_FIRST=0, A/B runs)
    __host__ __device__ __forceinline__ RimFirst(long gw_, long gh_, int mode = 1) : gw(gw_), gh(gh_) {      // (forced: a CALL of this
                                                                                                        //  constructor from the largest kernels faulted at address 0)
        n_all = gw * gh;
        n_rim = mode == 0 ? -1 : (gw <= 2 || gh <= 2) ? n_all : 2 * gw + 2 * (gh - 2);
    }
    static int mode_from_env() {
        const char *e = ab_env("XRS_RIM_FIRST");
        return e && e[0] == '0' ? 0 : 1;
    }
    __host__ long grid() const { return n_rim < 0 ? xcd_grid(n_all, 0) : n_rim + xcd_grid(n_all - n_rim, 0); }
    __device__ __forceinline__ bool locate(long block, long &gy, long &gx) const {
        if (n_rim < 0) {
            const long t = xcd_tile(block, n_all, 0);
            if (t < 0) return false;
            gy = t / gw; gx = t - gy * gw;
            return true;
        }
        if (block < n_rim) {
            if (n_rim == n_all) { gy = block / gw; gx = block - gy * gw; return true; }
            if (block < gw) { gy = 0; gx = block; return true; }
            if (block < 2 * gw) { gy = gh - 1; gx = block - gw; return true; }
            const long r = block - 2 * gw;
            gy = 1 + (r >> 1);
            gx = (r & 1) ? gw - 1 : 0;
            return true;
        }
        const long iw = gw - 2, ih = gh - 2;
        const long t = xcd_tile(block - n_rim, iw * ih, 0);
        if (t < 0) return false;
        gy = 1 + t / iw;
        gx = 1 + (t - (gy - 1) * iw);
        return true;
    }
};

struct WalkGeom {
    const float *in;
    long rows, cols, ld_in, ld_out;
    int halo_top, halo_bot;
    long tiles_x, n_tiles;
};

// The 2R+1 cells of input row yy around column x (NaN outside the raster / the shard's halo rows).
// EDGE = false: the wave's columns xw - R .. xw + 63 + R all lie inside the raster (wave-uniform fact): no column tests.
template <int R, bool EDGE>
__device__ __forceinline__ void walk_load_row(const WalkGeom &g, long yy, long xw, int lane, float (&v)[2 * R + 1]) {
    constexpr int K = 2 * R + 1;
    const long x = xw + lane;
    const float qnan = nan_f32();
    const bool row_ok = yy >= -(long)g.halo_top && yy < g.rows + g.halo_bot;        // wave-uniform
    if (row_ok) {
        const float *p = g.in + yy * g.ld_in + xw + lane;                           // scalar row base + lane
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (EDGE) {
                const long xc = x + k - R;
                v[k] = (xc >= 0 && xc < g.cols) ? p[k - R] : qnan;
            } else {
                v[k] = p[k - R];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = qnan;
    }
}

typedef float walk_v2f __attribute__((ext_vector_type(2)));

// ---- float32 row-major sum, max, min, range
//   * max / min: the running extremum over the centred run of half-width h, m_h = min(m_{h-1}, v[-h], v[+h]), costs
//     R `v_min3` per row and is exactly what output row y' - dy needs for h = hw(dy): 2R+1 more `v_min` instead of
//     one per tap (441 -> 37 per cell for R = 12);
//   * sum: the reference adds the taps sequentially in float32 (numba nansum keeps the array dtype), so the rounding
//     of every partial sum is part of the result and nothing can be shared or re-associated -- but output rows dy
//     and -dy append the SAME values in the SAME order, so their two accumulators share one packed `v_pk_add_f32`
//     per value (441 -> 233 instructions per cell).  NaN cells are added as +0.0 (exact: an accumulator that
//     starts at +0.0 is never -0.0) and skipped by IEEE minNum / maxNum; a window without a valid cell gives sum 0
//     and NaN for max / min / range, like the reference.
template <int R, typename Shape, bool WANT_SUM, bool WANT_MM>
struct WalkF32 {
    static constexpr int K = 2 * R + 1;
    walk_v2f sp[R > 0 ? R : 1];   // (slot j, slot 2R - j) for j < R; ring slot j = output row (input row) - (j - R)
    float sc;                     // slot R (dy = 0)
    float mn[K], mx[K];

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < R; ++j) sp[j] = (walk_v2f)(0.0f);
        sc = 0.0f;
#pragma unroll
        for (int j = 0; j < K; ++j) { mn[j] = INFINITY; mx[j] = -INFINITY; }
    }

    __device__ __forceinline__ void row(const float (&v)[K]) {
        if (WANT_MM) {
            if (!shape_has_hole<Shape>(R)) {
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
                        if (Shape::hw(R, dy < 0 ? -dy : dy) == h) {
                            mn[j] = fminf(mn[j], lo);
                            mx[j] = fmaxf(mx[j], hi);
                        }
                    }
                }
            } else {
                // rows with a hole: the extrema over the shell of cell pairs hwi < |dx| <= hw, per distinct row pattern
                constexpr ShapeRows<R, Shape> T{};
#pragma unroll
                for (int d = 0; d <= R; ++d) {
                    if (T.pat[d] != d) continue;
                    const int h1 = T.hw[d], h0 = T.hwi[d];
                    float lo = h0 < 0 ? v[R] : INFINITY, hi = h0 < 0 ? v[R] : -INFINITY;
#pragma unroll
                    for (int h = (h0 < 0 ? 1 : h0 + 1); h <= h1; ++h) {
                        lo = fminf(fminf(lo, v[R - h]), v[R + h]);
                        hi = fmaxf(fmaxf(hi, v[R - h]), v[R + h]);
                    }
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const int dy = j - R;
                        if (T.pat[dy < 0 ? -dy : dy] == d) {
                            mn[j] = fminf(mn[j], lo);
                            mx[j] = fmaxf(mx[j], hi);
                        }
                    }
                }
            }
        }
        if (WANT_SUM) {
            float z[K];
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] = isnan(v[k]) ? 0.0f : v[k];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int h = Shape::hw(R, R - j), h0 = Shape::hwi(R, R - j);
#pragma unroll
                for (int k = R - h; k <= R + h; ++k)
                    if (k - R > h0 || R - k > h0) sp[j] += (walk_v2f)(z[k]);
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (k - R > Shape::hwi(R, 0) || R - k > Shape::hwi(R, 0)) sc += z[k];
        }
    }

    // results of the completed output row (slot 2R)
    __device__ __forceinline__ void emit(long off, float *out_sum, float *out_max, float *out_min, float *out_range) const {
        if (WANT_SUM && out_sum) out_sum[off] = R > 0 ? sp[0].y : sc;
        if (WANT_MM) {
            const float qnan = nan_f32();
            const bool none = mn[2 * R] > mx[2 * R];                // no valid cell under the window
            if (out_max) out_max[off] = none ? qnan : mx[2 * R];
            if (out_min) out_min[off] = none ? qnan : mn[2 * R];
            if (out_range) out_range[off] = none ? qnan : mx[2 * R] - mn[2 * R];
        }
    }

    __device__ __forceinline__ void shift() {
        if (WANT_SUM && R > 0) {
            // slots 0..R-1 are the .x halves (moving up), slots R+1..2R the .y halves of sp[2R - slot]
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
};

// ---- float64 mean, var, std (numba nanmean / nanvar: float64 accumulation, two-pass variance, float32 store)
// Per input row a lane forms, from the centre outwards, the float64 sum S_h and sum of squares Q_h of the SHIFTED
// values d = v - c over the centred run of half-width h, plus the count C_h of valid cells; output row y' - dy
// adds (S, Q, C) of h = hw(dy) to its ring slot: 2R+1 triple-adds per row and column instead of one per tap.  c is
// the lane's own column value at the middle row of the tile (every output is produced by one lane, so the shift may
// differ per lane, and a nearby value keeps d small).  At the end  mean = c + S/n,  var = (Q - S^2/n)/n;  if that
// one-pass variance is not comfortably above the rounding noise of its operands (flat patches inside high-relief
// tiles, +-inf under the window) the output is recomputed tap by tap with the reference's two-pass loops.  Sums are
// re-associated relative to the reference's row-major order: invisible after the float32 rounding (tests: 1e-6).
__device__ __forceinline__ double walk_rcp(int n) {
    const double c = (double)n;
    double r = __builtin_amdgcn_rcp(c);
    r = fma(fma(-c, r, 1.0), r, r);
    return n ? r : nan("");
}

// One window by the reference's two passes (focal.py _calc_mean / _calc_var: the mean, then the squared deviations from
// it; NaN cells skipped), computed by the WHOLE wave: lane l takes taps l, l + 64, ... of the (2R+1)^2 square, float64
// partial sums meet in a wave reduction.  (A lane looping over its own window alone is 2 * ntaps dependent loads -- about
// 0.2 ms per output for 25x25, which one nodata boundary turned into a 30 ms kernel.)  Flat windows stay exact: every
// partial sum of m <= 2^10 equal float32 values is exact in float64, so the mean is the value and every deviation 0.
template <int R, typename Shape>
__device__ __forceinline__ void walk_exact_window(const WalkGeom &g, long yo, long xs, int lane, double &mean, double &var) {
    constexpr int K = 2 * R + 1, NT = K * K, IT = (NT + 63) / 64;
    const long y_lo = -(long)g.halo_top, y_hi = g.rows + g.halo_bot;
    auto tap = [&](int i) -> float {                       // NaN: not under the mask / outside the raster
        const int idx = lane + 64 * i;
        const int ky = idx / K, kx = idx - ky * K;
        const int dy = ky < R ? R - ky : ky - R, dx = kx < R ? R - kx : kx - R;
        const long yr = yo - R + ky, xr = xs - R + kx;
        const bool ok = idx < NT && dx <= Shape::hw(R, dy) && dx > Shape::hwi(R, dy) && yr >= y_lo && yr < y_hi && xr >= 0 && xr < g.cols;
        return ok ? g.in[yr * g.ld_in + xr] : nan_f32();
    };
    double s = 0.0, m = 0.0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const float v = tap(i);
        if (!isnan(v)) { s += (double)v; m += 1.0; }
    }
    s = wave_reduce<WrSum>(s);                             // (wave_reduce.h: DPP, wave-uniform results)
    m = wave_reduce<WrSum>(m);
    mean = m > 0.0 ? s / m : nan("");                      // true division: a flat window must give its value exactly
    double dev = 0.0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const float v = tap(i);
        if (!isnan(v)) { const double d = (double)v - mean; dev = fma(d, d, dev); }
    }
    dev = wave_reduce<WrSum>(dev);
    var = m > 0.0 ? dev / m : nan("");
}

template <int R, typename Shape, bool WANT_VAR = true>       // WANT_VAR = false: mean only (no squares, no guard)
struct WalkF64 {
    static constexpr int K = 2 * R + 1;
    double sd[K], sq[WANT_VAR ? K : 1];
    int cn[K];
    float amax, cf;               // running max |v - c| over everything this lane has read (guard scale); the shift

    __device__ __forceinline__ void init(const WalkGeom &g, long y0, long x) {
#pragma unroll
        for (int j = 0; j < K; ++j) { sd[j] = 0.0; cn[j] = 0; }
#pragma unroll
        for (int j = 0; j < (WANT_VAR ? K : 1); ++j) sq[j] = 0.0;
        amax = 0.0f;
        cf = 0.0f;
        const long yc = (y0 + CTH / 2 < g.rows ? y0 + CTH / 2 : g.rows - 1);
        float c0 = nan_f32();
        if (x < g.cols) c0 = g.in[yc * g.ld_in + x];
        // (a lane on nodata borrows a neighbour's value: a shift of 0 would make `amax` -- the guard's scale -- the data's level)
        const unsigned long long have = __ballot(isfinite(c0));
        const float c_any = __shfl(c0, have ? __ffsll((long long)have) - 1 : 0);
        cf = isfinite(c0) ? c0 : have ? c_any : 0.0f;
    }

    __device__ __forceinline__ void row(const float (&v)[K]) {
        const double shift = (double)cf;
        if (shape_has_hole<Shape>(R)) {
            // rows with a hole: every distinct row pattern summed over its own cells (hwi < |dx| <= hw), centre outwards
            constexpr ShapeRows<R, Shape> T{};
#pragma unroll
            for (int d = 0; d <= R; ++d) {
                if (T.pat[d] != d) continue;
                const int h1 = T.hw[d], h0 = T.hwi[d];
                double S = 0.0, Q = 0.0;
                int C = 0;
#pragma unroll
                for (int h = (h0 < 0 ? 0 : h0 + 1); h <= h1; ++h) {
#pragma unroll
                    for (int side = 0; side < (h == 0 ? 1 : 2); ++side) {
                        const float val = v[side == 0 ? R - h : R + h];
                        const bool ok = !isnan(val);
                        const double dd = ok ? (double)val - shift : 0.0;
                        S += dd;
                        C += ok ? 1 : 0;
                        if (WANT_VAR) {
                            Q = fma(dd, dd, Q);
                            amax = fmaxf(amax, isfinite(val) ? fabsf(val - cf) : 0.0f);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int dy = j - R;
                    if (T.pat[dy < 0 ? -dy : dy] == d) {
                        sd[j] += S;
                        cn[j] += C;
                        if (WANT_VAR) sq[j] += Q;
                    }
                }
            }
            return;
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
                C += ok ? 1 : 0;
                if (WANT_VAR) {
                    Q = fma(d, d, Q);
                    amax = fmaxf(amax, isfinite(val) ? fabsf(val - cf) : 0.0f);   // (+-inf: the sums go non-finite -> exact path)
                }
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int dy = j - R;
                if (Shape::hw(R, dy < 0 ? -dy : dy) == h) {
                    sd[j] += S;
                    cn[j] += C;
                    if (WANT_VAR) sq[j] += Q;
                }
            }
        }
    }

    // completed output row yo (slot 2R), column x.  Returns true WITHOUT storing when the window is ill-conditioned or
    // exactly flat (or holds +-inf): the caller recomputes it with `walk_exact_window`.
    __device__ __forceinline__ bool emit(const WalkGeom &g, long yo, long x, float *out_mean, float *out_var,
                                         float *out_std) const {
        const double shift = (double)cf;
        const int n = cn[2 * R];
        const double inv = walk_rcp(n);
        // mean of the shifted values: S x (1 / n) and one correction step.  Without it the quotient is a rounding off where S / n is
        // exact -- a window of equal cells v has S = n (v - c) exactly, and c + S (1 / n) came out as v + 1e-16 |v - c|: a lake of
        // 0.0 on a plateau at -1e5 read -7.7e-12 where the reference's nanmean says 0 (tests/fuzz_parity.py --windows).  With the
        // residual S - q n folded back in, the quotient of an exact multiple is exact.
        double ms = sd[2 * R] * inv;
        const double corrected = fma(fma(-ms, (double)n, sd[2 * R]), inv, ms);
        ms = corrected == corrected ? corrected : ms;                   // (+-inf under the window: its residual is inf - inf; the mean stays +-inf)
        if (!WANT_VAR) {
            // mean only: c + S/n needs no guard (a +-inf under the window gives +-inf / NaN like the reference's sum)
            if (out_mean) out_mean[yo * g.ld_out + x] = (float)(shift + ms);
            return false;
        }
        const double ssd = sq[WANT_VAR ? 2 * R : 0] - sd[2 * R] * ms;
        const double mean = shift + ms;
        double var = (ssd > 0.0 ? ssd : 0.0) * inv;
        // rounding noise of Q and S^2/n is ~ ntaps * eps * max(d^2); 1e6 of headroom as in kxk_runs.hip
        const double guard = 1e-9 * (double)shape_taps<Shape>(R) * ((double)amax * (double)amax);
        if (n == 1 && ssd == ssd) var = 0.0;                             // one valid (finite) cell: its value, variance 0
        else if (n != 0 && !(ssd >= guard)) return true;
        const long off = yo * g.ld_out + x;
        if (out_mean) out_mean[off] = (float)mean;
        if (out_var) out_var[off] = (float)var;
        if (out_std) out_std[off] = (float)sqrt(var);
        return false;
    }

    __device__ __forceinline__ void shift() {
#pragma unroll
        for (int j = K - 1; j > 0; --j) { sd[j] = sd[j - 1]; cn[j] = cn[j - 1]; }
        if (WANT_VAR) {
#pragma unroll
            for (int j = K - 1; j > 0; --j) sq[j] = sq[j - 1];
        }
        sd[0] = 0.0; sq[0] = 0.0; cn[0] = 0;
    }
};

// ---- convolve_2d with ONE weight value on a circle / box (the normalised kernels focal.hotspots and the reference's
// examples use: circle_kernel / np.ones divided by their sum).  Reference _convolve_2d_numpy (convolution.py:285-313):
// float64 `num += kernel[k, h] * data[..]` over the WHOLE (2R+1)^2 window (zero weights included, so a NaN or inf
// anywhere in the square makes the result NaN), NaN border of R cells, float32 store.  Here: w * (ntaps * c + S) with S
// the walker's float64 sum of shifted values under the mask; windows whose square holds a non-finite cell (counted
// with a running row total) are recomputed tap by tap in the reference's order.
template <int R, typename Shape>
struct WalkConv {
    static constexpr int K = 2 * R + 1;
    double sd[K];
    int bad_at[K];                // running non-finite total when the slot was opened
    int bad_total;
    float cf;

    __device__ __forceinline__ void init(const WalkGeom &g, long y0, long x) {
#pragma unroll
        for (int j = 0; j < K; ++j) { sd[j] = 0.0; bad_at[j] = 0; }
        bad_total = 0;
        cf = 0.0f;
        const long yc = (y0 + CTH / 2 < g.rows ? y0 + CTH / 2 : g.rows - 1);
        if (x < g.cols) {
            const float c0 = g.in[yc * g.ld_in + x];
            if (isfinite(c0)) cf = c0;
        }
    }

    __device__ __forceinline__ void row(const float (&v)[K]) {
        const double shift = (double)cf;
        double S = 0.0;
        int nbad = 0;
        if constexpr (shape_has_hole<Shape>(R)) {
            // annuli: centred partial sums P[h] of |dx| <= h; a row at offset dy takes P[hw] - P[hwi]
            double P[R + 1];
#pragma unroll
            for (int h = 0; h <= R; ++h) {
#pragma unroll
                for (int side = 0; side < (h == 0 ? 1 : 2); ++side) {
                    const float val = v[side == 0 ? R - h : R + h];
                    const bool ok = isfinite(val);
                    S += ok ? (double)val - shift : 0.0;
                    nbad += ok ? 0 : 1;
                }
                P[h] = S;
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int dy = j < R ? R - j : j - R, h1 = Shape::hw(R, dy), h0 = Shape::hwi(R, dy);
                sd[j] += h0 >= 0 ? P[h1] - P[h0] : P[h1];
            }
            bad_total += nbad;
            return;
        }
#pragma unroll
        for (int h = 0; h <= R; ++h) {
#pragma unroll
            for (int side = 0; side < (h == 0 ? 1 : 2); ++side) {
                const float val = v[side == 0 ? R - h : R + h];
                const bool ok = isfinite(val);
                S += ok ? (double)val - shift : 0.0;
                nbad += ok ? 0 : 1;
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int dy = j - R;
                if (Shape::hw(R, dy < 0 ? -dy : dy) == h) sd[j] += S;
            }
        }
        bad_total += nbad;            // the whole row segment x-R .. x+R belongs to every window that sees this row
    }

    // completed output row yo (slot 2R), column x; `weights`: the (2R+1)^2 float64 kernel in device memory
    __device__ __forceinline__ void emit(const WalkGeom &g, long yo, long x, float *out, double w,
                                         const double *weights) const {
        const long y_lo = -(long)g.halo_top, y_hi = g.rows + g.halo_bot;
        float res = nan_f32();
        if (yo - R >= y_lo && yo + R < y_hi && x - R >= 0 && x + R < g.cols) {
            if (bad_total - bad_at[2 * R] == 0) {
                res = (float)(w * ((double)shape_taps<Shape>(R) * (double)cf + sd[2 * R]));
            } else {
                double num = 0.0;                                       // the reference's loop, zero weights included
                for (int ky = 0; ky < K; ++ky)
                    for (int kx = 0; kx < K; ++kx)
                        num += weights[ky * K + kx] * (double)g.in[(yo - R + ky) * g.ld_in + (x - R + kx)];
                res = (float)num;
            }
        }
        st_stream(&out[yo * g.ld_out + x], res);
    }

    __device__ __forceinline__ void shift() {
#pragma unroll
        for (int j = K - 1; j > 0; --j) { sd[j] = sd[j - 1]; bad_at[j] = bad_at[j - 1]; }
        sd[0] = 0.0;
        bad_at[0] = bad_total;
    }
};

// the conv walker over output rows [y0, y_end) of the 64 columns starting at xw (wave-uniform)
template <int R, typename Shape>
__device__ __forceinline__ void walk_conv_columns(const WalkGeom &g, float *out, double w, const double *weights, long xw,
                                                  int lane, long y0, long y_end) {
    constexpr int K = 2 * R + 1;
    const long x = xw + lane;
    if (xw >= g.cols) return;
    WalkConv<R, Shape> c;
    c.init(g, y0, x);
    auto walk = [&](auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        for (long yy = y0 - R; yy < y_end + R; ++yy) {
            float v[K];
            walk_load_row<R, EDGE>(g, yy, xw, lane, v);
            c.row(v);
            const long yo = yy - R;
            if (yo >= y0 && (!EDGE || x < g.cols)) c.emit(g, yo, x, out, w, weights);
            c.shift();
        }
    };
    if (xw >= R && xw + 64 + R <= g.cols) walk(std::false_type{});
    else walk(std::true_type{});
}

template <int R, typename Shape>
__device__ __forceinline__ void walk_conv_tile(const WalkGeom &g, float *out, double w, const double *weights) {
    const long t = xcd_tile(blockIdx.x, g.n_tiles, XCD_UNIT(XRS_XCD_WALK, g.tiles_x));
    if (t < 0) return;
    const long ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long y0 = ty * CTH;
    walk_conv_columns<R, Shape>(g, out, w, weights, tx * 256 + wv * 64, lane, y0, y0 + CTH < g.rows ? y0 + CTH : g.rows);
}

// host: does `kernel` put ONE weight value on exactly the cells of the shape (zero elsewhere)?
template <int R, typename Shape>
inline bool is_uniform_shape(const double *kernel, double *weight) {
    constexpr int K = 2 * R + 1;
    const double w = kernel[R * K + R + Shape::hw(R, 0)];     // (the rim cell of the centre row: an annulus has no centre cell)
    if (!(w != 0.0) || !std::isfinite(w)) return false;
    for (int ky = 0; ky < K; ++ky) {
        const int dy = ky < R ? R - ky : ky - R, h = Shape::hw(R, dy), hi = Shape::hwi(R, dy);
        for (int kx = 0; kx < K; ++kx) {
            const int dx = kx < R ? R - kx : kx - R;
            if (kernel[ky * K + kx] != (dx <= h && dx > hi ? w : 0.0)) return false;
        }
    }
    *weight = w;
    return true;
}

// Nodata regions: is EVERY cell a wave's tile can see NaN (raster columns [x_lo, x_hi), rows [y_lo, y_hi), clipped to the
// raster and the shard's halo rows -- cells outside count as NaN, as every walker treats them)?  Stops at the first row
// with a valid cell, so a tile with data costs one row; a tile inside a nodata region costs one pass over cells the
// fast walker just brought into L2, and then needs no walk at all (the exact walkers are 5-10x slower than the fast ones,
// and a raster with an ocean has many such tiles).
template <int NL>                  // NL >= (x_hi - x_lo + 63) / 64: loads per lane and row
__device__ __forceinline__ bool walk_tile_all_nan(const WalkGeom &g, long x_lo, long x_hi, long y_lo, long y_hi, int lane) {
    x_lo = x_lo < 0 ? 0 : x_lo;
    x_hi = x_hi > g.cols ? g.cols : x_hi;
    y_lo = y_lo < -(long)g.halo_top ? -(long)g.halo_top : y_lo;
    y_hi = y_hi > g.rows + g.halo_bot ? g.rows + g.halo_bot : y_hi;
    if (x_lo >= x_hi) return true;
    // the first row alone, then 16 rows per verdict, every load of a batch independent of the others (columns clamped into
    // the range instead of tested: a short-circuit `valid || v == v` in a loop with a per-lane trip count made every load
    // wait for the one before -- 48 memory latencies per batch, and an all-NaN tile as expensive as a walked one)
    for (long y = y_lo, step = 1; y < y_hi; y += step, step = 16) {
        unsigned valid = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r >= step || y + r >= y_hi) break;
            const float *p = g.in + (y + r) * g.ld_in;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const long x = x_lo + lane + 64 * i;
                const float v = p[x < x_hi ? x : x_hi - 1];
                valid |= v == v ? 1u : 0u;
            }
        }
        if (__any(valid != 0)) return false;
    }
    return true;
}

// ... and its results: no valid cell under any window (focal.py:268-302: nanmean / nanvar / nanstd / nanmax / nanmin of an
// all-NaN window are NaN, nansum is 0).  `fill_sum`: the value for the sum plane (convolve_2d: NaN).
__device__ __forceinline__ void walk_fill_no_data(const WalkGeom &g, float *const *planes, int n_planes, float *sum_plane,
                                                  float fill_sum, long x_lo, long x_hi, long y0, long y_end, int lane) {
    x_hi = x_hi > g.cols ? g.cols : x_hi;
    const float qnan = nan_f32();
    for (long y = y0; y < y_end; ++y) {
        for (long x = x_lo + lane; x < x_hi; x += 64) {
            const long off = y * g.ld_out + x;
            for (int i = 0; i < n_planes; ++i) if (planes[i]) planes[i][off] = qnan;
            if (sum_plane) sum_plane[off] = fill_sum;
        }
    }
}

struct WalkOuts {
    float *sum, *max, *min, *range, *mean, *var, *std;      // any may be NULL
};

// One walker body for every combination: F32 = run the float32 statistics, F64 = run the moments.  The wave walks
// output rows [y0, y_end) of the 64 columns starting at xw (wave-uniform); kxk_wide.hip calls it for the tiles its
// float32 fast path hands back (non-finite cells under a window, ill-conditioned sums).
template <int R, typename Shape, bool F32, bool WANT_SUM, bool WANT_MM, bool F64, bool WANT_VAR = true>
__device__ __forceinline__ void walk_columns(const WalkGeom &g, const WalkOuts &o, long xw, int lane, long y0, long y_end) {
    constexpr int K = 2 * R + 1;
    const long x = xw + lane;
    if (xw >= g.cols) return;

    WalkF32<R, Shape, WANT_SUM, WANT_MM> a32;
    WalkF64<R, Shape, WANT_VAR> a64;
    if (F32) a32.init();
    if (F64) a64.init(g, y0, x);
    auto walk = [&](auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        for (long yy = y0 - R; yy < y_end + R; ++yy) {
            float v[K];
            walk_load_row<R, EDGE>(g, yy, xw, lane, v);
            if (F32) a32.row(v);
            if (F64) a64.row(v);
            const long yo = yy - R;                     // the output row that is now complete
            bool redo = false;
            if (yo >= y0 && (!EDGE || x < g.cols)) {
                if (F32) a32.emit(yo * g.ld_out + x, o.sum, o.max, o.min, o.range);
                if (F64) redo = a64.emit(g, yo, x, o.mean, o.var, o.std);
            }
            if (F64 && WANT_VAR) {
                unsigned long long todo = __ballot(redo);               // (every lane of the wave is here)
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    double mean, var;
                    walk_exact_window<R, Shape>(g, yo, xw + src, lane, mean, var);
                    if (lane == src) {
                        const long off = yo * g.ld_out + x;
                        if (o.mean) o.mean[off] = (float)mean;
                        if (o.var) o.var[off] = (float)var;
                        if (o.std) o.std[off] = (float)sqrt(var);
                    }
                }
            }
            if (F32) a32.shift();
            if (F64) a64.shift();
        }
    };
    if (xw >= R && xw + 64 + R <= g.cols) walk(std::false_type{});     // interior wave: no column predicates
    else walk(std::true_type{});
}

template <int R, typename Shape, bool F32, bool WANT_SUM, bool WANT_MM, bool F64, bool WANT_VAR = true>
__device__ __forceinline__ void walk_tile(const WalkGeom &g, const WalkOuts &o) {
    const long t = xcd_tile(blockIdx.x, g.n_tiles, XCD_UNIT(XRS_XCD_WALK, g.tiles_x));
    if (t < 0) return;
    const long ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long xw = tx * 256 + wv * 64;                 // first column of this wave (scalar)
    const long y0 = ty * CTH;
    const long y_end = (y0 + CTH < g.rows ? y0 + CTH : g.rows);        // output rows [y0, y_end)
    walk_columns<R, Shape, F32, WANT_SUM, WANT_MM, F64, WANT_VAR>(g, o, xw, lane, y0, y_end);
}

inline int walk_grid(WalkGeom &g, long *grid) {
    g.tiles_x = (g.cols + 255) / 256;
    g.n_tiles = g.tiles_x * ((g.rows + CTH - 1) / CTH);
    *grid = xcd_grid(g.n_tiles, XCD_UNIT(XRS_XCD_WALK, g.tiles_x));
    if (*grid > 0x7fffffffL) return fail("focal circle: raster too large for one launch");
    return 0;
}

}  // namespace xrs
