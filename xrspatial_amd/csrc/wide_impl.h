// Focal mean (and the window sum) over large circular / box masks -- focal.apply(raster, circle_kernel(...)) and
// focal_stats(..., ['mean']) with 7x7 .. 25x25 windows (xrspatial/focal.py:305-326 with _calc_mean :226-228; the
// reference gathers the window per cell and calls numba's nanmean: float64 sum / count, float32 store).
//
// The "wide" row walker: ONE 16-byte load per lane per input row, neighbours through LDS, float32 arithmetic on
// shifted values, a register ring with static indices.
//   * a wave owns a tile of 256 columns x WTH output rows and walks DOWN its input rows; a lane owns 4 adjacent
//     columns (one global_load_dwordx4 per row; the 2*HL halo columns come from one extra dword load of the first
//     2*HL lanes).  Rows are prefetched PFN rows ahead into registers.
//   * the row (minus a wave-uniform shift c, the cell at the tile centre) goes to LDS once (ds_write_b128) and every
//     lane reads back the 4 + 2*HL consecutive cells its four windows cover as aligned ds_read_b128 (conflict free).
//   * a lane-local prefix sum over those cells turns every centred run of the mask into ONE subtraction,
//     S_h(x) = P[x + h] - P[x - h - 1]; a circle of radius 12 has only 9 distinct half-widths.
//   * the 2R+1 output rows in flight live in a register ring, acc[(row - dy) mod (2R+1)]; the row loop is unrolled
//     2R+1 times so that every ring index is a compile-time constant: no register moves, 25 adds per cell and row.
//   * mean = c + S / n.  Everything is float32: the error of S is bounded by u * A * (a few 10^4) with A the largest
//     |v - c| of the tile, i.e. <= 7e-6 * A on the mean; the wave checks A <= 1.4 * min |mean| at the end of its tile
//     (which guarantees 1e-5 relative) and that every result is finite, and otherwise hands the whole tile to the
//     float64 column walker of circle_walk.h (NaN-skipping, counting, exact) -- nodata regions, +-inf, rasters whose
//     values straddle zero take that path.  Typical errors are ~1e-8 relative (tests: 1e-6 on both DEMs).
//   * raster edges (clipped windows): out-of-raster cells enter as d = 0 and the divisor is the geometric count of
//     in-raster cells, so edge tiles stay on the fast path.
// Included by kxk_wide_circle.hip and kxk_wide_box.hip, which define XRS_WIDE_SHAPE / XRS_WIDE_ENTRY.
// vs the one-column walker this replaces for `mean`: 25 dword loads + ~270 VALU instructions per cell and row ->
// 0.3 loads + ~50.  HBM-bound by construction (8 B per cell); measured numbers in DESIGN.md.
#include "circle_walk.h"

#include <utility>

using namespace xrs;

namespace {

constexpr int WTH = 128;          // output rows per wave tile

struct WideArgs {
    WalkGeom g;                   // in, rows, cols, ld_in, ld_out, halo_top, halo_bot (tiles_x / n_tiles: wave tiles)
    float *out_mean, *out_sum;    // either may be NULL
    long n_groups;                // workgroups = groups of 4 horizontally adjacent wave tiles
    long groups_x;
};

template <int R, typename Shape>
struct WideCfg {
    static constexpr int K = 2 * R + 1;
    static constexpr int HL = 4 * ((R + 3) / 4);          // halo columns each side, rounded up to whole float4s
    static constexpr int NV = 4 + 2 * HL;                  // cells a lane reads back per row
    static constexpr int NQ = NV / 4;
    static constexpr int STG = 256 + 2 * HL;               // staged cells per row
    static constexpr int NTAPS = shape_taps<Shape>(R);
    static constexpr int PFN = (K % 5 == 0) ? 5 : (K % 3 == 0) ? 3 : (K == 7 ? 7 : 3);    // rows prefetched
    static constexpr bool ROT = (K % PFN == 0);            // prefetch slots addressed by the (static) phase
    static constexpr bool level_used(int h) {
        for (int dy = 0; dy <= R; ++dy)
            if (Shape::hw(R, dy) == h) return true;
        return false;
    }
};

// number of in-raster cells under the window centred on (yo, x): rows [y_lo, y_hi), columns [0, cols)
template <int R, typename Shape>
__device__ __forceinline__ int clipped_count(long yo, long x, long y_lo, long y_hi, long cols) {
    int n = 0;
    for (int dy = -R; dy <= R; ++dy) {
        const long yr = yo + dy;
        if (yr < y_lo || yr >= y_hi) continue;
        const int h = Shape::hw(R, dy < 0 ? -dy : dy);
        const long a = x - h < 0 ? 0 : x - h, b = x + h > cols - 1 ? cols - 1 : x + h;
        n += (int)(b - a + 1);
    }
    return n;
}

template <int R, typename Shape, bool EDGE>
struct WideWalk {
    using C = WideCfg<R, Shape>;
    static constexpr int K = C::K, HL = C::HL, NV = C::NV, NQ = C::NQ, PFN = C::PFN;

    // ---- state
    float acc[K][4];
    xrs_f4u pf_own[PFN];
    float pf_halo[PFN];
    float amax, mmin;
    bool bad;
    int i;                         // input row counter: row y_first + i
    // ---- constants of the tile
    const WalkGeom &g;
    float *out_mean, *out_sum;
    float *lds;                    // this wave's row buffer (STG floats)
    long x_tile, y0, y_end, y_first;
    int n_in, lane;
    float c;                       // the shift
    float n_full[4];               // EDGE: cell count of a window whose rows are all inside, per owned column

    __device__ __forceinline__ WideWalk(const WalkGeom &g_, float *om, float *os, float *lds_, long xt, long y0_, long ye, int lane_)
        : g(g_), out_mean(om), out_sum(os), lds(lds_), x_tile(xt), y0(y0_), y_end(ye), lane(lane_) {}

    __device__ __forceinline__ void load_row(int il, xrs_f4u &own, float &halo) const {
        // staged cell s <-> raster column x_tile - HL + s; lane owns s = 4*lane .. 4*lane+3, lanes < 2*HL also cell 256+lane
        const long yy = y_first + il;
        const long xs = x_tile - HL + 4 * lane;
        if (!EDGE) {
            const float *p = g.in + yy * g.ld_in + (x_tile - HL);
            own = load_f4u(p + 4 * lane);
            halo = c;
            if (lane < 2 * HL) halo = p[256 + lane];
            return;
        }
        own.x = own.y = own.z = own.w = c;                   // out-of-raster cells: d = 0 after the shift
        halo = c;
        const bool row_ok = il < n_in && yy >= -(long)g.halo_top && yy < g.rows + g.halo_bot;     // wave-uniform
        if (!row_ok) return;
        const float *p = g.in + yy * g.ld_in;
        if (xs >= 0 && xs + 4 <= g.cols) {
            own = load_f4u(p + xs);
        } else {
            if (xs >= 0 && xs < g.cols) own.x = p[xs];
            if (xs + 1 >= 0 && xs + 1 < g.cols) own.y = p[xs + 1];
            if (xs + 2 >= 0 && xs + 2 < g.cols) own.z = p[xs + 2];
            if (xs + 3 >= 0 && xs + 3 < g.cols) own.w = p[xs + 3];
        }
        const long xh = x_tile - HL + 256 + lane;
        if (lane < 2 * HL && xh >= 0 && xh < g.cols) halo = p[xh];
    }

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[j][o] = 0.0f;
        amax = 0.0f;
        mmin = INFINITY;
        bad = false;
        i = 0;
        y_first = y0 - R;
        n_in = (int)(y_end - y0) + 2 * R;
        // shift: the cell at the tile centre (any finite value works; a nearby one keeps |v - c| small)
        const long yc = y0 + (y_end - y0) / 2, xc = (x_tile + 128 < g.cols ? x_tile + 128 : g.cols - 1);
        const float c0 = g.in[yc * g.ld_in + xc];
        c = isfinite(c0) ? c0 : 0.0f;
        if (EDGE) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
                n_full[o] = (float)clipped_count<R, Shape>(0, x_tile + 4 * lane + o, -(long)R, (long)R + 1, g.cols);
        }
#pragma unroll
        for (int s = 0; s < PFN; ++s) load_row(s, pf_own[s], pf_halo[s]);
    }

    // One input row.  No exits inside a round of K steps (a step past the last row is skipped as a whole): with early
    // returns the compiler sinks the ring updates of all K phases into the loop latch and spills their operands.
    template <int PHASE>
    __device__ __forceinline__ void step() {
        if (i < n_in) step_body<PHASE>();
        ++i;
    }

    template <int PHASE>
    __device__ __forceinline__ void step_body() {
        constexpr int SLOT = C::ROT ? PHASE % PFN : 0;
        const xrs_f4u q = pf_own[SLOT];
        const float hq = pf_halo[SLOT];
        if (!C::ROT) {
#pragma unroll
            for (int s = 0; s + 1 < PFN; ++s) { pf_own[s] = pf_own[s + 1]; pf_halo[s] = pf_halo[s + 1]; }
        }
        constexpr int REFILL = C::ROT ? SLOT : PFN - 1;
        if (EDGE || i + PFN < n_in) load_row(i + PFN, pf_own[REFILL], pf_halo[REFILL]);      // (EDGE: load_row tests the row itself)

        const long yy = y_first + i;
        const bool row_in = !EDGE || (yy >= -(long)g.halo_top && yy < g.rows + g.halo_bot);  // wave-uniform
        if (row_in) {
            // ---- shifted row -> LDS, each lane reads back the NV cells under its four windows
            const float d0 = q.x - c, d1 = q.y - c, d2 = q.z - c, d3 = q.w - c, dh = hq - c;
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(d0), fabsf(d1)), fmaxf(fabsf(d2), fabsf(d3))));
            amax = fmaxf(amax, fabsf(dh));                   // (lanes >= 2*HL: hq = c, dh = 0)
            bad |= !(isfinite(d0 + d1) && isfinite(d2 + d3) && isfinite(dh));
            float *row = lds;      // ONE row buffer: LDS serves a wave's instructions in order, so the next row's writes
                                   // (issued after this row's reads) cannot overtake them
            *reinterpret_cast<float4 *>(row + 4 * lane) = make_float4(d0, d1, d2, d3);
            if (lane < 2 * HL) row[256 + lane] = dh;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                 // (LDS serves one wave's instructions in order)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float w[NV];
#pragma unroll
            for (int b = 0; b < NQ; ++b) {
                const float4 t = *reinterpret_cast<const float4 *>(row + 4 * lane + 4 * b);
                w[4 * b] = t.x; w[4 * b + 1] = t.y; w[4 * b + 2] = t.z; w[4 * b + 3] = t.w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- lane-local prefix sums: P[k] = w[0] + .. + w[k]; cell o's centre is w[HL + o]
#pragma unroll
            for (int k = 1; k < NV; ++k) w[k] += w[k - 1];
            // ---- every distinct half-width once, into the ring slots of the output rows that see this row with it
#pragma unroll
            for (int h = 0; h <= R; ++h) {
                if (!C::level_used(h)) continue;
                float S[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int hi = HL + o + h, lo = HL + o - h - 1;
                    S[o] = lo >= 0 ? w[hi] - w[lo] : w[hi];
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int dy = j - R;
                    if (Shape::hw(R, dy < 0 ? -dy : dy) != h) continue;
                    const int idx = ((PHASE - dy) % K + K) % K;
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[idx][o] += S[o];
                }
            }
        }
        // ---- the output row R rows up is complete
        constexpr int DONE = ((PHASE - R) % K + K) % K;
        if (i >= 2 * R) {
            const long yo = y0 + (i - 2 * R);
            const long xo = x_tile + 4 * lane;
            float m[4], sm[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float n = (float)C::NTAPS;
                if (EDGE) {
                    const bool rows_in = yo - R >= -(long)g.halo_top && yo + R < g.rows + g.halo_bot;   // wave-uniform
                    n = rows_in ? n_full[o]
                                : (float)clipped_count<R, Shape>(yo, xo + o, -(long)g.halo_top, g.rows + g.halo_bot, g.cols);
                }
                const float s = acc[DONE][o];
                m[o] = EDGE ? c + s / n : fmaf(s, 1.0f / (float)C::NTAPS, c);
                sm[o] = fmaf(n, c, s);
                bad |= !isfinite(s);
                if (!EDGE || xo + o < g.cols) mmin = fminf(mmin, fabsf(m[o]));
            }
            if (!EDGE) {
                if (out_mean) store_f4u(out_mean + yo * g.ld_out + xo, m[0], m[1], m[2], m[3]);
                if (out_sum) store_f4u(out_sum + yo * g.ld_out + xo, sm[0], sm[1], sm[2], sm[3]);
            } else {
                const int nown = g.cols - xo >= 4 ? 4 : (g.cols - xo > 0 ? (int)(g.cols - xo) : 0);
                if (nown > 0) {
                    if (out_mean) store_cols4(out_mean + yo * g.ld_out + xo, m, nown);
                    if (out_sum) store_cols4(out_sum + yo * g.ld_out + xo, sm, nown);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[DONE][o] = 0.0f;
    }

    static __device__ __forceinline__ void store_cols4(float *p, const float (&v)[4], int n) {
        if (n >= 4) { store_f4u(p, v[0], v[1], v[2], v[3]); return; }
        p[0] = v[0];
        if (n > 1) p[1] = v[1];
        if (n > 2) p[2] = v[2];
    }

    template <int... P>
    __device__ __forceinline__ void round(std::integer_sequence<int, P...>) {
        (step<P>(), ...);
    }

    // true: every result of the tile is good; false: the caller redoes the tile with the float64 walker
    __device__ __forceinline__ bool run() {
        init();
        while (i < n_in) {
            round(std::make_integer_sequence<int, K>{});
            if (__any(bad)) return false;                    // a non-finite cell: stop early
        }
        // error bound of the float32 sums (header): |delta mean| <= u * A * (K * (2 * NV^2 + K) + K * NTAPS) / n
        float a = amax, mm = mmin;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            a = fmaxf(a, __shfl_xor(a, sft));
            mm = fminf(mm, __shfl_xor(mm, sft));
        }
        constexpr float U = 5.9604645e-8f;
        constexpr float COEF = U * (float)(K * (2 * NV * NV + K) + K * C::NTAPS) / (float)C::NTAPS * (EDGE ? 4.0f : 1.0f);
        const bool ok = !__any(bad) && (COEF * a <= 0.9e-5f * mm);
        return ok;
    }
};

template <int R, typename Shape>
__global__ void __launch_bounds__(256, 2) focal_wide_kernel(const WideArgs a) {
    using C = WideCfg<R, Shape>;
    __shared__ __attribute__((aligned(16))) float lds_rows[4][C::STG];
    const long gidx = xcd_tile(blockIdx.x, a.n_groups);
    if (gidx < 0) return;
    const long ty = gidx / a.groups_x, gx = gidx - ty * a.groups_x;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long x_tile = (gx * 4 + wv) * 256;
    const long y0 = ty * WTH;
    const WalkGeom &g = a.g;
    if (x_tile >= g.cols) return;
    const long y_end = y0 + WTH < g.rows ? y0 + WTH : g.rows;
    const bool interior = x_tile - C::HL >= 0 && x_tile + 256 + C::HL <= g.cols && y0 - R >= -(long)g.halo_top &&
                          y_end + R <= g.rows + g.halo_bot;
    bool ok;
    if (interior) {
        WideWalk<R, Shape, false> w(g, a.out_mean, a.out_sum, lds_rows[wv], x_tile, y0, y_end, lane);
        ok = w.run();
    } else {
        WideWalk<R, Shape, true> w(g, a.out_mean, a.out_sum, lds_rows[wv], x_tile, y0, y_end, lane);
        ok = w.run();
    }
    if (ok) return;
    // non-finite cells under a window, or sums too ill-conditioned for float32: the float64 column walker (NaN-skipping,
    // counting; mean from float64 sums, the sum with the reference's sequential float32 adds), 64 columns at a time
    const WalkOuts o = {a.out_sum, nullptr, nullptr, nullptr, a.out_mean, nullptr, nullptr};
    for (int q = 0; q < 4; ++q) {
        if (a.out_mean) walk_columns<R, Shape, false, false, false, true, false>(g, o, x_tile + 64 * q, lane, y0, y_end);
        if (a.out_sum) walk_columns<R, Shape, true, true, false, false, false>(g, o, x_tile + 64 * q, lane, y0, y_end);
    }
}

template <int R, typename Shape>
int launch_wide(WideArgs &a, hipStream_t s) {
    WalkGeom &g = a.g;
    g.tiles_x = (g.cols + 255) / 256;
    const long tiles_y = (g.rows + WTH - 1) / WTH;
    g.n_tiles = g.tiles_x * tiles_y;
    a.groups_x = (g.tiles_x + 3) / 4;
    a.n_groups = a.groups_x * tiles_y;
    const long grid = xcd_grid(a.n_groups);
    if (grid > 0x7fffffffL) return fail("focal mean: raster too large for one launch");
    hipLaunchKernelGGL((focal_wide_kernel<R, Shape>), dim3((unsigned)grid), dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

int dispatch_wide(WideArgs &a, const double *kernel, int r, hipStream_t s) {
    switch (r) {
#define XRS_WIDE_CASE(RR) case RR: return is_shape<RR, XRS_WIDE_SHAPE>(kernel) ? launch_wide<RR, XRS_WIDE_SHAPE>(a, s) : -1;
#ifndef XRS_WIDE_PROBE
        XRS_WIDE_CASE(3) XRS_WIDE_CASE(4) XRS_WIDE_CASE(5) XRS_WIDE_CASE(6) XRS_WIDE_CASE(7) XRS_WIDE_CASE(8)
        XRS_WIDE_CASE(9) XRS_WIDE_CASE(10) XRS_WIDE_CASE(11)
#endif
        XRS_WIDE_CASE(12)
#undef XRS_WIDE_CASE
        default: return -1;
    }
}

}  // namespace

namespace xrs {

// 0 = launched, -1 = not this shape with a radius of 3..12 cells (caller takes another kernel), > 0 = error.
int XRS_WIDE_ENTRY(const float *in, float *out_mean, float *out_sum, long rows, long cols, long ld_in, long ld_out,
                   const double *kernel, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    if (!out_mean && !out_sum) return 0;
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.g.in = in; a.g.rows = rows; a.g.cols = cols; a.g.ld_in = ld_in; a.g.ld_out = ld_out;
    a.g.halo_top = halo_top; a.g.halo_bot = halo_bot;
    a.out_mean = out_mean; a.out_sum = out_sum;
    return dispatch_wide(a, kernel, krows / 2, s);
}

}  // namespace xrs
