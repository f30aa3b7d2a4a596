// All seven focal statistics over SMALL circular / box masks (5x5, 7x7) in one pass -- focal_stats(agg, circle_kernel(1, 1, 2))
// with its default statistics, BASELINE configs[2] (xrspatial/focal.py:782-797 runs one apply() pass per statistic, each
// gathering the window per cell for a numba reducer, :226-258: nanmean / nanvar / nanstd with float64 accumulators, nansum
// in the array dtype, nanmax / nanmin).
//
// Round 1's column walker (circle_walk.h: one column per lane, 2R+1 dependent loads per row, float64 moments tap by tap,
// 256-byte output rows) spends 838 M wave instructions on a 16384^2 raster and writes its seven planes in 256-byte pieces
// (experiments/write_pattern.hip prices that geometry alone at 1.71 ms against 1.40 for 1 KiB rows).  Here:
//   * a wave owns 64 NC columns x ~126 output rows and walks DOWN; a lane owns NC adjacent columns (NC = 4 for 5x5: every
//     plane leaves as one 16-byte store per lane = 1 KiB per wave and row).  Rows arrive in a private LDS ring by LDS-DMA,
//     D rows ahead (lds_dma.h; the scheme of mom_impl.h); a lane reads the NC + 2R cells under its windows back;
//   * per row and column the running sums / extrema over CENTRED RUNS grow from the centre outwards -- level h costs two
//     additions per moment and one v_min3 / v_max3 -- and every distinct half-width of the mask is then what the output
//     rows that see this row with it need: the circular 5x5 has three row patterns, so 4 + 4 additions and 2 + 2 extrema
//     per cell and row instead of 13 taps four times over;
//   * the 2R+1 output rows in flight live in a register ring with compile-time indices (the loop is unrolled over one turn
//     of the ring); a slot starts from its first row's level instead of 0 / +-inf;
//   * moments are FLOAT64 sums of d = v - c and d^2 (c = the lane's own cell at the tile's middle row; d is exact, so the
//     result does not depend on how the raster is cut into tiles): mean = c + S / n, var = (Q - S^2 / n) / n.  A window of
//     equal cells is recognised by max == min -- the extrema are there anyway -- and gets mean = the value, var = 0 exactly;
//     a window whose variance drowns in the cancellation (Q / (n var) > 2^30) is recomputed by the whole wave with the
//     reference's two passes (walk_exact_window);
//   * NaN cells: the fast body does not look for them; a non-finite sum sends the TILE through the NaN-aware body (validity
//     decided once per cell, counts carried in a third ring, v_min / v_max skip NaN by themselves, an empty window gives NaN
//     and sum 0, +-inf flows through the sums like in the reference); tiles at the raster / shard edge run that body with
//     predicated loads (cells outside = NaN, so clipped windows need nothing else).
// ~65 wave instructions per 64 cells instead of ~200.  Included by kxk_sw_circle.hip / kxk_sw_box.hip (XRS_SW_SHAPE, XRS_SW_ENTRY).
#include "circle_walk.h"
#include "lds_dma.h"
#include "strip.h"

#include <utility>

using namespace xrs;

namespace {

struct SwArgs {
    WalkGeom g;                   // in, rows, cols, ld_in, ld_out, halo_top, halo_bot
    float *out[XRS_NUM_STATS];    // XRS_STAT_* order; any may be NULL
    long groups_x, tiles_y;       // workgroups = 4 horizontally adjacent wave tiles
    int tile_rows;                // output rows per tile (tile_rows + 2R input rows = whole turns of the ring)
    int rim_first;
};

template <int R, typename Shape, int NC>
struct SwCfg {
    static constexpr int K = 2 * R + 1;
    static constexpr int TW = 64 * NC;                     // columns per wave tile
    static constexpr int HS = 4;                           // staged halo columns each side (16-byte aligned rows)
    static_assert(R <= HS, "halo");
    static constexpr int NV = NC + 2 * R;                  // cells a lane reads back per row
    static constexpr int CELLS = TW + 2 * HS;              // staged cells per row
    static constexpr int RBF = CELLS <= 256 ? 256 : 320;   // floats per ring row (a 16-byte DMA writes a whole KiB, the dword one 256 B more)
    static constexpr int NDMA = CELLS <= 256 ? 1 : 2;
#ifndef XRS_SW_D
#define XRS_SW_D 6
#endif
    static constexpr int D = XRS_SW_D;                     // rows in flight by LDS-DMA; D + 1 row buffers per wave
    static constexpr int NTAPS = shape_taps<Shape>(R);
    static constexpr int nin(int base) { return ((base + 2 * R + K - 1) / K) * K; }
    static constexpr bool level_used(int h) {
        for (int dy = 0; dy <= R; ++dy)
            if (Shape::hw(R, dy) == h) return true;
        return false;
    }
};

__device__ __forceinline__ float sw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float sw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float sw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

typedef float sw_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sw_store_row(float *sbase, unsigned voff, const float (&v)[4]) {
    sw_v4f q; q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
    // (s_nop: a VALU write to the data registers of a store of more than 8 bytes needs a wait state after it; the compiler
    //  inserts those for its own stores, not for one it cannot see inside an asm statement -- without it the `range` plane
    //  came out with the NEXT computation's values in two of four columns)
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(voff), "v"(q), "s"(sbase) : "memory");
}
__device__ __forceinline__ void sw_store_row(float *sbase, unsigned voff, const float (&v)[2]) {
    lds_dma_v2f q; q[0] = v[0]; q[1] = v[1];
    st_row_nt(sbase, voff, q);
}

// NANS = false: the fast body (every cell under the tile's windows finite, or the tile is redone); NANS = true: validity
// per cell, counts in a third ring.  EDGE (implies NANS): predicated global loads instead of the DMA ring, cells outside
// the raster / the shard's halo rows are NaN, partial tiles.  NO: planes written per row AT LEAST (vmcnt bookkeeping).
template <int R, typename Shape, int NC, bool NANS, bool EDGE, int NO>
struct SwWalk {
    using C = SwCfg<R, Shape, NC>;
    static constexpr int K = C::K, NV = C::NV, D = C::D;
    static_assert(!EDGE || NANS, "edge tiles count their cells");

    double S[K][NC], Q[K][NC];
    float mn[K][NC], mx[K][NC];
    float cn[NANS ? K : 1][NC];
    double c;                      // the lane's shift
    unsigned long long badm;       // FAST: lanes that met a non-finite sum (wave-uniform)
    unsigned illp;                 // per row of the round (4 bits each): bit o set = output o of this lane needs the exact path
    unsigned long long illm;       // any lane, any row of the round (wave-uniform)
    int slot_in, slot_out, t, n_in;
    const float *dma_src;          // interior: (wave-uniform) first staged cell of the next row to DMA
    int dma_adv;
    long out_off;                  // interior: offset of the wave tile's next output row in every plane
    unsigned ring_addr;

    const SwArgs &a;
    const WalkGeom &g;
    float *lds;
    long x_tile, y0, y_end, y_first;
    int lane;

    __device__ __forceinline__ SwWalk(const SwArgs &a_, float *lds_, long xt, long y0_, long ye, int lane_)
        : a(a_), g(a_.g), lds(lds_), x_tile(xt), y0(y0_), y_end(ye), lane(lane_) {}

    // interior: input row il (clamped past the tile) -> ring slot; staged cell s <-> raster column x_tile - HS + s
    // (rows in order: the source pointer advances by a row per call, dma_adv more times, instead of a 64-bit multiply per row)
    __device__ __forceinline__ void dma_row(int slot) {
        const float *p = uniform_ptr(dma_src);
        dma_src += dma_adv > 0 ? g.ld_in : 0;
        --dma_adv;
        const unsigned dst = ring_addr + (unsigned)slot * (C::RBF * 4);
        static_assert(C::CELLS % 4 == 0, "whole 16-byte pieces");
        constexpr int QMAX = (C::CELLS < 256 ? C::CELLS : 256) / 4 - 1;
        glds16_s(p, 16u * (unsigned)(lane < QMAX ? lane : QMAX), dst);
        if (C::CELLS > 256) glds4_s(p, 4u * (unsigned)(256 + (lane < C::CELLS - 257 ? lane : C::CELLS - 257)), dst + 1024);
    }

    // the NV cells under the lane's windows in input row il: v[k] <-> raster column x_tile + NC lane - R + k
    __device__ __forceinline__ void row_cells(int il, float (&v)[NV]) {
        if (EDGE) {
            const long yy = y_first + il;
            const float qnan = nan_f32();
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = qnan;
            if (il >= n_in || yy < -(long)g.halo_top || yy >= g.rows + g.halo_bot) return;      // wave-uniform
            const float *p = g.in + yy * g.ld_in;
            const long x = x_tile + NC * lane - R;
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (x + k >= 0 && x + k < g.cols) v[k] = p[x + k];
            return;
        }
        // ring slot: staged cell index of v[0] = HS - R + NC lane (8-byte aligned for R = 2, NC = 4: float2 reads)
        typedef float lds2 __attribute__((ext_vector_type(2)));
        const float *row = lds + slot_out * C::RBF + (C::HS - R) + NC * lane;
        if ((C::HS - R) % 2 == 0 && NC % 2 == 0) {
#pragma unroll
            for (int k = 0; k + 1 < NV; k += 2) {
                const lds2 q = *reinterpret_cast<const lds2 *>(row + k);
                v[k] = q[0]; v[k + 1] = q[1];
            }
            if (NV & 1) v[NV - 1] = row[NV - 1];
        } else {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = row[k];
        }
    }

    __device__ __forceinline__ void init() {
        badm = 0;
        t = 0;
        y_first = y0 - R;
        n_in = EDGE ? (int)(y_end - y0) + 2 * R : C::nin((int)(y_end - y0));       // (interior tiles: whole turns of the ring)
        // the shift: the lane's own first column at the tile's middle row (any finite value works; a near one keeps d small)
        const long yc = y0 + (y_end - y0) / 2;
        long xc = x_tile + NC * lane;
        xc = xc < g.cols ? xc : g.cols - 1;
        const float v = g.in[yc * g.ld_in + xc];
        // (a lane on nodata borrows a neighbour's value: a shift of 0 would put the data's level into every d)
        const unsigned long long have = __builtin_amdgcn_ballot_w64(isfinite(v));
        const float v_any = __shfl(v, have ? __ffsll((long long)have) - 1 : 0);
        c = (double)(isfinite(v) ? v : have ? v_any : 0.0f);
        if (!EDGE) {
            ring_addr = lds_addr(lds);
            dma_src = uniform_ptr(g.in + y_first * g.ld_in + (x_tile - C::HS));
            dma_adv = n_in - 1;
            out_off = y0 * g.ld_out + x_tile;
            for (int r = 0; r < D; ++r) dma_row(r);
            slot_in = D;
            slot_out = 0;
        }
    }

    template <int PHASE>
    __device__ __forceinline__ void step() {
        const int i = t + PHASE;
        if (EDGE && i >= n_in) return;
        if (!EDGE) {
            dma_row(slot_in);
            slot_in = slot_in + 1 == D + 1 ? 0 : slot_in + 1;
            // row i was issued D steps ago; younger: D rows of DMAs and -- once the walk emits -- the stores of D steps
            if (i >= 2 * R + D) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D * (C::NDMA + NO)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D * C::NDMA) : "memory");
        }
        float v[NV];
        row_cells(i, v);
        if (!EDGE) slot_out = slot_out + 1 == D + 1 ? 0 : slot_out + 1;

        // ---- shifted values (NaN cells: d = 0, not counted)
        double d[NV], q[NV];
        float f[NANS ? NV : 1];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (NANS) {
                const bool ok = v[k] == v[k];
                f[k] = ok ? 1.0f : 0.0f;
                d[k] = ok ? (double)v[k] - c : 0.0;
            } else {
                d[k] = (double)v[k] - c;
            }
            q[k] = d[k] * d[k];
        }
        // ---- per owned column: sums / extrema over the centred runs, from the centre outwards; every used level into the
        // ring slots of the output rows that see this row with it
#pragma unroll
        for (int o = 0; o < NC; ++o) {
            const int ci = R + o;
            double s = d[ci], qq = q[ci];
            float lo = v[ci], hi = v[ci], cnt = NANS ? f[ci] : 0.0f;
#pragma unroll
            for (int h = 0; h <= R; ++h) {
                if (h > 0) {
                    s += d[ci - h] + d[ci + h];
                    qq += q[ci - h] + q[ci + h];
                    lo = sw_min3(lo, v[ci - h], v[ci + h]);
                    hi = sw_max3(hi, v[ci - h], v[ci + h]);
                    if (NANS) cnt += f[ci - h] + f[ci + h];
                }
                if (!C::level_used(h)) continue;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int dy = j - R;                              // this row is at offset dy of output row i - dy
                    if (Shape::hw(R, dy < 0 ? -dy : dy) != h) continue;
                    const int idx = ((PHASE - dy) % K + K) % K;
                    if (dy == -R) {                                    // a new output row: its first contribution
                        S[idx][o] = s; Q[idx][o] = qq; mn[idx][o] = lo; mx[idx][o] = hi;
                        if (NANS) cn[idx][o] = cnt;
                    } else {
                        S[idx][o] += s; Q[idx][o] += qq;
                        mn[idx][o] = sw_min(mn[idx][o], lo);
                        mx[idx][o] = sw_max(mx[idx][o], hi);
                        if (NANS) cn[idx][o] += cnt;
                    }
                }
            }
        }
        // ---- the output row R rows up is complete
        if (i < 2 * R) return;
        constexpr int DONE = ((PHASE - R) % K + K) % K;
        const long yo = y0 + (i - 2 * R);
        if (EDGE && yo >= y_end) return;
        const long xo = x_tile + NC * lane;
        // (interior tiles emit their rows in order: the offset advances by a row per emitted row instead of a 64-bit multiply)
        long rowoff;
        if (EDGE) rowoff = yo * g.ld_out + x_tile;
        else { rowoff = out_off; out_off += g.ld_out; }
        float r_mean[NC], r_var[NC], r_std[NC], r_sum[NC], r_range[NC], r_max[NC], r_min[NC];
        unsigned bits = 0;
#pragma unroll
        for (int o = 0; o < NC; ++o) {
            const double s = S[DONE][o], qq = Q[DONE][o];
            double n = (double)C::NTAPS, inv = 1.0 / (double)C::NTAPS;
            if (NANS) {
                n = (double)cn[DONE][o];
                inv = rcp_count((int)cn[DONE][o]);                    // (no valid cell: NaN)
            }
            const double ms = s * inv;
            const double e = fma(-s, ms, qq);                          // n * variance
            const float lo = mn[DONE][o], hi = mx[DONE][o];
            // every valid cell the same finite value (NaN extrema compare false; an all-inf window goes the exact way: NaN variance)
            const bool flat = lo == hi && fabsf(lo) < INFINITY;
            const bool live = !EDGE || xo + o < g.cols;
            const bool fin = fabs(s) < INFINITY;
            if (!NANS) badm |= __builtin_amdgcn_ballot_w64(!fin);      // a NaN / inf cell: the tile is redone
            // ill-conditioned (or +-inf under the window: e is NaN): the reference's two passes by the whole wave, after the round
            const bool has = NANS ? cn[DONE][o] > 0.0f : fin;
            if (live && has && !flat && !(e >= 0x1p-30 * qq)) bits |= 1u << o;
            const float var = (float)(e * inv);
            r_mean[o] = flat ? lo : (float)(c + ms);
            r_var[o] = flat ? 0.0f : var;
            r_std[o] = flat ? 0.0f : __builtin_amdgcn_sqrtf(var);
            r_sum[o] = NANS && cn[DONE][o] == 0.0f ? 0.0f : (float)fma(n, c, s);
            r_range[o] = hi - lo;
            r_max[o] = hi;
            r_min[o] = lo;
        }
        illp |= bits << (4 * PHASE);
        illm |= __builtin_amdgcn_ballot_w64(bits != 0);
        float *const *out = a.out;
        if (!EDGE) {
            const unsigned lane_b = (unsigned)(NC * 4) * (unsigned)lane;
            // (NO == all seven: every plane is there, no test per plane and row.  All seven stores in ONE asm statement -- one
            //  wait state instead of seven -- measured 1-3 % slower: every result then has to be ready before the first store)
            constexpr bool ALL = NO == XRS_NUM_STATS;
            if (ALL || out[XRS_STAT_MEAN]) sw_store_row(uniform_ptr(out[XRS_STAT_MEAN] + rowoff), lane_b, r_mean);
            if (ALL || out[XRS_STAT_MAX]) sw_store_row(uniform_ptr(out[XRS_STAT_MAX] + rowoff), lane_b, r_max);
            if (ALL || out[XRS_STAT_MIN]) sw_store_row(uniform_ptr(out[XRS_STAT_MIN] + rowoff), lane_b, r_min);
            if (ALL || out[XRS_STAT_RANGE]) sw_store_row(uniform_ptr(out[XRS_STAT_RANGE] + rowoff), lane_b, r_range);
            if (ALL || out[XRS_STAT_STD]) sw_store_row(uniform_ptr(out[XRS_STAT_STD] + rowoff), lane_b, r_std);
            if (ALL || out[XRS_STAT_VAR]) sw_store_row(uniform_ptr(out[XRS_STAT_VAR] + rowoff), lane_b, r_var);
            if (ALL || out[XRS_STAT_SUM]) sw_store_row(uniform_ptr(out[XRS_STAT_SUM] + rowoff), lane_b, r_sum);
        } else {
#pragma unroll
            for (int o = 0; o < NC; ++o) {
                if (xo + o >= g.cols) break;
                const long off = rowoff + NC * lane + o;
                if (out[XRS_STAT_MEAN]) out[XRS_STAT_MEAN][off] = r_mean[o];
                if (out[XRS_STAT_MAX]) out[XRS_STAT_MAX][off] = r_max[o];
                if (out[XRS_STAT_MIN]) out[XRS_STAT_MIN][off] = r_min[o];
                if (out[XRS_STAT_RANGE]) out[XRS_STAT_RANGE][off] = r_range[o];
                if (out[XRS_STAT_STD]) out[XRS_STAT_STD][off] = r_std[o];
                if (out[XRS_STAT_VAR]) out[XRS_STAT_VAR][off] = r_var[o];
                if (out[XRS_STAT_SUM]) out[XRS_STAT_SUM][off] = r_sum[o];
            }
        }
    }

    template <int... P>
    __device__ __forceinline__ void round(std::integer_sequence<int, P...>) {
        illm = 0;
        illp = 0;
        (step<P>(), ...);
        // The windows of the round that need the exact path (circle_walk.h's two passes): one at a time, the whole wave on each,
        // results over the ones the round stored (same wave, same addresses, later stores).  Here, once per round and NOT
        // unrolled: inlined into every phase and column of the unrolled walk the exact path made the round loop a 60 KB body
        // (two CUs share 64 KB of instruction cache).
        if (illm && !(!NANS && badm)) {
#pragma nounroll
            for (int ph = 0; ph < K; ++ph) {
                const int i = t + ph;
                const long yo = y0 + (i - 2 * R);
                const unsigned bits = illp >> (4 * ph) & 15u;
                if (i < 2 * R || !__builtin_amdgcn_ballot_w64(bits != 0)) continue;
#pragma nounroll
                for (int o = 0; o < NC; ++o) {
                    unsigned long long m = __builtin_amdgcn_ballot_w64((bits >> o & 1u) != 0);
                    while (m) {
                        const int src = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        double mean, var;
                        walk_exact_window<R, Shape>(g, yo, x_tile + NC * src + o, lane, mean, var);
                        if (lane == src) {
                            const long off = yo * g.ld_out + x_tile + NC * src + o;
                            if (a.out[XRS_STAT_MEAN]) a.out[XRS_STAT_MEAN][off] = (float)mean;
                            if (a.out[XRS_STAT_VAR]) a.out[XRS_STAT_VAR][off] = (float)var;
                            if (a.out[XRS_STAT_STD]) a.out[XRS_STAT_STD][off] = (float)sqrt(var);
                        }
                    }
                }
            }
        }
        t += K;                                            // one turn of the ring: the slot indices repeat
    }

    // true: every result of the tile stands; false (fast body only): a non-finite cell -- the caller redoes the tile
    __device__ __forceinline__ bool run() {
        init();
        while (t < n_in) {
            round(std::make_integer_sequence<int, K>{});
            if (!NANS && badm) return false;
        }
        return true;
    }
};

template <int R, typename Shape, int NC, int NO>
__global__ void __launch_bounds__(256, 2) focal_sw_kernel(const SwArgs a) {
    using C = SwCfg<R, Shape, NC>;
    __shared__ __attribute__((aligned(16))) float lds_rows[4][(C::D + 1) * C::RBF];
    long ty, gx;
    if (!RimFirst(a.groups_x, a.tiles_y, a.rim_first).locate(blockIdx.x, ty, gx)) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long x_tile = (gx * 4 + wv) * C::TW;
    const long y0 = ty * a.tile_rows;
    const WalkGeom &g = a.g;
    if (x_tile >= g.cols) return;
    const long y_end = y0 + a.tile_rows < g.rows ? y0 + a.tile_rows : g.rows;
    // interior: the staged rows (x_tile - HS .. x_tile + TW + HS) and the whole turns of the ring (nin rows from y0 - R) lie
    // inside the raster / the shard's halo rows
    const bool interior = x_tile - C::HS >= 0 && x_tile + C::TW + C::HS <= g.cols && y0 - R >= -(long)g.halo_top &&
                          y0 - R + C::nin((int)(y_end - y0)) <= g.rows + g.halo_bot && y_end - y0 == a.tile_rows;
    if (interior) {
        {
            SwWalk<R, Shape, NC, false, false, NO> w(a, lds_rows[wv], x_tile, y0, y_end, lane);
            if (w.run()) return;
        }
        SwWalk<R, Shape, NC, true, false, NO> w(a, lds_rows[wv], x_tile, y0, y_end, lane);     // a NaN / inf under a window
        w.run();
        return;
    }
    SwWalk<R, Shape, NC, true, true, NO> w(a, lds_rows[wv], x_tile, y0, y_end, lane);
    w.run();
}

template <int R, typename Shape, int NC>
int launch_sw(SwArgs &a, const double *kernel, hipStream_t s) {
    using C = SwCfg<R, Shape, NC>;
    if (!is_shape<R, Shape>(kernel)) return -1;
    WalkGeom &g = a.g;
    const long tiles_x = (g.cols + C::TW - 1) / C::TW;
    a.groups_x = (tiles_x + 3) / 4;
    // tile height: whole turns of the ring; ~128 rows (a tile pays 2R rows of run-in, and the launch wants several rounds of
    // resident workgroups)
    a.tile_rows = C::nin(124) - 2 * R;
    a.tiles_y = (g.rows + a.tile_rows - 1) / a.tile_rows;
    a.rim_first = 1;
    const long grid = RimFirst(a.groups_x, a.tiles_y, a.rim_first).grid();
    if (grid > 0x7fffffffL) return fail("focal statistics: raster too large for one launch");
    int n_out = 0;
    for (int i = 0; i < XRS_NUM_STATS; ++i) n_out += a.out[i] != nullptr;
    if (n_out == XRS_NUM_STATS) hipLaunchKernelGGL((focal_sw_kernel<R, Shape, NC, XRS_NUM_STATS>), dim3((unsigned)grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((focal_sw_kernel<R, Shape, NC, 1>), dim3((unsigned)grid), dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace xrs {

// 0 = launched, -1 = not this shape with a radius of 2 or 3 cells (caller takes another kernel), > 0 = error.
// outs: XRS_STAT_* order, NULL = not wanted.
int XRS_SW_ENTRY(const float *in, float *const *outs, long rows, long cols, long ld_in, long ld_out, const double *kernel,
                 int krows, int kcols, int halo_top, int halo_bot, hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    SwArgs a;
    memset(&a, 0, sizeof(a));
    a.g.in = in; a.g.rows = rows; a.g.cols = cols; a.g.ld_in = ld_in; a.g.ld_out = ld_out;
    a.g.halo_top = halo_top; a.g.halo_bot = halo_bot;
    bool any = false;
    for (int i = 0; i < XRS_NUM_STATS; ++i) { a.out[i] = outs[i]; any |= outs[i] != nullptr; }
    if (!any) return 0;
    switch (krows / 2) {
        case 2: return launch_sw<2, XRS_SW_SHAPE, 4>(a, kernel, s);
#ifndef XRS_SW_NO_R3
        case 3: return launch_sw<3, XRS_SW_SHAPE, 2>(a, kernel, s);
#endif
        default: return -1;
    }
}

}  // namespace xrs
