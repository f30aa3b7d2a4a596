// Focal max / min / range over large circular / box masks -- focal_stats(agg, circle_kernel(...), ['max', 'min', 'range'])
// and apply(func=_calc_max | _calc_min | _calc_range) with 9x9 .. 25x25 windows (xrspatial/focal.py:240-258: numba
// nanmax / nanmin over the cells under `kernel == 1`, window clipped at the raster edge; :782-797 runs one pass per
// statistic).
//
// Third generation of the extrema walk (first: circle_walk.h WalkF32, second: walk2_impl.h's extrema pass).  What the
// round-3 instruction-rate measurements (experiments/valu_rate2.hip, valu_rate3.hip) say about gfx950: v_min / v_max /
// v_min3 / v_max3 -- float or integer -- issue at ~1.9 ns per wave instruction and SIMD whatever the number of resident
// waves, twice the cost of v_add_f32; so the kernel is built to need as FEW of them as possible:
//   * a lane owns ONE column of a 64-column wave tile and walks down; per input row it reads the 2R+1 cells around its
//     column from LDS and forms the running extrema over centred runs from the centre outwards: R v_min3 + R v_max3;
//   * TWO input rows per step: every output row in flight receives the contributions of both rows in ONE v_min3
//     (accumulator, level of row i, level of row i + 1) -- 2R+2 ring operations per two rows instead of 2 (2R+1);
//     together 2R + (R + 1) instead of 2R + (2R + 1) extremum instructions per cell, row and statistic;
//   * rows reach LDS by LDS-DMA, D rows ahead, in a private ring per wave (no barriers, no registers: lds_dma.h; the
//     scheme of wide_impl.h), the ring of output rows lives in registers with compile-time indices (U = 10 rows per
//     unrolled round, rotated once per round);
//   * NaN cells are skipped by the hardware's minNum / maxNum, raster edges stage NaN for the cells outside, and a window
//     without any valid cell comes out NaN by itself (accumulators start from the first pair of levels, not from
//     +-inf): no per-cell bookkeeping, no fall-back path, bit-exact results.
// Included by kxk_ext_circle.hip / kxk_ext_box.hip, which define XRS_EXT_SHAPE / XRS_EXT_ENTRY.
#include "circle_walk.h"
#include "lds_dma.h"

#include <utility>

using namespace xrs;

namespace {

struct ExtArgs {
    WalkGeom g;                   // in, rows, cols, ld_in, ld_out, halo_top, halo_bot; tiles_x / n_tiles: workgroup tiles
    float *out_max, *out_min, *out_range;
    int tile_rows;                // output rows per tile (tile_rows + 2R input rows = a whole number of rounds)
    int rim_first;                // work order (circle_walk.h RimFirst)
};

template <int R, typename Shape>
struct ExtCfg {
    static constexpr int K = 2 * R + 1;
#ifndef XRS_EXT_U
#define XRS_EXT_U 10
#endif
    static constexpr int U = XRS_EXT_U;                    // rows per unrolled round (even: two rows per step)
    static_assert(U % 2 == 0, "two rows per step");
#ifndef XRS_EXT_D
#define XRS_EXT_D 8
#endif
    static constexpr int D = XRS_EXT_D;                    // rows in flight by LDS-DMA (even)
    static_assert(D % 2 == 0, "two rows per step");
    static constexpr int RB = D + 2;                       // row buffers per wave: the two rows being read + D in flight
    static constexpr int CELLS = 64 + 2 * R;               // staged cells per row: raster columns xw - R .. xw + 63 + R
    static constexpr int RBF = 128;                        // floats per row buffer (two dword DMAs of 64 lanes)
    static_assert(2 * R <= 64, "the halo cells are loaded by one lane each");
    // input rows a full tile walks: whole rounds covering `base` output rows + the 2R rows of run-in
    static constexpr int nin(int base) { return ((base + 2 * R + U - 1) / U) * U; }
};

__device__ __forceinline__ float ext_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ext_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ext_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float ext_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// NO: how many outputs the launch writes per row (sets the vmcnt bookkeeping of the DMA ring; fewer than the truth is safe)
template <int R, typename Shape, bool EDGE, int NO>
struct ExtWalk {
    using C = ExtCfg<R, Shape>;
    static constexpr int K = C::K, U = C::U, D = C::D;

    float mn[K], mx[K];
    float pf_own[EDGE ? U : 1], pf_halo[EDGE ? U : 1];     // EDGE: the rows of the current round, loaded up front
    int slot_in, slot_out;         // interior: ring slots of the next DMA / of the first row of the step
    unsigned ring_addr;
    int t;                         // input row counter: row y_first + t

    const ExtArgs &a;
    const WalkGeom &g;
    float *lds;                    // this wave's RB row buffers
    long xw, x, y0, y_end, y_first;
    int n_in, lane;
    const float *dma_src;          // interior: (wave-uniform) first staged cell of the next row to DMA
    int dma_adv;
    long out_off;                  // offset of the wave tile's next output row in every plane

    __device__ __forceinline__ ExtWalk(const ExtArgs &a_, float *lds_, long xw_, long y0_, long ye, int lane_)
        : a(a_), g(a_.g), lds(lds_), xw(xw_), x(xw_ + lane_), y0(y0_), y_end(ye), lane(lane_) {}

    // EDGE: predicated loads, NaN for every cell outside the raster (skipped by v_min3 / v_max3)
    __device__ __forceinline__ void load_row(int il, float &own, float &halo) const {
        const long yy = y_first + il;
        own = halo = nan_f32();
        const bool row_ok = il < n_in && yy >= -(long)g.halo_top && yy < g.rows + g.halo_bot;     // wave-uniform
        if (!row_ok) return;
        const float *p = g.in + yy * g.ld_in;
        const long xa = xw - R + lane, xb = xa + 64;
        if (xa >= 0 && xa < g.cols) own = p[xa];
        if (lane < 2 * R && xb >= 0 && xb < g.cols) halo = p[xb];
    }

    // interior: input row `il` (clamped past the tile) -> ring slot `slot`; staged cell s <-> raster column xw - R + s
    // (the rows are taken in order: the source pointer advances by a row per call -- dma_adv more times, rows past the tile
    //  repeat the last -- instead of being re-derived from the row index with a 64-bit scalar multiply)
    __device__ __forceinline__ void dma_row(int slot) {
        const float *p = uniform_ptr(dma_src);
        dma_src += dma_adv > 0 ? g.ld_in : 0;
        --dma_adv;
        const unsigned dst = ring_addr + (unsigned)slot * (C::RBF * 4);
        glds4_s(p, 4u * (unsigned)lane, dst);
        glds4_s(p, 4u * (unsigned)(64 + (lane < 2 * R ? lane : 2 * R - 1)), dst + 256);        // (lanes >= 2R: a slot nobody reads)
    }

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j) { mn[j] = 0.0f; mx[j] = 0.0f; }        // (every slot is assigned before its first use)
        t = 0;
        y_first = y0 - R;
        n_in = (int)(y_end - y0) + 2 * R;                // (interior tiles: a whole number of rounds)
        ring_addr = lds_addr(lds);
        out_off = y0 * g.ld_out + xw;
        if (!EDGE) {
            dma_src = uniform_ptr(g.in + y_first * g.ld_in + (xw - R));
            dma_adv = n_in - 1;
            for (int r = 0; r < D; ++r) dma_row(r);
            slot_in = D;
            slot_out = 0;
        }
    }

    // What an output row needs from one staged row at offset dy: index IDX(|dy|) of the arrays row_levels() fills.
    // Shapes without a hole: the running extrema over the centred runs, lo[h] / hi[h] = min / max of cells R - h .. R + h,
    // indexed by the half-width h = hw(dy): R v_min3 + R v_max3.  Shapes with a hole (annuli): indexed by |dy| itself -- rows
    // outside the hole take the centred run's value, rows across it the extrema over the SHELL of cell pairs
    // hwi < |dx| <= hw (a chain of hw - hwi v_min3 per distinct row pattern; minNum / maxNum skip NaN cells as before).
    static constexpr bool HOLE = shape_has_hole<Shape>(R);
    static constexpr int IDX(int dy) { return HOLE ? dy : Shape::hw(R, dy); }
    __device__ __forceinline__ void row_levels(unsigned row_addr, float (&lo)[R + 1], float (&hi)[R + 1]) const {
        float v[K];
        lds_cfloat *row = lds_row_ptr(row_addr + 4u * (unsigned)lane);
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = row[k];
        if (!HOLE) {
            lo[0] = hi[0] = v[R];
#pragma unroll
            for (int h = 1; h <= R; ++h) {
                lo[h] = ext_min3(lo[h - 1], v[R - h], v[R + h]);
                hi[h] = ext_max3(hi[h - 1], v[R - h], v[R + h]);
            }
            return;
        }
        // centred runs for the rows outside the hole (as far out as the widest of them)
        constexpr ShapeRows<R, Shape> T{};
        constexpr int HMAX = [] { int m = 0; constexpr ShapeRows<R, Shape> t{}; for (int d = 0; d <= R; ++d) if (t.hwi[d] < 0 && t.hw[d] > m) m = t.hw[d]; return m; }();
        float clo[R + 1], chi[R + 1];
        clo[0] = chi[0] = v[R];
#pragma unroll
        for (int h = 1; h <= HMAX; ++h) {
            clo[h] = ext_min3(clo[h - 1], v[R - h], v[R + h]);
            chi[h] = ext_max3(chi[h - 1], v[R - h], v[R + h]);
        }
#pragma unroll
        for (int d = 0; d <= R; ++d) {
            const int h1 = T.hw[d], h0 = T.hwi[d];
            if (h0 < 0) { lo[d] = clo[h1 <= HMAX ? h1 : 0]; hi[d] = chi[h1 <= HMAX ? h1 : 0]; continue; }
            if (T.pat[d] != d) { lo[d] = lo[T.pat[d]]; hi[d] = hi[T.pat[d]]; continue; }
            float a = ext_min(v[R - (h0 + 1)], v[R + (h0 + 1)]), b = ext_max(v[R - (h0 + 1)], v[R + (h0 + 1)]);
#pragma unroll
            for (int h = h0 + 2; h <= h1; ++h) {
                a = ext_min3(a, v[R - h], v[R + h]);
                b = ext_max3(b, v[R - h], v[R + h]);
            }
            lo[d] = a; hi[d] = b;
        }
    }

    // (output rows are emitted in order from y0: the offset of the wave tile's row advances by a row per call)
    __device__ __forceinline__ void emit(long yo, float lo, float hi) {
        const long off = out_off;                             // (wave-uniform row address + 4 * lane)
        out_off += g.ld_out;
        if (EDGE && (x >= g.cols || yo >= y_end)) return;
#ifdef XRS_FLOOR_NO_STORES                                     // (tools/floor_probe.sh: the walk without its output streams)
        if (g.rows >= 0) return;
#endif
        if (a.out_max) st_row_nt(uniform_ptr(a.out_max + off), 4u * (unsigned)lane, hi);
        if (a.out_min) st_row_nt(uniform_ptr(a.out_min + off), 4u * (unsigned)lane, lo);
        if (a.out_range) st_row_nt(uniform_ptr(a.out_range + off), 4u * (unsigned)lane, hi - lo);
    }

    // rows t + PH and t + PH + 1 (PH even)
    template <int PH>
    __device__ __forceinline__ void step2() {
        const int i = t + PH;
        unsigned row1, row2;           // LDS byte addresses of the two staged rows
        if (EDGE) {
            if (i >= n_in) return;
            // ---- both rows -> LDS (two buffers; LDS serves a wave's instructions in order, so the next step's writes
            // cannot overtake this step's reads)
            float *b1 = lds, *b2 = lds + C::RBF;
            b1[lane] = pf_own[PH];
            b2[lane] = pf_own[PH + 1];
            if (lane < 2 * R) { b1[64 + lane] = pf_halo[PH]; b2[64 + lane] = pf_halo[PH + 1]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            row1 = ring_addr; row2 = ring_addr + C::RBF * 4;
        } else {
            dma_row(slot_in);
            slot_in = slot_in + 1 == C::RB ? 0 : slot_in + 1;
            dma_row(slot_in);
            slot_in = slot_in + 1 == C::RB ? 0 : slot_in + 1;
            // Rows i, i + 1 were issued D / 2 steps ago.  Younger vector-memory operations: the DMAs of rows i + 2 ..
            // i + D + 1 (2 each) and -- once the walk emits, from row 2R on -- 2 NO stores per step in between.
            if (i >= 2 * R + D) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * D + (D / 2) * 2 * NO) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * D) : "memory");
            row1 = ring_addr + (unsigned)slot_out * (C::RBF * 4);
            slot_out = slot_out + 1 == C::RB ? 0 : slot_out + 1;
            row2 = ring_addr + (unsigned)slot_out * (C::RBF * 4);
            slot_out = slot_out + 1 == C::RB ? 0 : slot_out + 1;
        }
        float lo1[R + 1], hi1[R + 1], lo2[R + 1], hi2[R + 1];
#ifdef XRS_FLOOR_NO_ARITH                                      // (tools/floor_probe.sh: the DMA ring and the stores, no reads / extrema)
#pragma unroll
        for (int h = 0; h <= R; ++h) { lo1[h] = hi1[h] = lo2[h] = hi2[h] = 0.0f; }
        if (g.rows < 0)
#endif
        {
        row_levels(row1, lo1, hi1);
        row_levels(row2, lo2, hi2);
        }
        if (EDGE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- ring: the output row that sees row i at offset dy1 sees row i + 1 at dy1 + 1 (slot (PH - dy1) mod K)
#pragma unroll
        for (int dy1 = -R; dy1 < R; ++dy1) {
            constexpr int dummy = 0; (void)dummy;
            const int idx = ((PH - dy1) % K + K) % K;
            const int h1 = IDX(dy1 < 0 ? -dy1 : dy1), h2 = IDX(dy1 + 1 < 0 ? -(dy1 + 1) : dy1 + 1);
            if (dy1 == -R) {           // a new output row: its first two contributions
                mn[idx] = ext_min(lo1[h1], lo2[h2]);
                mx[idx] = ext_max(hi1[h1], hi2[h2]);
            } else {
                mn[idx] = ext_min3(mn[idx], lo1[h1], lo2[h2]);
                mx[idx] = ext_max3(mx[idx], hi1[h1], hi2[h2]);
            }
        }
        // ---- row i completes the output row R rows up (its slot restarts with row i + 1 at dy = -R); row i + 1
        // completes the next one (its slot restarts in the next step)
        constexpr int IA = ((PH - R) % K + K) % K, IB = ((PH - R + 1) % K + K) % K;
        constexpr int HR = IDX(R);
        const float loA = ext_min(mn[IA], lo1[HR]), hiA = ext_max(mx[IA], hi1[HR]);
        if (i >= 2 * R) {
            emit(y0 + (i - 2 * R), loA, hiA);
            emit(y0 + (i - 2 * R) + 1, mn[IB], mx[IB]);
        }
        mn[IA] = lo2[HR];
        mx[IA] = hi2[HR];
    }

    template <int... P>
    __device__ __forceinline__ void round(std::integer_sequence<int, P...>) {
        if (EDGE) {
#pragma unroll
            for (int r = 0; r < U; ++r) load_row(t + r, pf_own[r], pf_halo[r]);      // all loads of the round first
        }
        (step2<2 * P>(), ...);
        // the round started at row t with ring slot (j - t) mod K for output row j; the next one starts at t + U
        ring_rotate<K, U>(mn);
        ring_rotate<K, U>(mx);
        t += U;
    }

    __device__ __forceinline__ void run() {
        init();
        while (t < n_in) round(std::make_integer_sequence<int, U / 2>{});
    }
};

#ifndef XRS_EXT_WAVES
#define XRS_EXT_WAVES 3           // workgroups per CU = waves per SIMD
#endif
template <int R, typename Shape, int NO>
__global__ void __launch_bounds__(256, XRS_EXT_WAVES) focal_ext_kernel(const ExtArgs a) {
    using C = ExtCfg<R, Shape>;
    __shared__ __attribute__((aligned(16))) float lds_rows[4][C::RB * C::RBF];
    const WalkGeom &g = a.g;
    long ty, tx;
    if (!RimFirst(g.tiles_x, g.n_tiles / g.tiles_x, a.rim_first).locate(blockIdx.x, ty, tx)) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long xw = tx * 256 + wv * 64;
    const long y0 = ty * a.tile_rows;
    if (xw >= g.cols) return;
    const long y_end = y0 + a.tile_rows < g.rows ? y0 + a.tile_rows : g.rows;
    const bool interior = xw - R >= 0 && xw + 64 + R <= g.cols && y0 - R >= -(long)g.halo_top &&
                          y_end + R <= g.rows + g.halo_bot && y_end - y0 == a.tile_rows;
    if (interior) {
        ExtWalk<R, Shape, false, NO> w(a, lds_rows[wv], xw, y0, y_end, lane);
        w.run();
    } else {
        ExtWalk<R, Shape, true, NO> w(a, lds_rows[wv], xw, y0, y_end, lane);
        w.run();
    }
}

template <int R, typename Shape>
int launch_ext(ExtArgs &a, const double *kernel, hipStream_t s) {
    using C = ExtCfg<R, Shape>;
    if (!is_shape<R, Shape>(kernel)) return -1;
    WalkGeom &g = a.g;
    g.tiles_x = (g.cols + 255) / 256;
    static thread_local int wg_per_cu = 0;                     // (per instantiation: registers depend on the radius)
    if (!wg_per_cu) wg_per_cu = walk3_wg_per_cu(focal_ext_kernel<R, Shape, 3>, XRS_EXT_WAVES);
    a.tile_rows = C::nin(walk3_tile_base(g.rows, g.tiles_x, R, C::U, wg_per_cu)) - 2 * R;
    g.n_tiles = g.tiles_x * ((g.rows + a.tile_rows - 1) / a.tile_rows);
    a.rim_first = RimFirst::mode_from_env();
    const long grid = RimFirst(g.tiles_x, g.n_tiles / g.tiles_x, a.rim_first).grid();
    if (grid > 0x7fffffffL) return fail("focal max / min: raster too large for one launch");
    const int n_out = (a.out_max != nullptr) + (a.out_min != nullptr) + (a.out_range != nullptr);
    if (n_out == 3) hipLaunchKernelGGL((focal_ext_kernel<R, Shape, 3>), dim3((unsigned)grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((focal_ext_kernel<R, Shape, 1>), dim3((unsigned)grid), dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace xrs {

#ifndef XRS_EXT_ANNULUS_RMIN
// 0 = launched, -1 = not this shape with a radius of 4..12 cells (caller takes another kernel), > 0 = error.
int XRS_EXT_ENTRY(const float *in, float *out_max, float *out_min, float *out_range, long rows, long cols, long ld_in,
                  long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    if (!out_max && !out_min && !out_range) return 0;
    ExtArgs a;
    memset(&a, 0, sizeof(a));
    a.g.in = in; a.g.rows = rows; a.g.cols = cols; a.g.ld_in = ld_in; a.g.ld_out = ld_out;
    a.g.halo_top = halo_top; a.g.halo_bot = halo_bot;
    a.out_max = out_max; a.out_min = out_min; a.out_range = out_range;
    switch (krows / 2) {
#define XRS_EXT_CASE(RR) case RR: return launch_ext<RR, XRS_EXT_SHAPE>(a, kernel, s);
#ifndef XRS_EXT_PROBE
        XRS_EXT_CASE(4) XRS_EXT_CASE(5) XRS_EXT_CASE(6) XRS_EXT_CASE(7) XRS_EXT_CASE(8) XRS_EXT_CASE(9) XRS_EXT_CASE(10) XRS_EXT_CASE(11)
#endif
        XRS_EXT_CASE(12)
#undef XRS_EXT_CASE
        default: return -1;
    }
}
#else
// annulus_kernel(1, 1, R, RI) for XRS_EXT_ANNULUS_RMIN <= R <= XRS_EXT_ANNULUS_RMAX, 1 <= RI < R: one instantiation per pair.
// 0 = launched, -1 = not such an annulus, > 0 = error.
template <int RR, int RI>
int ext_annulus_pair(ExtArgs &a, const double *kernel, int ri, hipStream_t s) {
    if constexpr (RI >= RR) return -1;
    else {
        if (ri == RI) return launch_ext<RR, AnnulusShape<RI>>(a, kernel, s);
        return ext_annulus_pair<RR, RI + 1>(a, kernel, ri, s);
    }
}
template <int RR>
int ext_annulus_radius(ExtArgs &a, const double *kernel, int r, int ri, hipStream_t s) {
    if constexpr (RR > XRS_EXT_ANNULUS_RMAX) return -1;
    else {
        if (r == RR) return ext_annulus_pair<RR, 1>(a, kernel, ri, s);
        return ext_annulus_radius<RR + 1>(a, kernel, r, ri, s);
    }
}
int XRS_EXT_ENTRY(const float *in, float *out_max, float *out_min, float *out_range, long rows, long cols, long ld_in,
                  long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    const int r = krows / 2;
    if (r < XRS_EXT_ANNULUS_RMIN || r > XRS_EXT_ANNULUS_RMAX) return -1;
    const int ri = annulus_inner_radius(kernel, krows);
    if (ri < 1) return -1;
    if (!out_max && !out_min && !out_range) return 0;
    ExtArgs a;
    memset(&a, 0, sizeof(a));
    a.g.in = in; a.g.rows = rows; a.g.cols = cols; a.g.ld_in = ld_in; a.g.ld_out = ld_out;
    a.g.halo_top = halo_top; a.g.halo_bot = halo_bot;
    a.out_max = out_max; a.out_min = out_min; a.out_range = out_range;
    return ext_annulus_radius<XRS_EXT_ANNULUS_RMIN>(a, kernel, r, ri, s);
}
#endif

}  // namespace xrs
