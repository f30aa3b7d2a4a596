// Focal mean (and the window sum) over large circular / box masks -- focal.apply(raster, circle_kernel(...)) and
// focal_stats(..., ['mean']) with 7x7 .. 25x25 windows (xrspatial/focal.py:305-326 with _calc_mean :226-228; the
// reference gathers the window per cell and calls numba's nanmean: float64 sum / count, float32 store).
//
// The "wide" row walker: ONE 16-byte load per lane per input row, neighbours through LDS, float32 arithmetic on
// shifted values, a register ring with static indices.
//   * a wave owns a tile of 64 NC columns x ~130 output rows and walks DOWN its input rows; a lane owns NC adjacent
//     columns (NC = 2 since the end of round 2: half the registers of NC = 4 and a third wave per SIMD are worth more than
//     the shared halo cells).  Every input row reaches LDS once (64 NC + 2*HL cells) and every lane reads back the
//     NC + 2*HL consecutive cells its windows cover as aligned vector reads (conflict free), minus a wave-uniform shift c
//     (the cell at the tile centre).
//   * a lane-local prefix sum over those cells turns every centred run of the mask into ONE subtraction,
//     S_h(x) = P[x + h] - P[x - h - 1]; a circle of radius 12 has only 9 distinct half-widths.
//   * the 2R+1 output rows in flight live in a register ring, acc[(row - dy) mod (2R+1)]; the row loop is unrolled U = 5
//     times so that every ring index is a compile-time constant, and the ring is rotated by U slots once per U rows
//     (4 * (2R+1) / U register moves per row).  Unrolling all 2R+1 phases needs no moves at all but makes a 50 KB loop
//     body: 16 waves at different phases then stream it through the 64 KB instruction cache two CUs share, and the
//     kernel runs at instruction-fetch speed (measured: 1.20 ms against 0.6 ms; profiles/r02).
//   * mean = c + S / n.  Everything is float32: the error of S is bounded by u * A * (a few 10^4) with A the largest
//     |v - c| of the tile, i.e. <= 7e-6 * A on the mean; the wave checks A <= 1.4 * min |mean| at the end of its tile
//     (which guarantees 1e-5 relative) and that every result is finite, and otherwise hands the whole tile on: filled
//     without a walk if every cell it sees is NaN, else to the NaN-aware float32 walker of mom_nan_walk.h in its mean / sum
//     mode (nodata regions, scattered NaN cells), and from there -- +-inf, rasters whose values straddle zero -- to the
//     float64 column walker of circle_walk.h (NaN-skipping, counting, exact).  Typical errors are ~1e-8 relative
//     (tests: 1e-6 on both DEMs).
//   * nodata (round 5).  A NaN under some window used to end the tile's walk (one non-finite sum) and send the whole tile to the
//     NaN-aware walker below -- with 0.1 % of the cells NaN that is every tile, at ~0.34 ms of a wave's time each, and the 25x25
//     mean took 1.72 ms instead of 0.55.  Interior tiles of the mean / sum (circles and boxes) now carry NaN cells themselves.
//     At the head of a step every lane looks at the cells of the row it stages for the wave (its NC, the first lanes also the
//     halo cells: two LDS reads) and the wave votes; a row that holds NaN (rare) has those cells OVERWRITTEN in the ring with
//     the shift -- a cell equal to the shift adds nothing to any sum, so the reads, prefix sums, rings and the sliding box sum
//     behind it never know -- and their positions noted in a 152-bit bitmap.  Every lane then takes the bits of its own NV
//     cells from the bitmap and, for each of the 2R+1 output rows this input row lies under, adds the number of them inside
//     that row's run (one popcount per distinct half-width and column) to a LOST RING in LDS: one byte per owned column, slot
//     = the step that completes that output row, mod 2R+1 (3.2 KiB per wave at 25x25; atomic adds on the word two lanes
//     share).  An output row completed while a NaN row is among its 2R+1 input rows (a scalar shift register remembers)
//     reads its lane's entry, clears it and divides by n - lost.  No count ring in registers (they are not there: 160 of 168),
//     nothing on clean rows but the vote and one scalar test.  +-inf, windows that lost more than half their cells and tiles
//     whose rows have shown one lane more than 255 NaN cells in all still hand the tile on; so do edge tiles and the annuli.
//     (Forms that did not survive: the vote on the prefix total behind the read -- free, but the row then has to be read
//     again, and a second copy of the read spilled 37 registers into the round loop, whose scratch traffic sat in the vmcnt
//     bookkeeping of the DMA ring: 0.59 -> 2.2 ms on a CLEAN raster; a two-trip loop around one copy; counting the lost cells at
//     every output row from a ring of bitmap rows: a scalar loop of dependent LDS reads, 1.6 ms at 0.1 % NaN.)
//   * raster edges (clipped windows): out-of-raster cells enter as d = 0 and the divisor is the geometric count of
//     in-raster cells, so edge tiles stay on the fast path (a second, predicated instantiation of the same walk).
//   * memory latency: the interior walk prefetches its rows D = 8 ahead by LDS-DMA (global_load_lds_dwordx4: global ->
//     LDS without passing through registers) into a ring of D + 1 row buffers per wave, and waits with an explicit
//     `s_waitcnt vmcnt(2 D)` for the row it is about to read (2 DMAs per row; stores issued in between only make the
//     wait conservative).  Register prefetching does not work with this compiler: load results consumed by the NEXT
//     loop iteration are copied into their phi registers at the loop latch, every copy needs its load, and hipcc emits
//     `s_waitcnt vmcnt(0)` once per round -- the very latency the prefetch was meant to hide (0.91 ms for the 25x25
//     mean); loading a whole round up front costs 50 registers and exposes one HBM round trip per round (0.72 ms);
//     the DMA ring needs no registers at all (experiments/glds_walk.hip: the bare data movement runs at 0.42-0.47 ms).
//     The DMA delivers raw cells, so the shift is subtracted after the read-back (NV instead of NC subtractions).
//     Edge tiles keep plain predicated loads (one round = U rows loaded up front).
// Included by kxk_wide_circle.hip and kxk_wide_box.hip, which define XRS_WIDE_SHAPE / XRS_WIDE_ENTRY.
// vs the one-column walker this replaces for `mean`: 25 dword loads + ~270 VALU instructions per cell and row ->
// 0.3 loads + ~50.  HBM-bound by construction (8 B per cell); measured numbers in DESIGN.md.
#include "circle_walk.h"
#include "lds_dma.h"
#include "mom_nan_walk.h"
#include "wave_reduce.h"

#include <type_traits>
#include <utility>

using namespace xrs;

namespace {

#ifndef XRS_WIDE_NO_FALLBACK
#define XRS_WIDE_NO_FALLBACK 0
#endif
#ifndef XRS_WIDE_CARRY
#define XRS_WIDE_CARRY 1          // NaN tiles: the carrying walk (below) before the NaN-aware walker of mom_nan_walk.h
#endif
struct WideArgs {
    WalkGeom g;                   // in, rows, cols, ld_in, ld_out, halo_top, halo_bot (tiles_x / n_tiles: wave tiles)
    float *out;                   // the mean, the window sum, or the convolution (template parameter of the kernel)
    double wgt;                   // convolution: the one weight value of the kernel ...
    const double *weights;        // ... and the whole (2R+1)^2 kernel in device memory (exact path)
    int tile_rows;                // output rows per tile (tile_rows + 2R input rows = a whole number of rounds)
    int rim_first;                // work order (circle_walk.h RimFirst)
    long n_groups;                // workgroups = groups of 4 horizontally adjacent wave tiles
    long groups_x;
    unsigned *rescue;             // work-list of the wave tiles handed on to focal_wide_rescue_kernel: [0] count, [2..] tiles; or NULL
    unsigned rescue_cap;
};

template <int R, typename Shape>
struct WideCfg {
    static constexpr int K = 2 * R + 1;
#ifndef XRS_WIDE_NC
#define XRS_WIDE_NC 2
#endif
    static constexpr int NC = XRS_WIDE_NC;                 // columns per lane.  4 (256-column wave tiles): 250 VGPRs, 2 waves per SIMD, 25x25
                                                           // mean 0.65 ms; 2 (128-column tiles): 161 VGPRs, 3 waves, 0.57 ms although a lane
                                                           // then reads 26 cells for 2 outputs instead of 28 for 4 (profiles/r02/r02o_ab_wide_nc.log)
    static constexpr int TW = 64 * NC;                     // columns per wave tile
    static constexpr int HL = NC * ((R + NC - 1) / NC);    // halo columns each side, rounded up to whole lane groups
    static constexpr int NV = NC + 2 * HL;                 // cells a lane reads back per row
    static constexpr int NQ = NV / NC;
    static constexpr int STG = TW + 64;                    // staged cells per row (TW + 2*HL used; every lane writes one halo slot)
    static_assert(2 * HL <= 64, "the halo cells are loaded by one lane each");
    static constexpr int NTAPS = shape_taps<Shape>(R);
#ifndef XRS_WIDE_SLIDE
#define XRS_WIDE_SLIDE 1
#endif
    // np.ones boxes: every row of the window has the same half-width, so the window sum SLIDES down the raster -- V -= H(row
    // that left), V += H(entering row) -- with the ring holding the last row sums H instead of 2R+1 partial window sums: 3
    // ring operations per row and column instead of 2R+1.  That ring is not rotated (its slots change once per row, a rotation
    // would be 2R+1 moves per round, and so would the register copies at the end of a switch over the round's position in
    // it -- both measured): it has KR = a whole number of rounds >= 2R+1 slots, the loop body is all KR / U rounds in a straight
    // line, and a full tile walks a multiple of KR input rows.  V is re-summed from the ring once per KR rows, so the
    // recurrence rounds at most 2 KR times between two exact states.  Same box, same run (tools/ab_wide.sh): 25x25 mean
    // 0.55 -> 0.465 ms, uniform-weight 25x25 convolution 0.51 -> 0.465; 15x15 the same either way, 11x11 5 % slower (their
    // ring was 15 / 11 additions to begin with): radius >= 10 only.
    static constexpr bool SLIDE = XRS_WIDE_SLIDE && R >= 10 && std::is_same<Shape, BoxShape>::value;
#ifndef XRS_WALK_U
#define XRS_WALK_U 5
#endif
    static constexpr int U = XRS_WALK_U;                   // rows per unrolled round (the accumulator ring is rotated by U after each)
#ifndef XRS_WIDE_D
#define XRS_WIDE_D 8
#endif
    static constexpr int KR = SLIDE ? U * ((K + U - 1) / U) : K;   // ring slots
    static constexpr int D = XRS_WIDE_D;                   // interior tiles: rows in flight by LDS-DMA; D + 1 row buffers per wave
    static constexpr int RBF = (TW + 2 * HL > 256) ? (STG > 320 ? STG : 320) : 256;   // floats per ring row (the 16-byte DMA writes a whole KiB, the dword one 256 B more)
    static constexpr int LDS_WAVE = (D + 1) * RBF;    // floats of LDS per wave (a DMA writes whole KiB)
    // input rows a full tile walks: whole rounds covering `base` output rows + the 2R rows of run-in (round 3: the
    // tile height is chosen at launch, walk3_tile_base)
    static constexpr int UU = SLIDE ? KR : U;              // rows per trip of the walk loop = granularity of a tile's input rows
    static constexpr int nin(int base) { return ((base + 2 * R + UU - 1) / UU) * UU; }
    static constexpr int NE = 2 * R;                       // the first input row whose completion emits an output row
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
        const int h = Shape::hw(R, dy < 0 ? -dy : dy), h0 = Shape::hwi(R, dy < 0 ? -dy : dy);
        const long a = x - h < 0 ? 0 : x - h, b = x + h > cols - 1 ? cols - 1 : x + h;
        n += (int)(b - a + 1);
        if (h0 >= 0) {                                       // annuli: the hole's cells inside the raster
            const long a0 = x - h0 < 0 ? 0 : x - h0, b0 = x + h0 > cols - 1 ? cols - 1 : x + h0;
            n -= b0 >= a0 ? (int)(b0 - a0 + 1) : 0;
        }
    }
    return n;
}

// EDGE = false: a full tile whose whole input window lies inside the raster (no predicates, see the header);
// EDGE = true: everything else (predicated loads / stores, clipped counts, partial tiles).
// MODE: what is emitted -- the mean, the window sum, or convolve_2d with ONE weight value on the mask (convolution.py:285-313:
// w * sum over the full window, NaN within R cells of the raster edge, and NaN whenever the SQUARE window holds a
// non-finite cell, zero weights included: any non-finite cell among those the tile reads sends the tile to the exact walker).
enum : int { WIDE_MEAN = 0, WIDE_SUM = 1, WIDE_CONV = 2 };
template <int R, typename Shape, bool EDGE, int MODE, bool CARRY = false>
struct WideWalk {
    static constexpr bool SUM = MODE == WIDE_SUM, CONV = MODE == WIDE_CONV;
    using C = WideCfg<R, Shape>;
    static constexpr int K = C::K, HL = C::HL, NV = C::NV, NQ = C::NQ, U = C::U, NC = C::NC, TW = C::TW;
    // CARRY: nodata carried by the walk itself (header) -- the second walk of a tile whose first, plain walk met a NaN
    // (not the annuli: four of the radius-10 rings spill 1-4 registers into the round loop with it; their NaN tiles go the old way)
    static constexpr bool NANOK = CARRY && !CONV && !shape_has_hole<Shape>(R) && NC <= 2 && NV <= 32 && TW + 2 * HL <= 160 &&
                                  32 % NC == 0;
    static constexpr int NMW = 6;  // words of the bitmap (TW + 2 HL <= 160 bits, + one the last lane's read may touch)

    // ---- state
    float acc[C::KR][NC];          // ring: partial window sums of the 2R+1 output rows in flight (SLIDE: the last KR row sums H)
    float vsum[C::SLIDE ? NC : 1]; // SLIDE: the sliding window sum V
    float pf_own[EDGE ? U : 1][NC];   // EDGE: the rows of the current round, loaded up front
    float pf_halo[EDGE ? U : 1];
    int slot_in, slot_out;         // interior: ring slots of the next DMA / of the row being processed
    unsigned ring_addr;            // LDS byte address of this wave's ring
    const float *dma_src;          // interior: (wave-uniform) first staged cell of the next row to DMA ...
    int dma_adv;                   // ... and how many more times it advances (rows past the tile repeat the last one)
    float *out_row;                // interior: (wave-uniform) first cell of the next output row of this wave tile
    float amax, mmin;
    bool bad;
    unsigned inflight;             // (wave-uniform) bit b: input row t - b holds NaN cells (they were zeroed; positions in nanmap)
    bool saw_nan;                  // (wave-uniform) some row of the tile did
    unsigned *nanmap;              // LDS: NMW words, the NaN bitmap of the row being marked: bit s = staged cell s is NaN
    unsigned short *lostring;      // LDS: [K][64] -- slot (step mod K), lane: NaN cells under the windows of the output row that step
                                   // completes, one byte per owned column
    int lost_slot;                 // (wave-uniform) t mod K of the current step
    int span_total;                // (wave-uniform) NaN cells the rows of this tile have shown their worst lane, summed
    int lost_o[NC];                // the lost-ring entry of the output row this step completes (read early: its latency hides behind the row's sums)
    int t;                         // input row counter: row y_first + t
    // ---- constants of the tile
    const WalkGeom &g;
    float *out;
    float *lds;                    // this wave's (D + 1) row buffers of STG floats (EDGE uses the first)
    long x_tile, y0, y_end, y_first;
    int n_in, lane;
    float c;                       // the shift
    float wgt;                     // CONV: the weight
    float n_full[NC];              // EDGE: cell count of a window whose rows are all inside, per owned column

    __device__ __forceinline__ WideWalk(const WalkGeom &g_, float *out_, float *lds_, long xt, long y0_, long ye, int lane_)
        : g(g_), out(out_), lds(lds_), x_tile(xt), y0(y0_), y_end(ye), lane(lane_) {}

    typedef float vecNC __attribute__((ext_vector_type(NC), aligned(4)));

    __device__ __forceinline__ void load_row(int il, float (&own)[NC], float &halo) const {
        // staged cell s <-> raster column x_tile - HL + s; lane owns s = NC*lane .. NC*lane+NC-1; the 2*HL halo cells
        // s = TW + lane come from the first 2*HL lanes (the other lanes repeat the last one: no branch, no use)
        const long yy = y_first + il;
        const long xs = x_tile - HL + NC * lane;
#pragma unroll
        for (int e = 0; e < NC; ++e) own[e] = c;             // out-of-raster cells: d = 0 after the shift
        halo = c;
        const bool row_ok = il < n_in && yy >= -(long)g.halo_top && yy < g.rows + g.halo_bot;     // wave-uniform
        if (!row_ok) return;
        const float *p = g.in + yy * g.ld_in;
#pragma unroll
        for (int e = 0; e < NC; ++e)
            if (xs + e >= 0 && xs + e < g.cols) own[e] = p[xs + e];
        const long xh = x_tile - HL + TW + lane;
        if (lane < 2 * HL && xh >= 0 && xh < g.cols) halo = p[xh];
    }

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < C::KR; ++j)
#pragma unroll
            for (int o = 0; o < NC; ++o) acc[j][o] = 0.0f;
#pragma unroll
        for (int o = 0; o < (C::SLIDE ? NC : 1); ++o) vsum[o] = 0.0f;
        amax = 0.0f;
        mmin = INFINITY;
        bad = false;
        inflight = 0u;
        saw_nan = false;
        lost_slot = K - 1;
        span_total = 0;
        t = 0;
        y_first = y0 - R;
        n_in = (int)(y_end - y0) + 2 * R;                // (interior tiles: a whole number of rounds)
        // shift: the cell at the tile centre (any finite value works; a nearby one keeps |v - c| small)
        const long yc = y0 + (y_end - y0) / 2, xc = (x_tile + TW / 2 < g.cols ? x_tile + TW / 2 : g.cols - 1);
        float c0 = g.in[yc * g.ld_in + xc];
        if (!isfinite(c0)) {
            // the centre cell is nodata: the first finite cell of the 64 to its left on that row instead (shift 0 on a raster of
            // values around 1000 fails the rounding bound below: at 0.1 % nodata that was 8 tiles of 16 128 handed to the NaN-aware
            // walker -- and whichever of them ran last, ~0.3 ms of a lone wave at the end of the launch)
            const long xl = xc - lane >= 0 ? xc - lane : 0;
            const float cand = g.in[yc * g.ld_in + xl];
            const unsigned long long fin = __ballot(isfinite(cand));
            c0 = fin ? __shfl(cand, __builtin_ctzll(fin)) : 0.0f;
        }
        c = c0;
        if constexpr (NANOK) {
#pragma unroll
            for (int j = 0; j < K; ++j) lostring[j * 64 + lane] = 0;
        }
        if (EDGE) {
#pragma unroll
            for (int o = 0; o < NC; ++o)
                n_full[o] = (float)clipped_count<R, Shape>(0, x_tile + NC * lane + o, -(long)R, (long)R + 1, g.cols);
        } else {
            ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
            ring_addr = __builtin_amdgcn_readfirstlane(ring_addr);
            dma_src = uniform_ptr(g.in + y_first * g.ld_in + (x_tile - HL));
            dma_adv = n_in - 1;
            out_row = out + y0 * g.ld_out + x_tile;
            for (int r = 0; r < C::D; ++r) dma_row(r);
            slot_in = C::D;
            slot_out = 0;
        }
    }

    // interior: input row `il` (clamped past the tile) -> ring slot `slot`, as a linear image of the TW + 2*HL staged cells
    // (the rows are taken in order: the source pointer advances by one row per call instead of being re-derived from the
    // row index -- a 64-bit scalar multiply per row)
    __device__ __forceinline__ void dma_row(int slot) {
        constexpr int CELLS = TW + 2 * HL;
        const float *p = uniform_ptr(dma_src);                                              // (scalar base + lane offset)
        dma_src += dma_adv > 0 ? g.ld_in : 0;
        --dma_adv;
        const unsigned dst = ring_addr + (unsigned)slot * (C::RBF * 4);
        static_assert(CELLS % 4 == 0, "whole 16-byte pieces");
        constexpr int QMAX = (CELLS < 256 ? CELLS : 256) / 4 - 1;
        glds16_s(p, 16u * (unsigned)(lane < QMAX ? lane : QMAX), dst);
        if (CELLS > 256) glds4_s(p, 4u * (unsigned)(256 + (lane < CELLS - 257 ? lane : CELLS - 257)), dst + 1024);
    }
    static constexpr int NDMA = (TW + 2 * HL > 256) ? 2 : 1;       // DMA instructions per row

    // One row of the round.  There are no exits inside a round of U steps: with early returns the compiler sinks the
    // ring updates of all phases into the loop latch and spills their operands.
    template <int PHASE, int BASE>
    __device__ __forceinline__ void step() {
        const int i = t + PHASE;
        if (EDGE) {
            if (i < n_in) process<PHASE, BASE>(pf_own[PHASE], pf_halo[PHASE], i);
        } else {
            dma_row(slot_in);                                    // row i + D on its way while row i is processed
            slot_in = slot_in + 1 == C::D + 1 ? 0 : slot_in + 1;
            // Row i's DMAs were issued D steps ago.  Vector-memory operations younger than them: D * NDMA DMAs, plus --
            // once the walk emits (one store per step from row 2R on) -- the D stores in between: waiting for exactly that
            // many leaves the full D rows in flight; before that, counting no stores is the safe side.
            if (i >= 2 * R + C::D) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C::D * (NDMA + 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C::D * NDMA) : "memory");
            process<PHASE, BASE>(pf_own[0], 0.0f, i);
            slot_out = slot_out + 1 == C::D + 1 ? 0 : slot_out + 1;
        }
    }

    // The lane index as a value the compiler cannot trace back to `lane`: addresses derived from it are then computed where they are
    // used -- in the rare NaN blocks -- instead of being hoisted out of the round loop as loop invariants, where there is no
    // register for them: they were SPILLED, and their reload in the rare block waits with `s_waitcnt vmcnt(0)`, which drains the
    // DMA ring (8 rows in flight) on every row that holds NaN: 25x25 mean at 0.1 % NaN 1.06 instead of 0.8x ms.
    __device__ __forceinline__ int lane_here() const {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    }

    // (NANOK) the input row of this step holds NaN: every lane overwrites the NaN among ITS cells of the row in the ring (staged
    // cells NC l .. NC l + NC - 1; the first 2 HL / NC lanes also the halo cells TW + NC l ..) with the shift -- a cell equal to
    // the shift adds nothing to any sum -- and notes their positions in the bitmap (bit s = staged cell s).  Then every lane
    // takes the bits of its own NV cells from it and adds, for each of the 2R+1 output rows this input row lies under, the
    // number of them inside that row's run to the lost ring: slot (step that completes the output row) mod K.  The runs are
    // compile-time masks; ~100 instructions and 2R+1 read-modify-writes of LDS on the rare row, one read at every output.
    __device__ __forceinline__ void mark_row(float *row, int i) {
        typedef float ldsNC __attribute__((ext_vector_type(NC)));
        const int ln = lane_here();
        unsigned *bm = nanmap;
        if (ln < NMW) bm[ln] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // (LDS serves one wave's instructions in order)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            if (part && ln >= 2 * HL / NC) break;
            const int s0 = (part ? TW : 0) + NC * ln;
            ldsNC *p = reinterpret_cast<ldsNC *>(row + s0);
            ldsNC v = *p;
            unsigned mine = 0u;
#pragma unroll
            for (int e = 0; e < NC; ++e)
                if (isnan(v[e])) { v[e] = c; mine |= 1u << e; }
            if (mine) {
                *p = v;
                atomicOr(&bm[s0 >> 5], mine << (s0 & 31));     // (NC cells at a multiple of NC: never across two words)
            }
        }
        spread_lost(i);
    }

    // (NANOK) the bitmap of the current row is complete: every lane takes the bits of its own NV cells and adds, for each of the
    // 2R+1 output rows this input row lies under, the number of them inside that row's run to the lost ring
    __device__ __forceinline__ void spread_lost(int i) {
        unsigned *bm = nanmap;
        const int ln = lane_here();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // (LDS serves one wave's instructions in order)
        const int s0 = NC * ln;
        const unsigned long long two = ((unsigned long long)bm[(s0 >> 5) + 1] << 32) | bm[s0 >> 5];
        const unsigned span = (unsigned)(two >> (s0 & 31));    // bit k: the lane's cell w[k] of this row is NaN
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // (the next marked row clears the bitmap)
        // lost counts are bytes: once the rows of this tile have shown a lane more than 255 NaN cells in all (the sum of the
        // rows' worst lanes: a scalar) the tile is handed on (dense nodata -- the NaN-aware walker is the faster one there
        // anyway); below that no byte of the ring can overflow
        span_total += wave_reduce<WrMax>(__popc(span & ((1u << NV) - 1u)));
        bad |= span_total > 255;
        // every distinct half-width once (both owned columns packed: byte o), then one LDS add per output row in flight; the ring
        // holds one 16-bit entry per lane, two lanes to a word: an atomic add of the entry shifted to the lane's half
        constexpr ShapeRows<R, Shape> T{};
        unsigned lvl[R + 1];
#pragma unroll
        for (int h = 0; h <= R; ++h) {
            lvl[h] = 0u;
            if (!C::level_used(h)) continue;
#pragma unroll
            for (int o = 0; o < NC; ++o) lvl[h] |= (unsigned)__popc((span >> (HL + o - h)) & ((2u << (2 * h)) - 1u)) << (8 * o);
            lvl[h] <<= 16 * (ln & 1);
        }
        unsigned *ring32 = reinterpret_cast<unsigned *>(lostring) + (ln >> 1);
#pragma unroll
        for (int j = 0; j < K; ++j) {                          // the output row completed j steps from now sees this row at offset R - j
            // (no test for the run-in here -- it was 25 scalar branches in this block: a step that completes no output row
            // reads and clears its slot like any other, process())
            const int slot = lost_slot + j < K ? lost_slot + j : lost_slot + j - K;
            __hip_atomic_fetch_add(ring32 + slot * 32, lvl[T.hw[j < R ? R - j : j - R]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }

    // (NANOK) NaN cells under the windows of the output row this step completes: the lane's entry of the lost ring, cleared
    // for the step that will use the slot next
    __device__ __forceinline__ void lost_cells(int (&lost)[NC]) {
        unsigned short *p = lostring + lost_slot * 64 + lane_here();
        const unsigned v = *p;
        *p = 0;
#pragma unroll
        for (int o = 0; o < NC; ++o) lost[o] = (int)((v >> (8 * o)) & 255u);
    }

    template <int PHASE, int BASE>   // BASE: SLIDE's ring slot of the round's first row (0 otherwise)
    __device__ __forceinline__ void process(const float (&q)[NC], float hq, int i) {
        constexpr int SLOT = (BASE + PHASE) % C::KR;
        const long yy = y_first + i;
        const bool row_in = !EDGE || (yy >= -(long)g.halo_top && yy < g.rows + g.halo_bot);  // wave-uniform
        if constexpr (NANOK) {
            inflight = (inflight << 1) & ((1u << K) - 1u);
            lost_slot = lost_slot + 1 == K ? 0 : lost_slot + 1;          // == i mod K (init: K - 1)
        }
        if (row_in) {
            typedef float ldsNC __attribute__((ext_vector_type(NC)));
            float w[NV];
            if (EDGE) {
                // ---- shifted row -> LDS, each lane reads back the NV cells under its NC windows
                float d[NC];
                float dh = hq - c;
#pragma unroll
                for (int e = 0; e < NC; ++e) d[e] = q[e] - c;
                if constexpr (NANOK) {
                    // (the interior walk's vote, on the cells in registers: NaN -> d = 0, positions into the bitmap)
                    bool nn = !isfinite(dh);
#pragma unroll
                    for (int e = 0; e < NC; ++e) nn |= !isfinite(d[e]);
                    if (__builtin_expect(__any(nn) != 0, 0)) {
                        bad |= isinf(dh);
#pragma unroll
                        for (int e = 0; e < NC; ++e) bad |= isinf(d[e]);
                        const int ln = lane_here();
                        unsigned *bm = nanmap;
                        if (ln < NMW) bm[ln] = 0u;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        unsigned mine = 0u;
#pragma unroll
                        for (int e = 0; e < NC; ++e)
                            if (isnan(d[e])) { d[e] = 0.0f; mine |= 1u << e; }
                        if (mine) atomicOr(&bm[(NC * ln) >> 5], mine << ((NC * ln) & 31));
                        if (isnan(dh)) {
                            dh = 0.0f;
                            if (ln < 2 * HL) atomicOr(&bm[(TW + ln) >> 5], 1u << ((TW + ln) & 31));
                        }
                        spread_lost(i);
                        inflight |= 1u;
                        saw_nan = true;
                        if (__popc(inflight) > 18) bad = true;         // (dense nodata: the NaN-aware walker is the faster one)
                    }
                    if (__builtin_expect(inflight != 0u, 0)) lost_cells(lost_o);        // (run-in steps too: what was added for them is dropped)
                }
#pragma unroll
                for (int e = 0; e + 1 < NC; e += 2) amax = amax3(amax, d[e], d[e + 1]);
                amax = amax3(amax, dh, 0.0f);
                float *row = lds;  // ONE row buffer: LDS serves a wave's instructions in order, so the next row's writes
                                   // (issued after this row's reads) cannot overtake them
                ldsNC dq;
#pragma unroll
                for (int e = 0; e < NC; ++e) dq[e] = d[e];
                *reinterpret_cast<ldsNC *>(row + NC * lane) = dq;
                row[TW + lane] = dh;                             // (lanes >= 2*HL: a slot nobody reads)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();                 // (LDS serves one wave's instructions in order)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int b = 0; b < NQ; ++b) {
                    const ldsNC v4 = *reinterpret_cast<const ldsNC *>(row + NC * lane + NC * b);
#pragma unroll
                    for (int e = 0; e < NC; ++e) w[NC * b + e] = v4[e];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
                // ---- the row landed in the ring (raw cells).  (NANOK) First a look at the lane's OWN cells of it (the NC it stages
                // for the wave, the first lanes also the halo cells): a wave-wide vote, and a row that holds NaN has them
                // overwritten with the shift in the ring and noted in the bitmap (mark_row) BEFORE anybody reads the row.  The vote
                // stands at the head of the step, where the instruction stream is cut anyway; voting on the prefix total behind
                // the read (free, but the row then has to be read again: a second copy of the read spilled 37 registers into the
                // round loop, a two-trip loop around one copy cut the step in the middle of its dependent chains: 25x25 mean
                // 0.59 -> 2.2 / 0.80 ms on a clean raster).
                float *row = lds + slot_out * C::RBF;
                if constexpr (NANOK) {
                    const int s1 = lane < 2 * HL / NC ? TW + NC * lane : NC * lane;     // (the other lanes look at their own cells twice)
                    const ldsNC a0 = *reinterpret_cast<const ldsNC *>(row + NC * lane), a1 = *reinterpret_cast<const ldsNC *>(row + s1);
                    bool nn = false;
#pragma unroll
                    for (int e = 0; e < NC; ++e) nn |= !isfinite(a0[e]) || !isfinite(a1[e]);
                    if (__builtin_expect(__any(nn) != 0, 0)) {        // rare, wave-uniform
#pragma unroll
                        for (int e = 0; e < NC; ++e) bad |= isinf(a0[e]) || isinf(a1[e]);      // +-inf: the tile is handed on
                        mark_row(row, i);
                        inflight |= 1u;
                        saw_nan = true;
                        if (__popc(inflight) > 18) bad = true;         // (dense nodata: the NaN-aware walker is the faster one)
                    }
                    if (__builtin_expect(inflight != 0u, 0)) lost_cells(lost_o);        // (run-in steps too: what was added for them is dropped)
                }
#pragma unroll
                for (int b = 0; b < NQ; ++b) {
                    const ldsNC v4 = *reinterpret_cast<const ldsNC *>(row + NC * lane + NC * b);
#pragma unroll
                    for (int e = 0; e < NC; ++e) w[NC * b + e] = v4[e] - c;
                }
                // largest |d| of the tile: the lanes' own cells cover the tile's columns, the first / last cells of the
                // first / last lanes its halo columns
#pragma unroll
                for (int e = 0; e + 1 < NC; e += 2) {
                    amax = amax3(amax, w[HL + e], w[HL + e + 1]);
                    amax = amax3(amax, w[e], w[e + 1]);
                    amax = amax3(amax, w[NV - NC + e], w[NV - NC + e + 1]);
                }
            }
            // ---- lane-local prefix sums: P[k] = w[0] + .. + w[k]; cell o's centre is w[HL + o]
#pragma unroll
            for (int k = 1; k < NV; ++k) w[k] += w[k - 1];
            // the lanes' totals cover every cell the tile read in this row: the convolution must not meet a non-finite cell at all, and
            // the plain walk of the mean / sum gives up at the first row that holds one (not 2R rows later, when the first poisoned
            // window sum comes out): the carrying walk behind it then starts all the sooner
            if (CONV || !NANOK) bad |= !isfinite(w[NV - 1]);
            if (C::SLIDE) {
                // input row i lives in slot i mod KR; the row that leaves the window, i - (2R+1), K slots behind
                constexpr int LEFT = ((SLOT - K) % C::KR + C::KR) % C::KR;
#pragma unroll
                for (int o = 0; o < NC; ++o) {
                    const int hi = HL + o + R, lo = HL + o - R - 1;
                    vsum[o] -= acc[LEFT][o];
                    acc[SLOT][o] = lo >= 0 ? w[hi] - w[lo] : w[hi];
                    vsum[o] += acc[SLOT][o];
                }
            }
            if constexpr (shape_has_hole<Shape>(R)) {
                // ---- annuli: every distinct ROW PATTERN once -- the centred run of half-width hw minus the one of half-width
                // hwi -- through compile-time tables (ShapeRows), as in mom_impl.h
                constexpr ShapeRows<R, Shape> T{};
                auto run = [&](int h, int o) -> float {      // the centred run of half-width h under owned column o (static h, o)
                    const int hi = HL + o + h, lo = HL + o - h - 1;
                    return lo >= 0 ? w[hi] - w[lo] : w[hi];
                };
#pragma unroll
                for (int d = 0; d <= R; ++d) {
                    if (T.pat[d] != d) continue;
                    float S[NC];
#pragma unroll
                    for (int o = 0; o < NC; ++o) {
                        S[o] = run(T.hw[d], o);
                        if (T.hwi[d] >= 0) S[o] -= run(T.hwi[d] >= 0 ? T.hwi[d] : 0, o);
                    }
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const int dy = j - R;
                        if (T.pat[dy < 0 ? -dy : dy] != d) continue;
                        const int idx = ((PHASE - dy) % K + K) % K;
#pragma unroll
                        for (int o = 0; o < NC; ++o) acc[idx][o] += S[o];
                    }
                }
            }
            // ---- every distinct half-width once, into the ring slots of the output rows that see this row with it
#pragma unroll
            for (int h = 0; h <= R && !C::SLIDE && !shape_has_hole<Shape>(R); ++h) {
                if (!C::level_used(h)) continue;
                float S[NC];
#pragma unroll
                for (int o = 0; o < NC; ++o) {
                    const int hi = HL + o + h, lo = HL + o - h - 1;
                    S[o] = lo >= 0 ? w[hi] - w[lo] : w[hi];
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int dy = j - R;
                    if (Shape::hw(R, dy < 0 ? -dy : dy) != h) continue;
                    const int idx = ((PHASE - dy) % K + K) % K;
#pragma unroll
                    for (int o = 0; o < NC; ++o) acc[idx][o] += S[o];
                }
            }
        } else {
            // (EDGE) a row outside the raster: it holds no NaN and its row sum is 0
            if constexpr (NANOK) {
                if (__builtin_expect(inflight != 0u, 0)) lost_cells(lost_o);        // (run-in steps too: what was added for them is dropped)
            }
            if constexpr (C::SLIDE) {
                constexpr int LEFT = ((SLOT - K) % C::KR + C::KR) % C::KR;
#pragma unroll
                for (int o = 0; o < NC; ++o) {
                    vsum[o] -= acc[LEFT][o];
                    acc[SLOT][o] = 0.0f;
                }
            }
        }
        // ---- the output row R rows up is complete
        constexpr int DONE = ((PHASE - R) % K + K) % K;
        if (i >= 2 * R) {
            const long yo = y0 + (i - 2 * R);
            const long xo = x_tile + NC * lane;
            float res[NC];
#pragma unroll
            for (int o = 0; o < NC; ++o) {
                float n = (float)C::NTAPS;
                if (EDGE && !CONV) {
                    const bool rows_in = yo - R >= -(long)g.halo_top && yo + R < g.rows + g.halo_bot;   // wave-uniform
                    n = rows_in ? n_full[o]
                                : (float)clipped_count<R, Shape>(yo, xo + o, -(long)g.halo_top, g.rows + g.halo_bot, g.cols);
                }
                const float s = C::SLIDE ? vsum[o] : acc[DONE][o];
                if (NANOK && __builtin_expect(inflight != 0u, 0)) {   // (wave-uniform) NaN rows under this output row's windows
                    const int lost = lost_o[o];
                    const float nf = n - (float)lost;                  // (n: the window's cells inside the raster)
                    // (a window without a valid cell: mean NaN, sum 0 -- numba nanmean / nansum of an empty window)
                    const float m = lost ? (nf > 0.0f ? fmaf(s, __builtin_amdgcn_rcpf(nf), c) : nan_f32())
                                         : (EDGE ? c + s / n : fmaf(s, 1.0f / (float)C::NTAPS, c));
                    res[o] = SUM ? fmaf(nf, c, s) : m;
                    bad |= !isfinite(s) || 2.0f * (float)lost > n;
                    if (!EDGE || xo + o < g.cols) mmin = fminf(mmin, fabsf(m));
                    continue;
                }
                if (CONV) {
                    // full windows only: NaN within R cells of the raster (or shard halo) edge
                    const bool full = !EDGE || (yo - R >= -(long)g.halo_top && yo + R < g.rows + g.halo_bot &&
                                                xo + o - R >= 0 && xo + o + R < g.cols);
                    const float m = fmaf(s, 1.0f / (float)C::NTAPS, c);
                    res[o] = full ? wgt * fmaf((float)C::NTAPS, c, s) : nan_f32();
                    bad |= !isfinite(s);
                    if (full) mmin = fminf(mmin, fabsf(m));
                    continue;
                }
                const float m = EDGE ? c + s / n : fmaf(s, 1.0f / (float)C::NTAPS, c);
                res[o] = SUM ? fmaf(n, c, s) : m;
                bad |= !isfinite(s);
                if (!EDGE || xo + o < g.cols) mmin = fminf(mmin, fabsf(m));
            }
            float *po = out + yo * g.ld_out + xo;
            if (!EDGE) {
                typedef float stNC __attribute__((ext_vector_type(NC), aligned(4)));
                stNC rq;
#pragma unroll
                for (int o = 0; o < NC; ++o) rq[o] = res[o];
#ifdef XRS_FLOOR_NO_STORES                                     // (tools/floor_probe.sh: the walk without its output stream)
                if (g.rows < 0)
#endif
                if constexpr (NC == 2) st_row_nt(out_row, 8u * (unsigned)lane, rq);
                else __builtin_nontemporal_store(rq, reinterpret_cast<stNC *>(out_row + NC * lane));
                out_row += g.ld_out;
            } else {
#pragma unroll
                for (int o = 0; o < NC; ++o)
                    if (xo + o < g.cols) po[o] = res[o];
            }
        }
        if (!C::SLIDE) {
#pragma unroll
            for (int o = 0; o < NC; ++o) acc[DONE][o] = 0.0f;
        }
    }

    template <int BASE, int... P>
    __device__ __forceinline__ void round(std::integer_sequence<int, P...>) {
        if (EDGE) (load_row(t + P, pf_own[P], pf_halo[P]), ...);      // edge tiles: all loads of the round first
        (step<P, BASE>(), ...);
        t += U;
        if constexpr (!C::SLIDE) {
            // the round started at row t with ring slot (j - t) mod K for output row j; the next one starts at t + U
            ring_rotate<K, U>(acc);
        } else if constexpr (BASE + U == C::KR) {
            // V afresh from the K newest slots (the ring's last row is the newest): a balanced tree, error <= 5 u |V|
            float v[K][NC];
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int o = 0; o < NC; ++o) v[j][o] = acc[C::KR - 1 - j][o];
#pragma unroll
            for (int n = K; n > 1; n = (n + 1) / 2)
#pragma unroll
                for (int j = 0; j < n / 2; ++j)
#pragma unroll
                    for (int o = 0; o < NC; ++o) v[j][o] += v[n - 1 - j][o];
#pragma unroll
            for (int o = 0; o < NC; ++o) vsum[o] = v[0][o];
        }
    }

    // SLIDE: all positions of a round in the ring, one after the other
    template <int... B>
    __device__ __forceinline__ void ring_cycle(std::integer_sequence<int, B...>) {
        constexpr auto phases = std::make_integer_sequence<int, U>{};
        (round<B * U>(phases), ...);
    }

    // true: every result of the tile is good; false: the caller redoes the tile with the float64 walker
    __device__ __forceinline__ bool run() {
        init();
        constexpr auto phases = std::make_integer_sequence<int, U>{};
        while (t < n_in) {
            if constexpr (C::SLIDE) ring_cycle(std::make_integer_sequence<int, C::KR / U>{});
            else round<0>(phases);
            if (__any(bad)) return false;                    // a non-finite cell: stop early
        }
        // error bound of the float32 sums (header): |delta mean| <= u * A * (K * (2 * NV^2 + K) + K * NTAPS) / n
        float a = amax, mm = mmin;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            a = fmaxf(a, __shfl_xor(a, sft));
            mm = fminf(mm, __shfl_xor(mm, sft));
        }
        constexpr float UNIT = 5.9604645e-8f;
        // SLIDE: 2R+1 row sums with 2 NV^2 u A each, the re-summation tree (5 u |V|) and 2 KR roundings of |V| <= NTAPS A
        // (annuli: a row with a hole is the difference of two runs -- four prefix values instead of two)
        constexpr int PV = shape_has_hole<Shape>(R) ? 4 : 2;
        constexpr float COEF = UNIT * (float)(C::SLIDE ? K * 2 * NV * NV + (5 + 2 * C::KR) * C::NTAPS
                                                       : K * (PV * NV * NV + K) + K * C::NTAPS) / (float)C::NTAPS * (EDGE ? 4.0f : 1.0f);
        // (windows that lost cells to nodata -- at most half of them -- divide the same rounding by a smaller count)
        return !__any(bad) && (COEF * (saw_nan ? 2.0f : 1.0f) * a <= 0.9e-5f * mm);
    }
};

template <int R, typename Shape, int MODE>
#ifndef XRS_WIDE_WAVES
#define XRS_WIDE_WAVES 3        // workgroups per CU = waves per SIMD (4 spills at radius 12: 2.2 ms)
#endif
__global__ void __launch_bounds__(256, XRS_WIDE_WAVES) focal_wide_kernel(const WideArgs a) {
    using C = WideCfg<R, Shape>;
    __shared__ __attribute__((aligned(16))) float lds_rows[4][C::LDS_WAVE];
    __shared__ unsigned nan_row[4][8];                         // per wave: the NaN bitmap of the row being marked (interior tiles)
    __shared__ unsigned short lost_ring[4][(MODE != WIDE_CONV && !shape_has_hole<Shape>(R)) ? C::K * 64 : 1];   // per wave: NaN cells under the windows in flight
    long ty, gx;                   // (rim tiles first: circle_walk.h)
    if (!RimFirst(a.groups_x, a.n_groups / a.groups_x, a.rim_first).locate(blockIdx.x, ty, gx)) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long x_tile = (gx * 4 + wv) * C::TW;
    const long y0 = ty * a.tile_rows;
    const WalkGeom &g = a.g;
    if (x_tile >= g.cols) return;
    const long y_end = y0 + a.tile_rows < g.rows ? y0 + a.tile_rows : g.rows;
    const bool interior = x_tile - C::HL >= 0 && x_tile + C::TW + C::HL <= g.cols && y0 - R >= -(long)g.halo_top &&
                          y_end + R <= g.rows + g.halo_bot && y_end - y0 == a.tile_rows;
    bool ok;
    if (interior) {
        WideWalk<R, Shape, false, MODE> w(g, a.out, lds_rows[wv], x_tile, y0, y_end, lane);
        w.wgt = (float)a.wgt;
        ok = w.run();
    } else {
        WideWalk<R, Shape, true, MODE> w(g, a.out, lds_rows[wv], x_tile, y0, y_end, lane);
        w.wgt = (float)a.wgt;
        ok = w.run();
    }
    if (ok || XRS_WIDE_NO_FALLBACK) return;
    constexpr bool SUM = MODE == WIDE_SUM;
    // inside a nodata region (every cell the tile sees is NaN): mean NaN, sum 0, convolution NaN -- nothing to walk
    if (walk_tile_all_nan<(C::TW + 2 * R + 63) / 64>(g, x_tile - R, x_tile + C::TW + R, y0 - R, y_end + R, lane)) {
        float *const planes[1] = {SUM ? nullptr : a.out};
        if (C::TW == 128 && x_tile + 128 <= g.cols) fill_tile128_nt(a.out, g.ld_out, x_tile, y0, y_end, lane, SUM ? 0.0f : nan_f32());
        else walk_fill_no_data(g, planes, 1, SUM ? a.out : nullptr, 0.0f, x_tile, x_tile + C::TW, y0, y_end, lane);
        return;
    }
    // NaN cells under a window (scattered nodata, a nodata region's rim): the SAME walk again, this time carrying them (CARRY:
    // vote on the staged cells, in-ring repair, lost ring in LDS -- ~1.3x a plain walk).  The plain walk in front of it stopped
    // at the first round that met a NaN, so a clean raster never pays for any of this.  +-inf, dense nodata, windows with fewer
    // than half their cells and ill-conditioned sums still fail here and go on to the walkers below.
    if constexpr (MODE != WIDE_CONV && !shape_has_hole<Shape>(R) && XRS_WIDE_CARRY) {
        bool ok2;
        if (interior) {
            WideWalk<R, Shape, false, MODE, true> w(g, a.out, lds_rows[wv], x_tile, y0, y_end, lane);
            w.nanmap = nan_row[wv];
            w.lostring = lost_ring[wv];
            ok2 = w.run();
        } else {
            WideWalk<R, Shape, true, MODE, true> w(g, a.out, lds_rows[wv], x_tile, y0, y_end, lane);
            w.nanmap = nan_row[wv];
            w.lostring = lost_ring[wv];
            ok2 = w.run();
        }
        if (ok2) return;
    }
    if (MODE == WIDE_CONV) {
        // a non-finite cell in reach, or sums too ill-conditioned for float32: the float64 conv walker (tap by tap, in the
        // reference's order, where a window holds a non-finite cell)
        for (int q = 0; q < C::NC; ++q) walk_conv_columns<R, Shape>(g, a.out, a.wgt, a.weights, x_tile + 64 * q, lane, y0, y_end);
        return;
    }
    // the slow tiles -- the rim of a nodata region, dense nodata -- are noted for focal_wide_rescue_kernel, which takes them apart
    // into half tiles x row bands for every wave of the chip (mom_impl.h has the same arrangement and the reasons)
    if (a.rescue) {
        unsigned idx = 0;
        if (lane == 0) idx = atomicAdd(a.rescue, 1u);
        idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
        if (idx < a.rescue_cap) {
            if (lane == 0) a.rescue[2 + idx] = (unsigned)((ty * a.groups_x + gx) * 4 + wv);
            return;
        }
    }
    // (no list, or a full one: in place.)
    // NaN cells under a window (nodata, the raster's edge): the NaN-aware float32 walker of mom_nan_walk.h without its
    // squares -- validity and counts carried with the sums, the shift trails the walk, the rounding of S bounded at the end
    // of the tile like this kernel's own -- 64 columns at a time.  What fails that (+-inf, values straddling zero), and
    // sums too ill-conditioned for float32 in the first place: the float64 column walker (NaN-skipping, counting; mean from
    // float64 sums, the sum with the reference's sequential float32 adds).
    const WalkOuts o = {SUM ? a.out : nullptr, nullptr, nullptr, nullptr, SUM ? nullptr : a.out, nullptr, nullptr};
    MomArgs ma;
    ma.g = g;
    ma.out_sum = SUM ? a.out : nullptr; ma.out_mean = SUM ? nullptr : a.out; ma.out_var = nullptr; ma.out_std = nullptr;
    for (int q = 0; q < C::NC; ++q) {
        if (x_tile + 64 * q >= g.cols) break;
#ifndef XRS_WIDE_NO_NANWALK
        {
            MomWalkN<R, Shape, SUM ? MOM_SUM : MOM_MEAN> w(ma, lds_rows[wv], x_tile + 64 * q, y0, y_end, lane);
            if (w.run()) continue;
        }
#endif
        if (!SUM) walk_columns<R, Shape, false, false, false, true, false>(g, o, x_tile + 64 * q, lane, y0, y_end);
        else walk_columns<R, Shape, true, true, false, false, false>(g, o, x_tile + 64 * q, lane, y0, y_end);
    }
}

// The tiles focal_wide_kernel noted: the NaN-aware float32 walker band by band, windows it flags (+-inf under them) recomputed
// one by one in float64, a band whose rounding bound fails as a whole through the float64 column walker.
template <int R, typename Shape, int MODE>
__global__ void __launch_bounds__(256, 2) focal_wide_rescue_kernel(const WideArgs a) {
    using C = WideCfg<R, Shape>;
    constexpr bool SUM = MODE == WIDE_SUM;
    __shared__ __attribute__((aligned(16))) float stage[4][2 * (64 + 2 * R)];
    __shared__ unsigned short fixes[4][1024];
    const unsigned count = a.rescue[0] < a.rescue_cap ? a.rescue[0] : a.rescue_cap;
    if (!count) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const WalkGeom &g = a.g;
    MomArgs ma;
    ma.g = g;
    ma.out_sum = SUM ? a.out : nullptr; ma.out_mean = SUM ? nullptr : a.out; ma.out_var = nullptr; ma.out_std = nullptr;
    ma.todo = nullptr; ma.rescue = nullptr; ma.rescue_cap = 0;
    const WalkOuts o = {SUM ? a.out : nullptr, nullptr, nullptr, nullptr, SUM ? nullptr : a.out, nullptr, nullptr};
    const long waves = (long)gridDim.x * 4;
    const long want_nb = waves / ((long)count * C::NC);
    const int max_nb = (a.tile_rows + 15) / 16;
    const int nb = want_nb < 1 ? 1 : want_nb > max_nb ? max_nb : (int)want_nb;
    const int band_rows = (a.tile_rows + nb - 1) / nb;
    const long items = (long)count * C::NC * nb;
    for (long it = (long)blockIdx.x * 4 + wv; it < items; it += waves) {
        const unsigned ent = a.rescue[2 + it / (C::NC * nb)];
        const int sub = (int)(it % (C::NC * nb)), q = sub / nb, band = sub % nb;
        const long grp = ent >> 2;
        const long ty = grp / a.groups_x, gx = grp % a.groups_x;
        const long xw = (gx * 4 + (long)(ent & 3u)) * C::TW + 64 * q;
        if (xw >= g.cols) continue;
        const long yt0 = ty * a.tile_rows;
        const long yt1 = yt0 + a.tile_rows < g.rows ? yt0 + a.tile_rows : g.rows;
        const long y0 = yt0 + (long)band * band_rows;
        const long y_end = y0 + band_rows < yt1 ? y0 + band_rows : yt1;
        if (y0 >= y_end) continue;
        MomWalkN<R, Shape, SUM ? MOM_SUM : MOM_MEAN> w(ma, stage[wv], xw, y0, y_end, lane);
        w.fix_list = fixes[wv];
        w.fix_cap = 1024;
        if (w.run()) {
            if (w.n_fix) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                mom_fix_cells<R, Shape>(ma, fixes[wv], w.n_fix, xw, y0, lane);
            }
        } else if (!SUM) {
            walk_columns<R, Shape, false, false, false, true, false>(g, o, xw, lane, y0, y_end);
        } else {
            walk_columns<R, Shape, true, true, false, false, false>(g, o, xw, lane, y0, y_end);
        }
    }
}

template <int R, typename Shape, int MODE>
int launch_wide_rescue(WideArgs &a, long tiles_y, hipStream_t s) {
    static thread_local int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    hipLaunchKernelGGL((focal_wide_rescue_kernel<R, Shape, MODE>), dim3((unsigned)(cus * 2)), dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <int R, typename Shape>
int launch_wide(WideArgs &a, float *out_mean, float *out_sum, hipStream_t s) {
    using C = WideCfg<R, Shape>;
    WalkGeom &g = a.g;
    g.tiles_x = (g.cols + C::TW - 1) / C::TW;
    static thread_local int wg_per_cu = 0;                     // (per instantiation: registers depend on the radius)
    if (!wg_per_cu) wg_per_cu = walk3_wg_per_cu(focal_wide_kernel<R, Shape, WIDE_MEAN>, XRS_WIDE_WAVES);
    a.tile_rows = C::nin(walk3_tile_base(g.rows, (g.tiles_x + 3) / 4, R, C::UU, wg_per_cu)) - 2 * R;
    const long tiles_y = (g.rows + a.tile_rows - 1) / a.tile_rows;
    g.n_tiles = g.tiles_x * tiles_y;
    a.groups_x = (g.tiles_x + 3) / 4;
    a.n_groups = a.groups_x * tiles_y;
    a.rim_first = RimFirst::mode_from_env();
    const long grid = RimFirst(a.groups_x, tiles_y, a.rim_first).grid();
    if (grid > 0x7fffffffL) return fail("focal mean: raster too large for one launch");
    a.rescue = mom_rescue_slot();
    a.rescue_cap = (unsigned)(g.tiles_x * tiles_y);
    if (a.rescue && mom_rescue_bytes(g.rows, g.cols) < 8 + 4 * (size_t)a.rescue_cap) a.rescue = nullptr;
    if (out_mean) {
        a.out = out_mean;
        if (a.rescue) XRS_HIP(hipMemsetAsync(a.rescue, 0, 8, s));
        hipLaunchKernelGGL((focal_wide_kernel<R, Shape, WIDE_MEAN>), dim3((unsigned)grid), dim3(256), 0, s, a);
        XRS_LAUNCH_CHECK();
        if (a.rescue)
            if (int rc = launch_wide_rescue<R, Shape, WIDE_MEAN>(a, tiles_y, s)) return rc;
    }
    if constexpr (!shape_has_hole<Shape>(R)) {                 // (annuli: the mean and the convolution only -- the entry refuses a sum)
        if (out_sum) {
            a.out = out_sum;
            if (a.rescue) XRS_HIP(hipMemsetAsync(a.rescue, 0, 8, s));
            hipLaunchKernelGGL((focal_wide_kernel<R, Shape, WIDE_SUM>), dim3((unsigned)grid), dim3(256), 0, s, a);
            XRS_LAUNCH_CHECK();
            if (a.rescue)
                if (int rc = launch_wide_rescue<R, Shape, WIDE_SUM>(a, tiles_y, s)) return rc;
        }
    }
    return 0;
}

template <int R, typename Shape>
int launch_wide_conv(WideArgs &a, float *out, const double *kernel, const double *weights_dev, hipStream_t s) {
    using C = WideCfg<R, Shape>;
    if (!is_uniform_shape<R, Shape>(kernel, &a.wgt)) return -1;
    WalkGeom &g = a.g;
    g.tiles_x = (g.cols + C::TW - 1) / C::TW;
    static thread_local int wg_per_cu = 0;                     // (per instantiation: registers depend on the radius)
    if (!wg_per_cu) wg_per_cu = walk3_wg_per_cu(focal_wide_kernel<R, Shape, WIDE_MEAN>, XRS_WIDE_WAVES);
    a.tile_rows = C::nin(walk3_tile_base(g.rows, (g.tiles_x + 3) / 4, R, C::UU, wg_per_cu)) - 2 * R;
    const long tiles_y = (g.rows + a.tile_rows - 1) / a.tile_rows;
    g.n_tiles = g.tiles_x * tiles_y;
    a.groups_x = (g.tiles_x + 3) / 4;
    a.n_groups = a.groups_x * tiles_y;
    a.weights = weights_dev;
    a.out = out;
    a.rim_first = RimFirst::mode_from_env();
    const long grid = RimFirst(a.groups_x, tiles_y, a.rim_first).grid();
    if (grid > 0x7fffffffL) return fail("convolve_2d: raster too large for one launch");
    hipLaunchKernelGGL((focal_wide_kernel<R, Shape, WIDE_CONV>), dim3((unsigned)grid), dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

#ifndef XRS_WIDE_ANNULUS_R
int dispatch_wide_conv(WideArgs &a, float *out, const double *kernel, const double *weights_dev, int r, hipStream_t s) {
    switch (r) {
#define XRS_WIDE_CASE(RR) case RR: return launch_wide_conv<RR, XRS_WIDE_SHAPE>(a, out, kernel, weights_dev, s);
#ifndef XRS_WIDE_PROBE
        XRS_WIDE_CASE(3) XRS_WIDE_CASE(4) XRS_WIDE_CASE(5) XRS_WIDE_CASE(6) XRS_WIDE_CASE(7) XRS_WIDE_CASE(8)
        XRS_WIDE_CASE(9) XRS_WIDE_CASE(10) XRS_WIDE_CASE(11)
#endif
        XRS_WIDE_CASE(12)
#undef XRS_WIDE_CASE
        default: return -1;
    }
}

int dispatch_wide(WideArgs &a, float *out_mean, float *out_sum, const double *kernel, int r, hipStream_t s) {
    switch (r) {
#define XRS_WIDE_CASE(RR) case RR: return is_shape<RR, XRS_WIDE_SHAPE>(kernel) ? launch_wide<RR, XRS_WIDE_SHAPE>(a, out_mean, out_sum, s) : -1;
#ifndef XRS_WIDE_PROBE
        XRS_WIDE_CASE(3) XRS_WIDE_CASE(4) XRS_WIDE_CASE(5) XRS_WIDE_CASE(6) XRS_WIDE_CASE(7) XRS_WIDE_CASE(8)
        XRS_WIDE_CASE(9) XRS_WIDE_CASE(10) XRS_WIDE_CASE(11)
#endif
        XRS_WIDE_CASE(12)
#undef XRS_WIDE_CASE
        default: return -1;
    }
}
#else
// annulus_kernel(1, 1, XRS_WIDE_ANNULUS_R, RI), 1 <= RI < R: one instantiation pair (mean, convolution) per inner radius, one
// translation unit per outer radius (as kxk_mom_ann*.hip)
template <int RI>
int wide_annulus_pair(WideArgs &a, float *out_mean, float *out_conv, const double *kernel, const double *weights_dev, int ri, hipStream_t s) {
    if constexpr (RI >= XRS_WIDE_ANNULUS_R) return -1;
    else {
        if (ri != RI) return wide_annulus_pair<RI + 1>(a, out_mean, out_conv, kernel, weights_dev, ri, s);
        if (out_conv) return launch_wide_conv<XRS_WIDE_ANNULUS_R, AnnulusShape<RI>>(a, out_conv, kernel, weights_dev, s);
        return is_shape<XRS_WIDE_ANNULUS_R, AnnulusShape<RI>>(kernel) ? launch_wide<XRS_WIDE_ANNULUS_R, AnnulusShape<RI>>(a, out_mean, nullptr, s) : -1;
    }
}
#endif

}  // namespace

namespace xrs {

#ifdef XRS_WIDE_ANNULUS_R
// Exactly one of out_mean / out_conv.  0 = launched, -1 = not annulus_kernel(1, 1, XRS_WIDE_ANNULUS_R, RI) (for out_conv: times
// one weight value), > 0 = error.  `weights_dev`: the kernel as float64 in device memory (convolution only).
int XRS_WIDE_ENTRY(const float *in, float *out_mean, float *out_conv, long rows, long cols, long ld_in, long ld_out,
                   const double *kernel, const double *weights_dev, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s) {
    if (krows != kcols || krows / 2 != XRS_WIDE_ANNULUS_R || !(krows & 1) || (!out_mean == !out_conv)) return -1;
    // (the inner radius as the mask draws it: the first selected cell of the centre row, whatever its weight)
    const int R = XRS_WIDE_ANNULUS_R;
    int ri = -1;
    while (ri + 1 <= R && kernel[R * krows + R + ri + 1] == 0.0) ++ri;
    if (ri < 1 || ri >= R) return -1;
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.g.in = in; a.g.rows = rows; a.g.cols = cols; a.g.ld_in = ld_in; a.g.ld_out = ld_out;
    a.g.halo_top = halo_top; a.g.halo_bot = halo_bot;
    return wide_annulus_pair<1>(a, out_mean, out_conv, kernel, weights_dev, ri, s);
}
#else
// 0 = launched, -1 = not this shape with a radius of 3..12 cells (caller takes another kernel), > 0 = error.
// (mean and sum together: two launches)
int XRS_WIDE_ENTRY(const float *in, float *out_mean, float *out_sum, long rows, long cols, long ld_in, long ld_out,
                   const double *kernel, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    if (!out_mean && !out_sum) return 0;
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.g.in = in; a.g.rows = rows; a.g.cols = cols; a.g.ld_in = ld_in; a.g.ld_out = ld_out;
    a.g.halo_top = halo_top; a.g.halo_bot = halo_bot;
    return dispatch_wide(a, out_mean, out_sum, kernel, krows / 2, s);
}

// convolve_2d with one weight value on this shape (normalised circle_kernel / np.ones): 0 = launched, -1 = not that,
// > 0 = error.  `weights_dev`: the kernel as float64 in device memory, for windows that hold a non-finite cell.
int XRS_WIDE_CONV_ENTRY(const float *in, float *out, long rows, long cols, long ld_in, long ld_out, const double *kernel,
                        const double *weights_dev, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.g.in = in; a.g.rows = rows; a.g.cols = cols; a.g.ld_in = ld_in; a.g.ld_out = ld_out;
    a.g.halo_top = halo_top; a.g.halo_bot = halo_bot;
    return dispatch_wide_conv(a, out, kernel, weights_dev, krows / 2, s);
}
#endif

}  // namespace xrs
