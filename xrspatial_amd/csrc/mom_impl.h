// Focal mean / var / std / sum over large circular / box masks in ONE pass -- focal_stats(agg, circle_kernel(...),
// ['mean', 'std', 'var', 'sum']) with 9x9 .. 25x25 windows (xrspatial/focal.py:782-797 runs one apply() pass per statistic,
// each gathering the window per cell for a numba reducer, :226-258: nanmean / nanvar / nanstd with float64 accumulators,
// nansum in the array dtype).
//
// Third generation of the moments walk (first: circle_walk.h WalkF64, second: walk2_impl.h's moments pass: float64 sums
// of values shifted by one constant per wave tile -- ~100 float64 instructions per cell and row, which on gfx950 issue
// at half the rate of v_add_f32 / v_mul_f32: experiments/valu_rate2.hip).  Here everything per row is a float32 add,
// subtract or multiply, and what float32 cannot hold is kept out of the sums by construction:
//   * the wide row walk of wide_impl.h: a wave owns 64 NC columns x ~130 output rows and walks down; rows arrive in a
//     private LDS ring by LDS-DMA (lds_dma.h), D rows ahead; a lane owns NC adjacent columns, reads the NC + 2 HL cells
//     under their windows, subtracts ITS shift c, squares, and builds lane-local prefix sums of w and w^2: every distinct
//     half-width of the mask is then ONE subtraction per cell and moment, added into the register ring of the 2R+1
//     output rows in flight (compile-time indices: U = 5 rows per unrolled round, ring rotated once per round);
//   * the shift TRAILS the walk.  One constant per tile makes sum w^2 ~ n (var + m^2) with m the distance between the
//     window mean and the shift: on a slope of g per cell m reaches 70 g at the end of a tile against var = 36 g^2 for
//     a radius-12 circle, and float32 loses 7 bits to the cancellation.  Instead every lane re-centres once per round:
//     c' = c + (mean of the widest runs of the round's five rows: 5 (2R+1) cells of the lane's own columns just behind
//     the walk; one row alone jitters enough on a noisy raster to trip the guard of 11x11 windows in 13 % of the tiles) -- and the 2R+1 partial sums move with it by exact algebra, S' = S - N d, Q' = Q - d (S + S'), N = the
//     (compile-time) number of cells the slot has seen and d = c' - c.  sum w^2 then stays within a small multiple of
//     n var: the emulation of these exact operations (experiments/f32_moments_emul.py) gives var within 8e-7 of a
//     float64 two-pass reference on the steep parity-stress DEM, 4e-7 on the benchmark DEM; with one shift per tile 5e-3;
//   * a guard per output decides whether float32 was good enough: with B = Q + n max(d^2 of the slot's re-centrings)
//     bounding every partial sum the slot went through, var is accepted if n var >= B / 5 (amplification <= 5: the
//     emulation and the GPU runs put the error at <= 2e-6 there; a bound of 10 let 1.5e-5 through on windows that straddle
//     a cliff, where several large re-centrings follow each other) and mean / sum
//     if mean^2 >= 0.04 B / n; otherwise -- flat windows next to relief, values
//     straddling zero, NaN / inf anywhere under a window (the comparisons fail on non-finite sums) -- the tile is
//     redone: an all-NaN tile is filled without a walk, a tile with NaN cells goes through the NaN-aware float32 walker of
//     mom_nan_walk.h (counts carried with the sums), and only what fails THAT guard -- +-inf, windows with a few close
//     valid cells at the rim of a nodata region, ill-conditioned sums -- reaches the exact float64 NaN-skipping walker of
//     circle_walk.h.  A window over equal cells passes with S = Q = 0 exactly (its shift has converged onto the value) or
//     is redone;
//   * sum = n c + S (one fma), mean = c + S / n, var = (Q - S^2 / n) / n, std = sqrt(var) in float32.
// Included by kxk_mom_circle.hip / kxk_mom_box.hip, which define XRS_MOM_SHAPE / XRS_MOM_ENTRY.
#include "mom_nan_walk.h"

namespace {

// number of in-raster cells under the window centred on (yo, x): rows [y_lo, y_hi), columns [0, cols)
template <int R, typename Shape>
__device__ __forceinline__ int mom_clipped_count(long yo, long x, long y_lo, long y_hi, long cols) {
    int n = 0;
    for (int dy = -R; dy <= R; ++dy) {
        const long yr = yo + dy;
        if (yr < y_lo || yr >= y_hi) continue;
        const int h = Shape::hw(R, dy < 0 ? -dy : dy), h0 = Shape::hwi(R, dy < 0 ? -dy : dy);
        const long a = x - h < 0 ? 0 : x - h, b = x + h > cols - 1 ? cols - 1 : x + h;
        n += b >= a ? (int)(b - a + 1) : 0;
        if (h0 >= 0) {                                       // the hole's cells inside the raster
            const long a0 = x - h0 < 0 ? 0 : x - h0, b0 = x + h0 > cols - 1 ? cols - 1 : x + h0;
            n -= b0 >= a0 ? (int)(b0 - a0 + 1) : 0;
        }
    }
    return n;
}

// EDGE = false: a full tile whose whole input window lies inside the raster (LDS-DMA ring, no predicates, trailing shift);
// EDGE = true: tiles at the raster / shard edge and partial tiles: predicated loads staged through LDS with NaN for the
// cells outside (a reader turns them into w = 0), divisors = the geometric count of in-raster cells under each window,
// ONE shift per lane for the whole tile (no re-centring: the number of cells a partial sum has seen is not a compile-time
// constant here).  The guard is the same, so on steep relief an edge tile is more likely to end in the exact walker.
#ifndef XRS_MOM_CARRY_ALWAYS
#define XRS_MOM_CARRY_ALWAYS 1    // the carrying walk's output rows: 1 = always through the lost ring (no branch), 0 = only under NaN rows
#endif
// CARRY (interior tiles, solid shapes): nodata carried by the walk itself -- the second walk of a tile whose first, plain walk
// met a NaN (the scheme of wide_impl.h's carrying walk).  At the head of a step the lanes vote on the cells of the row that has
// just landed in the ring; a row that holds NaN has them overwritten IN THE RING with `fill`, one finite value per tile, and
// their positions noted in a bitmap; every lane then takes the bits under its own windows and adds, for each of the 2R+1
// output rows the input row lies under, the number of them inside that row's run to a ring of lost counts in LDS.  From there
// on a nodata cell is an ordinary cell of value `fill`: the prefix sums, the ring, the re-centring with its compile-time
// cell counts are the plain walk's.  An output row whose windows lost L cells takes them out again: S -= L (fill - c),
// Q -= L (fill - c)^2, n = ntaps - L -- the sums are about the lane's CURRENT shift c, whatever it was when the cells went
// in.  The guard sees the Q that was summed (fill cells included: they are what the rounding happened on), so a fill value
// far from the window (a tile with more relief than 20 - 40 window standard deviations) fails it and the tile goes on
// to the NaN-aware walker as before; so do +-inf, dense nodata and windows with fewer than half their cells.
template <int R, typename Shape, int OM, bool EDGE, bool CARRY = false>
struct MomWalk {
    static constexpr bool NANOK = CARRY && !EDGE && !shape_has_hole<Shape>(R) && MomCfg<R, Shape>::NC == 2 &&
                                  MomCfg<R, Shape>::NV <= 32 && MomCfg<R, Shape>::CELLS <= 160;
    static constexpr int NMW = 6;  // words of the bitmap (CELLS <= 160 bits, + one the last lane's read may touch)
    static constexpr int NO = OM == 0 ? 1 : ((OM & 1) + (OM >> 1 & 1) + (OM >> 2 & 1) + (OM >> 3 & 1));
    __device__ __forceinline__ bool want(int bit, const float *p) const { return OM ? (OM & bit) != 0 : p != nullptr; }
    using C = MomCfg<R, Shape>;
    static constexpr int K = C::K, HL = C::HL, NV = C::NV, NQ = C::NQ, U = C::U, NC = C::NC, TW = C::TW, D = C::D;
    // Annuli.  A row that crosses the hole is the difference of two centred runs, and the hole's cells are under both: a cell far
    // from the shift THERE -- a spike at the very centre of the ring, an unmasked -32768 -- is no tap of the window, adds nothing
    // to its Q, and leaves ~u |its square| of rounding in each of the ~4 (2 RI + 1) additions and subtractions of the rows that
    // cross the hole (a random walk: its root).  hk bounds the hole's rows still under a window (their widest: the row through
    // the centre; a maximum decaying by 0.93 per round, >= 0.7 while the row matters); accepted while that random walk stays
    // below 3e-6 of n var and 2e-6 of n |mean|.  On ordinary relief hk ~ (2 RI + 1) sigma^2: far from binding.  (Solid shapes
    // have no such cells since a run is summed from the centre outwards, moment_pass.)
    static constexpr bool HOLE = shape_has_hole<Shape>(R);
    static constexpr int HOLE_HW = HOLE ? Shape::hwi(R, 0) : 0;
    static constexpr float HK_OPS = 4.0f * (float)(2 * HOLE_HW + 1);
    static constexpr float HK_DECAY = 0.93f;
    static constexpr float HK_VAR = 0.0284f * const_sqrt(HK_OPS);   // 2^-24 / (3e-6 x 0.7) x sqrt(ops)
    static constexpr float HK_MEAN = 1.27e-3f * HK_OPS;             // (2^-24 / 2e-6)^2 / 0.7 x ops
    // HK_MEAN if the mean or the sum is wanted, else 0 (from gm: no register of its own)
    __device__ __forceinline__ float gh() const { return gm * ((float)C::NTAPS * HK_MEAN / 0.04f); }

    float accS[K][NC], accQ[K][NC];
    float c, c_next;               // the lane's shift; sum of the round's widest runs about it (-> its successor)
    float dq_last, dq_old;         // d^2 of the last re-centring; decaying maximum of the older ones
    float dqn;                     // NTAPS * max of both: what the re-centrings added to the partial sums of squares
    float hk;                      // (annuli) decaying maximum of the sums of (v - c)^2 over the cells of the hole's widest row, see HK_VAR
    int slot_in, slot_out;
    unsigned ring_addr;
    float pf_own[EDGE ? U : 1][NC], pf_halo[EDGE ? U : 1];     // EDGE: the rows of the current round, loaded up front
    float n_full[NC];              // EDGE: cell count of a window whose rows are all inside, per owned column
    long y_end;
    int n_in;
    unsigned long long badm;       // lanes with an output that failed its guard (wave-uniform)
    float gm;                      // 0.04 / n if mean or sum is requested, else 0 (that guard always passes)
    int t;

    const MomArgs &a;
    const WalkGeom &g;
    float *lds;
    long x_tile, y0, y_first;
    int lane;
    const float *dma_src;          // interior: (wave-uniform) first staged cell of the next row to DMA; the rows are taken in order,
    int dma_adv;                   // so the pointer advances by a row per DMA (dma_adv more times: rows past the tile repeat the last)
    long out_off;                  // interior: offset of the wave tile's next output row in every plane
    // ---- CARRY
    float fill;                    // (wave-uniform) the value a nodata cell is replaced with in the ring
    unsigned inflight;             // (wave-uniform) bit b: input row t - b held NaN cells
    int lost_slot;                 // (wave-uniform) t mod K of the current step
    int span_total;                // (wave-uniform) NaN cells the rows of this tile have shown their worst lane, summed
    unsigned *nanmap;              // LDS: NMW words, the NaN bitmap of the row being marked: bit s = staged cell s is NaN
    unsigned short *lostring;      // LDS: [K][64] -- slot (step mod K), lane

    __device__ __forceinline__ MomWalk(const MomArgs &a_, float *lds_, long xt, long y0_, long ye, int lane_)
        : a(a_), g(a_.g), lds(lds_), x_tile(xt), y0(y0_), lane(lane_) { y_end = ye; }

    // EDGE: staged cell s <-> raster column x_tile - HL + s; lane owns s = NC*lane .. NC*lane+NC-1, the 2*HL halo cells
    // s = TW + lane come from the first 2*HL lanes.  NaN outside the raster; a NaN cell INSIDE it cannot be told from that
    // by the readers, so the loader reports it (the tile then goes to the exact walker, which skips and counts)
    __device__ __forceinline__ void load_row(int il, float (&own)[NC], float &halo) {
        const long yy = y_first + il;
        const long xs = x_tile - HL + NC * lane;
#pragma unroll
        for (int e = 0; e < NC; ++e) own[e] = nan_f32();
        halo = nan_f32();
        const bool row_ok = il < n_in && yy >= -(long)g.halo_top && yy < g.rows + g.halo_bot;     // wave-uniform
        if (!row_ok) return;
        const float *p = g.in + yy * g.ld_in;
        bool hole = false;
#pragma unroll
        for (int e = 0; e < NC; ++e)
            if (xs + e >= 0 && xs + e < g.cols) { own[e] = p[xs + e]; hole |= own[e] != own[e]; }
        const long xh = x_tile - HL + TW + lane;
        if (lane < 2 * HL && xh >= 0 && xh < g.cols) { halo = p[xh]; hole |= halo != halo; }
        badm |= __builtin_amdgcn_ballot_w64(hole);
    }

    __device__ __forceinline__ void dma_row(int slot) {
        const float *p = uniform_ptr(dma_src);
        dma_src += dma_adv > 0 ? g.ld_in : 0;
        --dma_adv;
        // (whole 16-byte pieces: CELLS = 86 -- one column per lane, radius 11: the annuli of kxk_mom_ann11.hip -- is not a multiple
        //  of four; until round 6 the count was rounded DOWN there and the row's last two cells, which only lanes 62 and 63
        //  read, kept whatever the ring slot held before.  tests/fuzz_parity.py --windows found it.)
        constexpr int QMAX = C::CELLS_DMA / 4 - 1;
        // (CARRY: the lane's DMA offset computed afresh per row -- four instructions -- instead of living in a register the
        // carrying walk does not have: radius 12 with four planes spilled exactly one)
        int ln = lane;
        if constexpr (NANOK) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        glds16_s(p, 16u * (unsigned)(ln < QMAX ? ln : QMAX), ring_addr + (unsigned)slot * (C::RBF * 4));
    }

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int o = 0; o < NC; ++o) { accS[j][o] = 0.0f; accQ[j][o] = 0.0f; }
        dq_last = dq_old = dqn = 0.0f;
        hk = 0.0f;
        badm = 0;
        gm = (want(MOM_MEAN, a.out_mean) || want(MOM_SUM, a.out_sum)) ? 0.04f / (float)C::NTAPS : 0.0f;
        t = 0;
        y_first = y0 - R;
        n_in = (int)(y_end - y0) + 2 * R;                // (interior tiles: a whole number of rounds)
        ring_addr = lds_addr(lds);
        if (EDGE) {
            // the one shift of an edge tile: the mean of the lane's own columns in eight rows spread over the tile (a single
            // cell of a noisy raster is off the window means by the noise itself, and sum w^2 pays for it: var + m^2)
            const long h8 = (y_end - y0) / 8;
            float acc = 0.0f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const long yr = y0 + (y_end - y0) / 16 + r * h8;
#pragma unroll
                for (int o = 0; o < NC; ++o) {
                    const long xc = x_tile + NC * lane + o < g.cols ? x_tile + NC * lane + o : g.cols - 1;
                    acc += g.in[yr * g.ld_in + xc];
                }
            }
            acc *= 1.0f / (float)(8 * NC);
            c = isfinite(acc) ? acc : 0.0f;
            c_next = c;
#pragma unroll
            for (int o = 0; o < NC; ++o)
                n_full[o] = (float)mom_clipped_count<R, Shape>(0, x_tile + NC * lane + o, -(long)R, (long)R + 1, g.cols);
            return;
        }
        // first shift: the mean of four cells of the lane's own columns in the first two rows (any finite value works and
        // the first re-centring replaces it before any output row is complete; a single noisy cell would make that first
        // step large enough to trip the guard of the tile's first output rows)
        const float *p0 = g.in + y_first * g.ld_in + x_tile + NC * lane;
        const float c0 = 0.25f * ((p0[0] + p0[NC - 1]) + (p0[g.ld_in] + p0[g.ld_in + NC - 1]));
        c = isfinite(c0) ? c0 : 0.0f;
        if constexpr (NANOK) {
            // the fill value: the cell at the tile centre, or the first finite one of the 64 to its left
            const long yc = y0 + (y_end - y0) / 2, xc = x_tile + TW / 2;
            const float cand = g.in[yc * g.ld_in + (xc - lane >= 0 ? xc - lane : 0)];
            const unsigned long long fin = __ballot(isfinite(cand));
            fill = fin ? __shfl(cand, __builtin_ctzll(fin)) : 0.0f;
            fill = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fill)));
            if (!isfinite(c0)) c = fill;
            inflight = 0u;
            lost_slot = K - 1;
            span_total = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) lostring[j * 64 + lane] = 0;
        }
        c_next = c;
        dma_src = uniform_ptr(g.in + y_first * g.ld_in + (x_tile - HL));
        dma_adv = n_in - 1;
        out_off = y0 * g.ld_out + x_tile;
        for (int r = 0; r < D; ++r) dma_row(r);
        slot_in = D;
        slot_out = 0;
    }

    typedef float ldsNC __attribute__((ext_vector_type(NC)));
    typedef float stNC __attribute__((ext_vector_type(NC), aligned(4)));

    // NC results per lane to the wave-uniform row address `p` (streaming store, scalar base + 32-bit lane offset: the
    // compiler otherwise keeps a 64-bit lane address per output plane alive across the walk)
    __device__ __forceinline__ void store_row(float *p, stNC v) const {
#ifdef XRS_FLOOR_NO_STORES                                     // (tools/floor_probe.sh: the walk without its output streams)
        if (g.rows >= 0) return;
#endif
        const unsigned lane_b = (unsigned)(NC * 4) * (unsigned)(NANOK ? lane_here() : lane);
        if (NC == 2) { lds_dma_v2f q; q[0] = v[0]; q[1] = v[NC - 1]; st_row_nt(uniform_ptr(p), lane_b, q); }
        else if (NC == 1) st_row_nt(uniform_ptr(p), lane_b, v[0]);
        else __builtin_nontemporal_store(v, reinterpret_cast<stNC *>(reinterpret_cast<char *>(p) + lane_b));
    }

    // (CARRY) The lane index as a value the compiler cannot trace back to `lane`: addresses derived from it are computed where
    // they are used -- in the rare NaN blocks -- instead of being hoisted out of the round loop as loop invariants, where
    // there is no register for them (wide_impl.h: a spilled address is reloaded behind `s_waitcnt vmcnt(0)`, which drains
    // the DMA ring)
    // -- and computed afresh rather than copied: nothing of it stays live across the loop
    __device__ __forceinline__ int lane_here() const {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    }

    // (CARRY) the input row of this step holds NaN: every lane overwrites the NaN among ITS cells of the row in the ring (staged
    // cells NC l .. NC l + NC - 1; the first 2 HL / NC lanes also the halo cells TW + NC l ..) with `fill` and notes their
    // positions in the bitmap (bit s = staged cell s); then the lost counts of the 2R+1 output rows under it
    __device__ __forceinline__ bool mark_row(float *row) {
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
                if (isnan(v[e])) { v[e] = fill; mine |= 1u << e; }
            if (mine) {
                *p = v;
                atomicOr(&bm[s0 >> 5], mine << (s0 & 31));     // (NC cells at a multiple of NC: never across two words)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int s0 = NC * ln;
        const unsigned long long two = ((unsigned long long)bm[(s0 >> 5) + 1] << 32) | bm[s0 >> 5];
        const unsigned span = (unsigned)(two >> (s0 & 31));    // bit k: the lane's cell w[k] of this row is NaN
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // (the next marked row clears the bitmap)
        // lost counts are bytes: once the rows of this tile have shown a lane more than 255 NaN cells in all (the sum of the
        // rows' worst lanes: a scalar) the tile is handed on (dense nodata); below that no byte of the ring can overflow
        span_total += wave_reduce<WrMax>(__popc(span & ((1u << NV) - 1u)));
        // every distinct half-width once (both owned columns packed: byte o), then one LDS add per output row in flight; the
        // ring holds one 16-bit entry per lane, two lanes to a word: an atomic add of the entry shifted to the lane's half
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
            // (no test for the run-in here -- 25 scalar branches in this block: a step that completes no output row clears
            // its slot instead, step())
            const int slot = lost_slot + j < K ? lost_slot + j : lost_slot + j - K;
            __hip_atomic_fetch_add(ring32 + slot * 32, lvl[T.hw[j < R ? R - j : j - R]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return span_total > 255;
    }

    // (CARRY) NaN cells under the windows of the output row this step completes: the lane's entry of the lost ring, cleared
    // for the step that will use the slot next
    __device__ __forceinline__ unsigned lost_cells() {
        unsigned short *p = lostring + lost_slot * 64 + lane_here();
        const unsigned v = *p;
        unsigned zero;                                         // (a fresh zero: a constant one would live in a register across the loop)
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
        *p = (unsigned short)zero;
        return v;
    }

    template <int PHASE, bool SQ>
    __device__ __forceinline__ void moment_pass(unsigned row_addr) {
        float w[NV];
        typedef __attribute__((address_space(3))) const ldsNC lds_cvec;
        lds_cvec *row = (lds_cvec *)(size_t)row_addr;
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
            const ldsNC v = row[b];
#pragma unroll
            for (int e = 0; e < NC; ++e) {
                float d = v[e] - c;
                if (EDGE) d = d == d ? d : 0.0f;                  // a cell outside the raster
                w[NC * b + e] = SQ ? d * d : d;
            }
        }
        // Running sums FROM THE CENTRE OUTWARDS: w[j] = cells j .. HL for j <= HL, cells HL+1 .. j for j > HL, so that the centred
        // run of half-width h under owned column o is w[HL+o-h] + w[HL+o+h] -- a sum over the run's own cells and nothing else.
        // (Until round 6: one prefix sum from the left, a run = the difference of two prefixes.  A prefix also holds the cells
        // to the LEFT of the run, and a cell far from the shift there -- the last column of a plateau 1e5 above the window, a lake
        // at 1.6e7 -- is in no row of the window, adds nothing to its Q, and still left ~u |prefix| of rounding in each of the
        // window's ~NV K additions and subtractions: var off by up to 5 %, mean by 1e-4, with the guard none the wiser -- the
        // windows that DO contain the far cell pass it honestly, their variance is large.  tests/fuzz_parity.py --windows,
        // cliffs one column outside a window.  Same number of additions, no subtraction, and nothing to guard.)
        static_assert(NC <= 2, "two chains serve the centres HL and HL + 1");
#pragma unroll
        for (int j = HL - 1; j >= 0; --j) w[j] += w[j + 1];
#pragma unroll
        for (int j = HL + 2; j < NV; ++j) w[j] += w[j - 1];
        auto run_sum = [&](int h, int o) -> float {            // (static h, o; h = 0: the centre cell itself, both chains start there)
            return h == 0 ? w[HL + o] : w[HL + o - h] + w[HL + o + h];
        };
        if constexpr (!shape_has_hole<Shape>(R)) {
            // every distinct half-width once, into the ring slots of the output rows that see this row with it
#pragma unroll
            for (int h = 0; h <= R; ++h) {
                if (!C::level_used(h)) continue;
                float S[NC];
#pragma unroll
                for (int o = 0; o < NC; ++o) S[o] = run_sum(h, o);
                if (!EDGE && !SQ && h == R) c_next = PHASE == 0 ? S[0] : c_next + S[0];       // (hw(0) == R for every shape)
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int dy = j - R;
                    if (Shape::hw(R, dy < 0 ? -dy : dy) != h) continue;
                    const int idx = ((PHASE - dy) % K + K) % K;
#pragma unroll
                    for (int o = 0; o < NC; ++o) {
                        if (SQ) accQ[idx][o] += S[o];
                        else accS[idx][o] += S[o];
                    }
                }
            }
        } else {
            // a shape with a hole (annuli): every distinct ROW PATTERN once -- the centred run of half-width hw minus the one of
            // half-width hwi -- through compile-time tables (ShapeRows: evaluated once, not per loop iteration)
            constexpr ShapeRows<R, Shape> T{};
            // (the hole's cells are still under both runs of a row pattern: a far cell INSIDE the hole reaches the sums)
            auto run = [&](int h, int o) -> float { return run_sum(h, o); };
            if (SQ) hk = fmaxf(hk, NC == 1 ? run(HOLE_HW, 0) : run(HOLE_HW, 0) + run(HOLE_HW, NC - 1));
            if (!EDGE && !SQ) {                                // (the widest run is nobody's level here: its own subtraction)
                const float W = run(R, 0);
                c_next = PHASE == 0 ? W : c_next + W;
            }
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
                    for (int o = 0; o < NC; ++o) {
                        if (SQ) accQ[idx][o] += S[o];
                        else accS[idx][o] += S[o];
                    }
                }
            }
        }
    }

    template <int PHASE>
    __device__ __forceinline__ void step() {
        const int i = t + PHASE;
        unsigned row;                  // LDS byte address of the lane's first cell in the staged row
        if (EDGE) {
            if (i >= n_in) return;
            float *stage = lds;    // ONE row buffer: LDS serves a wave's instructions in order, so the next row's writes
                                   // (issued after this row's reads) cannot overtake them
            ldsNC q;
#pragma unroll
            for (int e = 0; e < NC; ++e) q[e] = pf_own[PHASE][e];
            *reinterpret_cast<ldsNC *>(stage + NC * lane) = q;
            stage[TW + lane] = pf_halo[PHASE];                   // (lanes >= 2*HL: a slot nobody reads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            row = ring_addr + (unsigned)(NC * 4) * (unsigned)lane;
        } else {
            dma_row(slot_in);
            slot_in = slot_in + 1 == D + 1 ? 0 : slot_in + 1;
            // row i was issued D steps ago; younger: D DMAs and -- once the walk emits, from row 2R on -- NO stores per step
            if (i >= 2 * R + D) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D * (1 + NO)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D) : "memory");
            row = ring_addr + (unsigned)slot_out * (C::RBF * 4) + (unsigned)(NC * 4) * (unsigned)(NANOK ? lane_here() : lane);
            if constexpr (NANOK) {
                // ---- the row landed in the ring (raw cells): a look at the lane's OWN cells of it (the NC it stages for the wave,
                // the first lanes also the halo cells), a wave-wide vote, and a row that holds NaN is repaired and counted
                // BEFORE anybody reads it
                inflight = (inflight << 1) & ((1u << K) - 1u);
                lost_slot = lost_slot + 1 == K ? 0 : lost_slot + 1;          // == i mod K (init: K - 1)
                float *rowp = lds + slot_out * C::RBF;
                typedef __attribute__((address_space(3))) const ldsNC lds_cvec;
                const int ln = lane_here();
                const unsigned halo_b = ln < 2 * HL / NC ? 4u * (unsigned)TW : 0u;     // (the other lanes look at their own cells twice)
                const ldsNC a0 = *(lds_cvec *)(size_t)row, a1 = *(lds_cvec *)(size_t)(row + halo_b);
                bool nn = false;
#pragma unroll
                for (int e = 0; e < NC; ++e) nn |= !isfinite(a0[e]) || !isfinite(a1[e]);
                if (__builtin_expect(__any(nn) != 0, 0)) {        // rare, wave-uniform
                    bool inf = false;
#pragma unroll
                    for (int e = 0; e < NC; ++e) inf |= isinf(a0[e]) || isinf(a1[e]);          // +-inf: the tile is handed on
                    inflight |= 1u;
                    // (dense nodata: the NaN-aware walker is the faster one)
                    const bool stop = mark_row(rowp) || __popc(inflight) > 18 || __any(inf);
                    badm |= (unsigned long long)__builtin_amdgcn_readfirstlane(stop ? 1 : 0);   // (a scalar: the verdicts stay in SGPRs)
                }
            }
            slot_out = slot_out + 1 == D + 1 ? 0 : slot_out + 1;
        }

        // ---- the NV cells under the lane's windows about the lane's shift, lane-local prefix sums, and every distinct
        // half-width once into the ring slots of the output rows that see this row with it: first the values, then their
        // squares.  (The compiler merges the two passes' reads and subtractions.  Forcing a second set of reads with a
        // memory barrier between the passes -- XRS_MOM_SPLIT_READS -- should need 26 registers less; this compiler then
        // spills 53 instead.)
        asm volatile("" : "+v"(row));                         // (one base register + immediate offsets for the reads)
#ifdef XRS_FLOOR_NO_ARITH                                      // (tools/floor_probe.sh: the DMA ring and the stores, no reads / sums)
        if (g.rows < 0)
#endif
        moment_pass<PHASE, false>(row);
#ifdef XRS_MOM_SPLIT_READS
        asm volatile("" ::: "memory");
#endif
#ifndef XRS_MOM_T_NOQ
#ifdef XRS_FLOOR_NO_ARITH
        if (g.rows < 0)
#endif
        moment_pass<PHASE, true>(row);
#endif

        if (EDGE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        // ---- the output row R rows up is complete
        constexpr int DONE = ((PHASE - R) % K + K) % K;
        if (EDGE && i >= 2 * R) {
            const long yo = y0 + (i - 2 * R);
            const long xo = x_tile + NC * lane;
            const bool rows_in = yo - R >= -(long)g.halo_top && yo + R < g.rows + g.halo_bot;       // wave-uniform
#pragma unroll
            for (int o = 0; o < NC; ++o) {
                const float n = rows_in ? n_full[o]
                                        : (float)mom_clipped_count<R, Shape>(yo, xo + o, -(long)g.halo_top, g.rows + g.halo_bot, g.cols);
                const float S = accS[DONE][o], Q = accQ[DONE][o];
                const float ms = S / n;
                const float mean = c + ms;
                const float e = Q - S * ms;
                const bool live = xo + o < g.cols;
                badm |= __builtin_amdgcn_ballot_w64(live && !(e >= 0.2f * Q));
                badm |= __builtin_amdgcn_ballot_w64(live && !(mean * mean * n >= (gm * (float)C::NTAPS) * Q));
                if (HOLE) badm |= __builtin_amdgcn_ballot_w64(live && (!(e >= HK_VAR * hk) || !((mean * n) * (mean * n) >= gh() * hk)));
                const float var = e / n;
                if (live) {
                    const long off = yo * g.ld_out + xo + o;
                    if (want(MOM_MEAN, a.out_mean)) a.out_mean[off] = mean;
                    if (want(MOM_VAR, a.out_var)) a.out_var[off] = var;
                    if (want(MOM_STD, a.out_std)) a.out_std[off] = sqrtf(var);
                    if (want(MOM_SUM, a.out_sum)) a.out_sum[off] = fmaf(n, c, S);
                }
            }
        }
        if (!EDGE && i >= 2 * R) {
            // (row base in scalar registers + one 32-bit lane offset; written as inline asm because the compiler otherwise
            // keeps a 64-bit lane address per output plane alive across the whole walk: 8 registers this kernel does not have)
            const long rowoff = out_off;
            out_off += g.ld_out;
            constexpr float inv = 1.0f / (float)C::NTAPS;
            stNC r_sum, r_mean, r_var, r_std;
            // (wave-uniform) NaN rows under this output row's windows.  (Marked likely -- at 0.1 % nodata it is 97 % of the output
            // rows -- the block moves in line and the round loop spills 10 registers; out of line it is the block that spills,
            // the one of the round's last step only: one column at a time there, below)
            if (NANOK && (XRS_MOM_CARRY_ALWAYS || __builtin_expect(inflight != 0u, 0))) {
                const unsigned lost_pk = lost_cells();
                const float dl = fill - c, dl2 = dl * dl;
#pragma unroll
                for (int o = 0; o < NC; ++o) {
                    const unsigned lost = (lost_pk >> (8 * o)) & 255u;
                    const float L = (float)lost;
                    const float n = (float)C::NTAPS - L;
                    // (v_rcp_f32, 1 ulp, for every window -- choosing the plain walk's constant for windows that lost nothing costs
                    // a compare whose zero wants a vector register, and there is none to spare: radius 12 with four planes spilled it)
                    const float rn = __builtin_amdgcn_rcpf(n);
                    const float Qa = accQ[DONE][o];                            // as summed: what the rounding happened on
                    const float S = fmaf(-L, dl, accS[DONE][o]), Q = fmaf(-L, dl2, Qa);
                    const float ms = S * rn;
                    const float mean = c + ms;
                    const float e = Q - S * ms;
                    const float B = Qa + dqn;
                    badm |= __builtin_amdgcn_ballot_w64(!(e >= 0.2f * B));
                    badm |= __builtin_amdgcn_ballot_w64(!(mean * mean * n >= (gm * (float)C::NTAPS) * B));
                    badm |= __builtin_amdgcn_ballot_w64(L > 0.5f * (float)C::NTAPS);
                    const float var = e * rn;
                    r_mean[o] = mean;
                    r_var[o] = var;
                    r_std[o] = __builtin_amdgcn_sqrtf(var);           // (v_sqrt_f32, 1 ulp: sqrtf's correction steps want five more registers)
                    r_sum[o] = fmaf(n, c, S);
                    __builtin_amdgcn_sched_barrier(0);             // (the columns one after the other: fewer values alive at once)
                }
            } else
#pragma unroll
            for (int o = 0; o < NC; ++o) {
                const float S = accS[DONE][o], Q = accQ[DONE][o];
                const float ms = S * inv;
                const float mean = c + ms;
                const float e = Q - S * ms;                   // n * variance
                const float B = Q + dqn;                      // bounds every partial sum of squares of this output row
                // (negated comparisons: a non-finite sum fails them)
                // (ballots: the verdicts stay in scalar registers)
                badm |= __builtin_amdgcn_ballot_w64(!(e >= 0.2f * B));
                badm |= __builtin_amdgcn_ballot_w64(!(mean * mean >= gm * B));
                if (HOLE) badm |= __builtin_amdgcn_ballot_w64(!(e >= HK_VAR * hk) || !(mean * mean >= (gh() * inv * inv) * hk));
                const float var = e * inv;
                r_mean[o] = mean;
                r_var[o] = var;
                r_std[o] = sqrtf(var);
                r_sum[o] = fmaf((float)C::NTAPS, c, S);
            }
            if (want(MOM_MEAN, a.out_mean)) store_row(a.out_mean + rowoff, r_mean);
            if (want(MOM_VAR, a.out_var)) store_row(a.out_var + rowoff, r_var);
            if (want(MOM_STD, a.out_std)) store_row(a.out_std + rowoff, r_std);
            if (want(MOM_SUM, a.out_sum)) store_row(a.out_sum + rowoff, r_sum);
        }
        if constexpr (NANOK) {
            // (the run-in: what NaN rows added for a step that completes no output row is dropped here, before the slot comes round again)
            if (i < 2 * R) (void)lost_cells();
        }
#pragma unroll
        for (int o = 0; o < NC; ++o) { accS[DONE][o] = 0.0f; accQ[DONE][o] = 0.0f; }
    }

    // the partial sums of every output row in flight, moved from shift c to c_next (ring already rotated: slot idx will be
    // hit next at offset -idx or K - idx and has seen C::seen(idx) cells)
    template <int... J>
    __device__ __forceinline__ void recentre(std::integer_sequence<int, J...>) {
        c_next = fmaf(c_next, 1.0f / (float)(K * U), c);      // (c_next was the sum of the round's widest runs about c)
        const float d = c_next - c;                           // exact: the step the walk really takes
        auto one = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr float N = (float)C::seen(j);
            if (C::seen(j) == 0) return;
#pragma unroll
            for (int o = 0; o < NC; ++o) {
                const float S = accS[j][o];
                const float S2 = S - N * d;
                accQ[j][o] -= d * (S + S2);
                accS[j][o] = S2;
            }
        };
        (one(std::integral_constant<int, J>{}), ...);
        c = c_next;
#ifndef XRS_MOM_T_NONANSTOP
        badm |= __builtin_amdgcn_ballot_w64(c != c);             // a NaN under the round's runs: stop now, not 2R rows later
#endif
        if constexpr (NANOK) {
            // (the carrying walk has no registers for the two-term history: ONE decaying maximum, x 0.85 per round -- 0.85, 0.72,
            // 0.61, 0.52: never below MomCfg's table either, a little more conservative than the plain walk's)
            dqn = fmaxf((float)C::NTAPS * (d * d), dqn * C::HIST_FIRST);
        } else {
            dq_old = fmaxf(dq_last * C::HIST_FIRST, dq_old * C::HIST_DECAY);
            dq_last = d * d;
            dqn = (float)C::NTAPS * fmaxf(dq_last, dq_old);
        }
    }

    template <int... P>
    __device__ __forceinline__ void round(std::integer_sequence<int, P...>) {
        if (EDGE) (load_row(t + P, pf_own[P], pf_halo[P]), ...);      // edge tiles: all loads of the round first
        // (CARRY: the re-centring at the head of the NEXT round instead of behind this round's last step -- the same place in the
        // walk, another place in the code: behind the last step its arithmetic was scheduled into that step's out-of-line nodata
        // output block, which then spilled 20 registers)
        if (NANOK && t > 0) recentre(std::make_integer_sequence<int, K>{});
        (step<P>(), ...);
        ring_rotate<K, U>(accS);
        ring_rotate<K, U>(accQ);
        t += U;
        if (HOLE) hk *= HK_DECAY;
#ifndef XRS_MOM_T_NORECENTRE
        if (!EDGE && !NANOK) recentre(std::make_integer_sequence<int, K>{});
#endif
    }

    // true: every result of the tile is good; false: the caller redoes the tile with the float64 walker
    __device__ __forceinline__ bool run() {
        init();
        while (t < n_in) {
            round(std::make_integer_sequence<int, U>{});
            if (badm) return false;
        }
        return true;
    }
};


// raster edges, non-finite cells under a window, sums too ill-conditioned for float32: the exact float64 column walker
// (NaN-skipping, counting, the reference's two-pass variance where it matters), 64 columns at a time.  
template <int R, typename Shape>
__device__ __forceinline__ void mom_exact_tile(const MomArgs &a, long x_tile, int lane, long y0, long y_end, int q_first = 0, int q_count = -1) {
    using C = MomCfg<R, Shape>;
    const WalkGeom &g = a.g;
    const WalkOuts o = {a.out_sum, nullptr, nullptr, nullptr, a.out_mean, a.out_var, a.out_std};
    for (int q = q_first; q < (q_count < 0 ? C::NC : q_first + q_count); ++q) {
        if (a.out_mean || a.out_var || a.out_std) {
            if (a.out_var || a.out_std) walk_columns<R, Shape, false, false, false, true, true>(g, o, x_tile + 64 * q, lane, y0, y_end);
            else walk_columns<R, Shape, false, false, false, true, false>(g, o, x_tile + 64 * q, lane, y0, y_end);
        }
        if (a.out_sum) walk_columns<R, Shape, true, true, false, false, false>(g, o, x_tile + 64 * q, lane, y0, y_end);
    }
}

#ifndef XRS_MOM_CARRY
#define XRS_MOM_CARRY 1           // NaN tiles: the carrying walk before the NaN-aware one-column walker
#endif
#ifndef XRS_MOM_WAVES
#define XRS_MOM_WAVES 2           // workgroups per CU = waves per SIMD
#endif
template <int R, typename Shape, int OM>
__global__ void __launch_bounds__(256, XRS_MOM_WAVES) focal_mom_kernel(const MomArgs a) {
    using C = MomCfg<R, Shape>;
    __shared__ __attribute__((aligned(16))) float lds_rows[4][(C::D + 1) * C::RBF];
    constexpr bool CARRIES = XRS_MOM_CARRY && OM != 0 && MomWalk<R, Shape, OM, false, true>::NANOK;
    __shared__ unsigned nan_row[4][8];                         // per wave: the NaN bitmap of the row being marked
    __shared__ unsigned short lost_ring[4][CARRIES ? C::K * 64 : 1];   // per wave: NaN cells under the windows in flight
    long ty, gx;
    if (!RimFirst(a.groups_x, a.n_groups / a.groups_x, a.rim_first).locate(blockIdx.x, ty, gx)) return;
    if (std::is_same<Shape, BoxShape>::value && a.todo && !a.todo[ty * a.groups_x + gx]) return;   // (boxsep.hip did this tile)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long x_tile = (gx * 4 + wv) * C::TW;
    const long y0 = ty * a.tile_rows;
    const WalkGeom &g = a.g;
    if (x_tile >= g.cols) return;
    const long y_end = y0 + a.tile_rows < g.rows ? y0 + a.tile_rows : g.rows;
    const bool interior = x_tile - C::HL >= 0 && x_tile - C::HL + C::CELLS_DMA <= g.cols && y0 - R >= -(long)g.halo_top &&
                          y_end + R <= g.rows + g.halo_bot && y_end - y0 == a.tile_rows;
    if (interior) {
        MomWalk<R, Shape, OM, false> w(a, lds_rows[wv], x_tile, y0, y_end, lane);
        if (w.run()) return;
    } else {
#ifdef XRS_MOM_T_SKIPEDGE
        return;
#endif
        // (rim tiles without NaN cells on gentle relief stay on the two-column walk: one fixed shift per lane, geometric counts)
        MomWalk<R, Shape, OM, true> w(a, lds_rows[wv], x_tile, y0, y_end, lane);
        if (w.run()) return;
    }
#ifdef XRS_MOM_NO_FALLBACK
    return;
#endif
    // the fast walk met a non-finite sum or failed its guard.  Inside a nodata region: nothing to walk
#ifndef XRS_MOM_T_NOALLNAN
    if (walk_tile_all_nan<(C::TW + 2 * R + 63) / 64>(g, x_tile - R, x_tile + C::TW + R, y0 - R, y_end + R, lane)) {
        float *const planes[3] = {a.out_mean, a.out_var, a.out_std};
        if (C::TW == 128 && x_tile + 128 <= g.cols) {
            for (int i = 0; i < 3; ++i) if (planes[i]) fill_tile128_nt(planes[i], g.ld_out, x_tile, y0, y_end, lane, nan_f32());
            if (a.out_sum) fill_tile128_nt(a.out_sum, g.ld_out, x_tile, y0, y_end, lane, 0.0f);
        } else {
            walk_fill_no_data(g, planes, 3, a.out_sum, 0.0f, x_tile, x_tile + C::TW, y0, y_end, lane);
        }
        return;
    }
#endif
    // NaN cells under a window (scattered nodata, a nodata region's rim): the SAME two-column walk again, this time carrying
    // them (MomWalk<.., CARRY>: ~1.2x a plain walk; the plain walk in front of it stopped at the first round that met a
    // NaN, so a clean raster pays nothing).  Interior tiles of solid shapes with a compile-time plane set.
    if constexpr (CARRIES) {
        if (interior) {
            MomWalk<R, Shape, OM, false, true> w(a, lds_rows[wv], x_tile, y0, y_end, lane);
            w.nanmap = nan_row[wv];
            w.lostring = lost_ring[wv];
            if (w.run()) return;
#ifdef XRS_MOM_CARRY_ONLY          // (probe builds: what the carrying walk alone costs, and which tiles it hands on -- their outputs stay unwritten)
            return;
#endif
        }
    }
    // What is left -- the rim of a nodata region, dense nodata, raster-edge tiles on nodata, +-inf -- is slow work for one wave
    // (the one-column NaN-aware walker, in the worst case the exact float64 walker behind it: 0.7 - 2.5 ms), and a launch
    // cannot end before its last lone wave does.  With a work-list the tile is noted and this wave is free; focal_mom_rescue_kernel,
    // launched behind this kernel, takes the noted tiles apart into half tiles x row bands, one per wave, all at once.
    if (a.rescue) {
        unsigned idx = 0;
        if (lane == 0) idx = atomicAdd(a.rescue, 1u);
        idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
        if (idx < a.rescue_cap) {
            if (lane == 0) a.rescue[2 + idx] = (unsigned)((ty * a.groups_x + gx) * 4 + wv);
            return;
        }
    }
    // (no list, or a full one: in place.)  The NaN-aware float32 walker, 64 columns at a time; what fails THAT guard (+-inf,
    // windows with a few valid cells at the edge of a nodata region, ill-conditioned sums) goes to the exact float64 walker.
    // (Tried: the four waves of the workgroup sharing those exact walks behind a barrier -- a nodata boundary leaves one
    // slow tile per tile row, ~0.8 ms of kernel tail.  It shortened that raster's time by 7 % and cost EVERY raster 5 %:
    // the changed control flow pushed scalar registers of the interior loop into VGPR lanes, 48-56 v_readlane per round
    // instead of 8.  A kernel argument read only by the NaN-aware walker did the same.  tools/readlanes.sh counts them.)
    for (int q = 0; q < C::NC; ++q) {
        if (x_tile + 64 * q >= g.cols) break;
#ifndef XRS_MOM_NO_NANWALK
        MomWalkN<R, Shape, OM> w(a, lds_rows[wv], x_tile + 64 * q, y0, y_end, lane);
        if (w.run()) continue;
#endif
#ifndef XRS_MOM_NO_EXACT           // (debug builds: keep what the NaN-aware walker wrote -- tests/mom_boundary_probe.py)
        mom_exact_tile<R, Shape>(a, x_tile, lane, y0, y_end, q, 1);
#endif
    }
}

// The tiles focal_mom_kernel noted, redone with every wave of the chip: an item = (wave tile, 64-column half, band of rows); the
// NaN-aware float32 walker runs the band (2R rows of run-in each: bands only while the list is short enough for their
// latency to matter more than their overhead), windows that fail its guard are recomputed one by one in float64
// (mom_fix_cells) instead of condemning the band, and only a band with more of those than the list holds -- +-inf spread
// over it, flat ground next to relief -- takes the exact float64 column walker.
constexpr int RESCUE_FIX = 1024;
template <int R, typename Shape, int OM>
__global__ void __launch_bounds__(256, 2) focal_mom_rescue_kernel(const MomArgs a) {     // (2 waves per SIMD: the float64 walker behind
                                                                                          //  the band may spill; it is the rare path)
    using C = MomCfg<R, Shape>;
    __shared__ __attribute__((aligned(16))) float stage[4][2 * (64 + 2 * R)];
    __shared__ unsigned short fixes[4][RESCUE_FIX];
    const unsigned count = a.rescue[0] < a.rescue_cap ? a.rescue[0] : a.rescue_cap;
    if (!count) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const WalkGeom &g = a.g;
    // as many bands per half tile as there are waves to take them (every band pays 2R rows of run-in: latency against work),
    // of 16 output rows at least
    const long waves = (long)gridDim.x * 4;
    long want_nb = waves / ((long)count * C::NC);
    const int max_nb = (a.tile_rows + 15) / 16;
    const int nb = want_nb < 1 ? 1 : want_nb > max_nb ? max_nb : (int)want_nb;
    const int band_rows = (a.tile_rows + nb - 1) / nb;
    const long items = (long)count * C::NC * nb;
    for (long it = (long)blockIdx.x * 4 + wv; it < items; it += (long)gridDim.x * 4) {
        const unsigned ent = a.rescue[2 + it / (C::NC * nb)];
        const int sub = (int)(it % (C::NC * nb)), q = sub / nb, band = sub % nb;
        const long grp = ent >> 2;
        const long ty = grp / a.groups_x, gx = grp % a.groups_x;
        const long x_tile = (gx * 4 + (long)(ent & 3u)) * C::TW;
        const long xw = x_tile + 64 * q;
        if (xw >= g.cols) continue;
        const long yt0 = ty * a.tile_rows;
        const long yt1 = yt0 + a.tile_rows < g.rows ? yt0 + a.tile_rows : g.rows;
        const long y0 = yt0 + (long)band * band_rows;
        const long y_end = y0 + band_rows < yt1 ? y0 + band_rows : yt1;
        if (y0 >= y_end) continue;
#ifdef XRS_RESCUE_EXACT_ONLY        // (probe builds: which stage of the rescue is responsible for a wrong cell)
        mom_exact_tile<R, Shape>(a, x_tile, lane, y0, y_end, q, 1);
        continue;
#endif
        MomWalkN<R, Shape, OM> w(a, stage[wv], xw, y0, y_end, lane);
#ifndef XRS_RESCUE_NO_FIX
        w.fix_list = fixes[wv];
        w.fix_cap = RESCUE_FIX;
#endif
        if (w.run()) {
            if (w.n_fix) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                mom_fix_cells<R, Shape>(a, fixes[wv], w.n_fix, xw, y0, lane);
            }
        } else {
            // the float64 column walker: a launch of its own behind this one (inlined here its registers are this kernel's --
            // 332 against 256, 129 of them spilled in the NaN-aware walker's loop, the common path)
            unsigned idx = 0;
            if (lane == 0) idx = atomicAdd(a.exact, 1u);
            idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
            // (the list holds every band this launch can form: items <= max(waves, 2 tiles) <= exact_cap, xrs_common.h)
            if (lane == 0 && idx < a.exact_cap) {
                uint4 e4;
                e4.x = (unsigned)x_tile; e4.y = (unsigned)q; e4.z = (unsigned)y0; e4.w = (unsigned)y_end;
                reinterpret_cast<uint4 *>(a.exact + 4)[idx] = e4;
            }
        }
    }
}

template <int R, typename Shape>
__global__ void __launch_bounds__(256) focal_mom_exact_kernel(const MomArgs a) {
    const unsigned count = a.exact[0] < a.exact_cap ? a.exact[0] : a.exact_cap;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (unsigned it = blockIdx.x * 4 + wv; it < count; it += gridDim.x * 4) {
        const uint4 e4 = reinterpret_cast<const uint4 *>(a.exact + 4)[it];
        mom_exact_tile<R, Shape>(a, (long)e4.x, lane, (long)e4.z, (long)e4.w, (int)e4.y, 1);
    }
}

template <int R, typename Shape>
int launch_mom(MomArgs &a, const double *kernel, hipStream_t s) {
    using C = MomCfg<R, Shape>;
    if (!is_shape<R, Shape>(kernel)) return -1;
    WalkGeom &g = a.g;
    g.tiles_x = (g.cols + C::TW - 1) / C::TW;
    static thread_local int wg_per_cu = 0;                     // (per instantiation: registers depend on the radius)
    if (!wg_per_cu) wg_per_cu = walk3_wg_per_cu(focal_mom_kernel<R, Shape, shape_has_hole<Shape>(R) ? 0 : (MOM_SUM | MOM_MEAN | MOM_VAR | MOM_STD)>, XRS_MOM_WAVES);
    a.tile_rows = C::nin(walk3_tile_base(g.rows, (g.tiles_x + 3) / 4, R, C::U, wg_per_cu)) - 2 * R;
    const long tiles_y = (g.rows + a.tile_rows - 1) / a.tile_rows;
    g.n_tiles = g.tiles_x * tiles_y;
    a.groups_x = (g.tiles_x + 3) / 4;
    a.n_groups = a.groups_x * tiles_y;
    a.rim_first = RimFirst::mode_from_env();
    const long grid = RimFirst(a.groups_x, tiles_y, a.rim_first).grid();
    if (grid > 0x7fffffffL) return fail("focal moments: raster too large for one launch");
    if (std::is_same<Shape, BoxShape>::value && a.todo) {
        // np.ones((k, k)): the separable walk of boxsep.hip first; this kernel then redoes the tiles it marked (NaN / inf cells,
        // flat windows away from its shift) and nothing else
        const int rc = launch_box_sep(g.in, a.out_sum, a.out_mean, a.out_var, a.out_std, g.rows, g.cols, g.ld_in, g.ld_out,
                                      C::K, C::K, g.halo_top, g.halo_bot, const_cast<unsigned char *>(a.todo), a.groups_x, a.tile_rows,
                                      4 * C::TW, s);
        if (rc > 0) return rc;
        if (rc < 0) a.todo = nullptr;
    }
    const int om = (a.out_sum ? MOM_SUM : 0) | (a.out_mean ? MOM_MEAN : 0) | (a.out_var ? MOM_VAR : 0) | (a.out_std ? MOM_STD : 0);
    constexpr int ALL = MOM_SUM | MOM_MEAN | MOM_VAR | MOM_STD, MVS = MOM_MEAN | MOM_VAR | MOM_STD;
    a.rescue = mom_rescue_slot();
    if (a.rescue) {
        a.rescue_cap = (unsigned)(g.tiles_x * tiles_y);
        if (mom_rescue_bytes(g.rows, g.cols) < 8 + 4 * (size_t)a.rescue_cap) a.rescue = nullptr;
        else {
            XRS_HIP(hipMemsetAsync(a.rescue, 0, 8, s));
            a.exact = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(a.rescue) + mom_exact_offset(g.rows, g.cols));
            a.exact_cap = (unsigned)mom_exact_cap(g.rows, g.cols);
            XRS_HIP(hipMemsetAsync(a.exact, 0, 16, s));
        }
    }
    // (annuli: one instantiation per (outer, inner) radius pair, the run-time plane set -- 66 pairs up to radius 12)
    if constexpr (shape_has_hole<Shape>(R)) {
        (void)om; (void)ALL; (void)MVS;
        hipLaunchKernelGGL((focal_mom_kernel<R, Shape, 0>), dim3((unsigned)grid), dim3(256), 0, s, a);
    } else {
        if (om == ALL) hipLaunchKernelGGL((focal_mom_kernel<R, Shape, ALL>), dim3((unsigned)grid), dim3(256), 0, s, a);
        else if (om == MVS) hipLaunchKernelGGL((focal_mom_kernel<R, Shape, MVS>), dim3((unsigned)grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((focal_mom_kernel<R, Shape, 0>), dim3((unsigned)grid), dim3(256), 0, s, a);
    }
    XRS_LAUNCH_CHECK();
    if (a.rescue) {
        // (the plane set at run time: one instantiation per shape; an empty list costs the launch, ~5 us)
        int dev = 0, n_cu = 256;
        hipDeviceProp_t prop;
        static thread_local int cus = 0;
        if (!cus) cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : n_cu;
        hipLaunchKernelGGL((focal_mom_rescue_kernel<R, Shape, 0>), dim3((unsigned)(cus * 2)), dim3(256), 0, s, a);
        XRS_LAUNCH_CHECK();
        hipLaunchKernelGGL((focal_mom_exact_kernel<R, Shape>), dim3((unsigned)(cus * 2)), dim3(256), 0, s, a);
        XRS_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

namespace xrs {

#ifndef XRS_MOM_ANNULUS_R
// 0 = launched, -1 = not this shape with a radius of 4..12 cells (caller takes another kernel), > 0 = error.
int XRS_MOM_ENTRY(const float *in, float *out_sum, float *out_mean, float *out_var, float *out_std, long rows, long cols,
                  long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                  hipStream_t s, unsigned char *todo_dev) {
    if (krows != kcols || !(krows & 1)) return -1;
    if (!out_sum && !out_mean && !out_var && !out_std) return 0;
    MomArgs a;
    memset(&a, 0, sizeof(a));
    a.g.in = in; a.g.rows = rows; a.g.cols = cols; a.g.ld_in = ld_in; a.g.ld_out = ld_out;
    a.g.halo_top = halo_top; a.g.halo_bot = halo_bot;
    a.out_sum = out_sum; a.out_mean = out_mean; a.out_var = out_var; a.out_std = out_std;
    a.todo = todo_dev;
    switch (krows / 2) {
#define XRS_MOM_CASE(RR) case RR: return launch_mom<RR, XRS_MOM_SHAPE>(a, kernel, s);
#ifndef XRS_MOM_PROBE
        XRS_MOM_CASE(4) XRS_MOM_CASE(5) XRS_MOM_CASE(6) XRS_MOM_CASE(7) XRS_MOM_CASE(8) XRS_MOM_CASE(9) XRS_MOM_CASE(10) XRS_MOM_CASE(11)
#endif
        XRS_MOM_CASE(12)
#undef XRS_MOM_CASE
        default: return -1;
    }
}
#else
// annulus_kernel(1, 1, XRS_MOM_ANNULUS_R, RI), 1 <= RI < R: one instantiation per inner radius (one translation unit per
// outer radius: the moments kernel is the slow one to compile).  0 = launched, -1 = not such an annulus, > 0 = error.
template <int RI>
int mom_annulus_pair(MomArgs &a, const double *kernel, int ri, hipStream_t s) {
    if constexpr (RI >= XRS_MOM_ANNULUS_R) return -1;
    else {
        if (ri == RI) return launch_mom<XRS_MOM_ANNULUS_R, AnnulusShape<RI>>(a, kernel, s);
        return mom_annulus_pair<RI + 1>(a, kernel, ri, s);
    }
}
int XRS_MOM_ENTRY(const float *in, float *out_sum, float *out_mean, float *out_var, float *out_std, long rows, long cols,
                  long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                  hipStream_t s) {
    if (krows != kcols || krows / 2 != XRS_MOM_ANNULUS_R || !(krows & 1)) return -1;
    const int ri = annulus_inner_radius(kernel, krows);
    if (ri < 1) return -1;
    if (!out_sum && !out_mean && !out_var && !out_std) return 0;
    MomArgs a;
    memset(&a, 0, sizeof(a));
    a.g.in = in; a.g.rows = rows; a.g.cols = cols; a.g.ld_in = ld_in; a.g.ld_out = ld_out;
    a.g.halo_top = halo_top; a.g.halo_bot = halo_bot;
    a.out_sum = out_sum; a.out_mean = out_mean; a.out_var = out_var; a.out_std = out_std;
    return mom_annulus_pair<1>(a, kernel, ri, s);
}
#endif

}  // namespace xrs
