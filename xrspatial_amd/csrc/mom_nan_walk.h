// The NaN-aware float32 moments walker (MomWalkN) with the argument block and the per-radius constants it shares with
// mom_impl.h's two-column walker.  Its own header because two kernels end in it: the moments kernel (mom_impl.h: mean / var /
// std / sum) and the wide mean / sum kernel (wide_impl.h), for the tiles whose fast walk met a NaN.
#pragma once
#include "circle_walk.h"
#include "lds_dma.h"

#include "wave_reduce.h"

#include <utility>

using namespace xrs;

namespace {

struct MomArgs {
    WalkGeom g;                   // in, rows, cols, ld_in, ld_out, halo_top, halo_bot (tiles_x / n_tiles: wave tiles)
    float *out_sum, *out_mean, *out_var, *out_std;
    long n_groups, groups_x;      // workgroups = groups of 4 horizontally adjacent wave tiles
    int tile_rows;                // output rows per tile (tile_rows + 2R input rows = a whole number of rounds)
    int rim_first;                // work order (circle_walk.h RimFirst)
    const unsigned char *todo;    // boxes behind boxsep.hip's fast walk: one byte per workgroup tile, 0 = nothing to do; else NULL
    unsigned *rescue;             // work-list of the wave tiles the fast walks handed on: [0] count, [2..] tiles; or NULL
    unsigned rescue_cap;          // entries it holds
    unsigned *exact;              // bands the rescue launch hands on to focal_mom_exact_kernel: [0] count, then from [4] on
    unsigned exact_cap;           // 16-byte entries {x_tile, q, y0, y_end}
};


constexpr float const_sqrt(float x) {                      // (compile-time constants only)
    float r = x > 1.0f ? x : 1.0f;
    for (int i = 0; i < 40; ++i) r = 0.5f * (r + x / r);
    return r;
}

template <int R, typename Shape>
struct MomCfg {
    static constexpr int K = 2 * R + 1;
#ifndef XRS_MOM_NC
#define XRS_MOM_NC 2
#endif
    static constexpr int NC = XRS_MOM_NC;                  // columns per lane
    static constexpr int TW = 64 * NC;                     // columns per wave tile
    static constexpr int HL = NC * ((R + NC - 1) / NC);    // halo columns each side, whole lane groups
    static constexpr int NV = NC + 2 * HL;                 // cells a lane reads back per row
    static constexpr int NQ = NV / NC;
    static constexpr int CELLS = TW + 2 * HL;
    static constexpr int CELLS_DMA = (CELLS + 3) / 4 * 4;  // ... as the LDS-DMA moves them: 16 bytes per lane (interior tiles have them all)
    static_assert(CELLS_DMA <= 256, "one 16-byte DMA per row");
    static constexpr int NTAPS = shape_taps<Shape>(R);
#ifndef XRS_MOM_U
#define XRS_MOM_U 5
#endif
    static constexpr int U = XRS_MOM_U;                    // rows per unrolled round = rows between two re-centrings
#ifndef XRS_MOM_D
#define XRS_MOM_D 8
#endif
    static constexpr int D = XRS_MOM_D;                    // rows in flight by LDS-DMA; D + 1 row buffers per wave
    static constexpr int RBF = 256;                        // floats per row buffer (the 16-byte DMA writes a whole KiB)
    // input rows a full tile walks: whole rounds covering `base` output rows + the 2R rows of run-in
    static constexpr int nin(int base) { return ((base + 2 * R + U - 1) / U) * U; }
    // How much of a re-centring by d is still inside the partial sums of the rows about to be emitted: a row emitted k
    // rounds later was at most K - (k - 1) U - 1 input rows old when it happened, i.e. held that fraction of its cells
    // (radius 12, U = 5: 1, 0.84, 0.6, 0.33, 0.09, 0).  Kept as two numbers: d^2 of the last re-centring and a decaying
    // maximum of the older ones (x 0.85, then x 0.7 per round: 0.85, 0.6, 0.42, 0.29 ... -- never below the table).
    static constexpr float HIST_FIRST = 0.85f, HIST_DECAY = 0.7f;
    static constexpr bool level_used(int h) {
        for (int dy = 0; dy <= R; ++dy)
            if (Shape::hw(R, dy) == h) return true;
        return false;
    }
    // cells a ring slot has accumulated at a round boundary: the slot that the next row will hit at offset dy_next
    // has seen the rows at offsets -R .. dy_next - 1
    static constexpr int seen(int idx) {
        const int dy_next = idx <= R ? -idx : K - idx;
        int n = 0;
        for (int dy = -R; dy < dy_next; ++dy) n += shape_row_cells<Shape>(R, dy < 0 ? -dy : dy);
        return n;
    }
};

// OM: the outputs the launch writes (bit 0 sum, 1 mean, 2 var, 3 std) as a compile-time set, or 0 = whatever pointers are
// non-null at run time.  The common sets are compiled in because the run-time form keeps its "is this plane wanted" flags
// in vector registers this kernel does not have (one spilled flag = one scratch reload + s_waitcnt vmcnt(0) per round,
// which drains the DMA ring); it also sets the vmcnt bookkeeping of the ring (assuming fewer stores than the truth is safe).
enum : int { MOM_SUM = 1, MOM_MEAN = 2, MOM_VAR = 4, MOM_STD = 8 };

// ---- The NaN-aware walker: raster edges, nodata regions, scattered NaN cells -- still float32, still about a shift that
// trails the walk.  One column per lane (64-column half tiles); every staged cell travels as z = (valid ? v : 0) and
// f = (valid ? 1 : 0), valid = inside the raster and not NaN (the loading lane decides, once per cell); a reader forms
// w = z - c f, and THREE lane-local prefix sums -- of f, w and w^2 -- give every centred run's count, sum and sum of squares
// with one subtraction each; three register rings (N, S, Q) carry the 2R+1 output rows in flight.  With the counts in the
// ring the re-centring needs no compile-time cell counts (S' = S - N d with the slot's own N), so it works at the raster
// edge and next to nodata exactly as in the interior: mean = c + S / N, var = (Q - S^2 / N) / N, sum = N c + S, NaN (sum: 0)
// for a window without a valid cell (numba nanmean / nanvar / nansum of an empty window).  Same guard as the fast walk;
// +-inf or a failed guard hand the half tile to the exact float64 walker.
// Round 2 and the first form of this file sent every tile with a NaN cell to that exact walker: a raster with 0.1 %
// scattered NaN took 5.0 ms instead of 1.3 for mean + var + std, one with a nodata third 33 ms (tools/nan_probe.py).
template <int R, typename Shape, int OM>
struct MomWalkN {
    __device__ __forceinline__ bool want(int bit, const float *p) const { return OM ? (OM & bit) != 0 : p != nullptr; }
    using C = MomCfg<R, Shape>;
    static constexpr int K = C::K, U = C::U;
    static constexpr int STG = 64 + 2 * R;                 // staged cells per row: raster columns xw - R .. xw + 63 + R

    // HAVE_Q = false (only mean and / or sum wanted: the wide kernel's NaN tiles): no squares, no variance guard -- the
    // rounding of S is bounded at the end of the tile instead, like the wide kernel bounds its own: (a few 10^4 u) A against
    // the smallest |mean| written, A = the largest |v - c| any lane staged + the spread of the lanes' shifts
    static constexpr bool HAVE_Q = OM == 0 || (OM & (MOM_VAR | MOM_STD)) != 0;
    float accN[K], accS[K], accQ[HAVE_Q ? K : 1];
    float a_max, m_min, c_lo, c_hi;                        // (HAVE_Q = false) largest staged |w|, smallest |mean|, range of c
    float pf_own[U], pf_halo[U];                           // the rows of the current round, loaded up front
    float c, snapS, snapN;                                 // the shift; sum / count of the round's widest runs about it
    float dq_last, dq_old, dqm;                            // d^2 of the last re-centring, decaying maximum of the older ones, max
    unsigned long long badm;
    bool seeded;                                           // (wave-uniform) the first shift has been chosen
    float gmf;
    int t, n_in;

    const MomArgs &a;
    const WalkGeom &g;
    float *lds;                                            // Z[STG] then F[STG]
    unsigned short *fix_list = nullptr;                    // LDS: outputs that failed their guard, (row in the band) << 6 | lane; the
    int fix_cap = 0, n_fix = 0;                            // caller recomputes them one by one (mom_fix_cells); NULL: a failure hands on the half tile
    static constexpr int NH = (K + U - 1) / U + 1;         // rounds whose rows can still lie under a window being completed
    float a_hist[NH];                                      // (fix list only, wave-uniform) largest |v - c| staged in each of them
    float a_span = 0.0f;                                   // ... and their maximum
    unsigned lds_z;                                        // LDS byte address of Z[lane]
    long xw, x, y0, y_end, y_first;
    int lane;

    __device__ __forceinline__ MomWalkN(const MomArgs &a_, float *lds_, long xw_, long y0_, long ye, int lane_)
        : a(a_), g(a_.g), lds(lds_), xw(xw_), x(xw_ + lane_), y0(y0_), y_end(ye), lane(lane_) {}

    __device__ __forceinline__ void load_row(int il, float &own, float &halo) const {
        const long yy = y_first + il;
        own = halo = nan_f32();                                 // outside the raster == not valid
        const bool row_ok = il < n_in && yy >= -(long)g.halo_top && yy < g.rows + g.halo_bot;     // wave-uniform
        if (!row_ok) return;
        const float *p = g.in + yy * g.ld_in;
        const long xa = xw - R + lane, xb = xa + 64;
        if (xa >= 0 && xa < g.cols) own = p[xa];
        if (lane < 2 * R && xb >= 0 && xb < g.cols) halo = p[xb];
    }

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j) { accN[j] = 0.0f; accS[j] = 0.0f; accQ[HAVE_Q ? j : 0] = 0.0f; }
        a_max = 0.0f; m_min = INFINITY;
        snapS = snapN = 0.0f;
        dq_last = dq_old = dqm = 0.0f;
        badm = 0;
        gmf = (want(MOM_MEAN, a.out_mean) || want(MOM_SUM, a.out_sum)) ? 0.04f : 0.0f;
        t = 0;
        y_first = y0 - R;
        n_in = (int)(y_end - y0) + 2 * R;
        lds_z = lds_addr(lds) + 4u * (unsigned)lane;
        c = 0.0f;
        seeded = false;
#pragma unroll
        for (int q = 0; q < NH; ++q) a_hist[q] = 0.0f;
        a_span = 0.0f;
    }

    // first shift, from the rows of the FIRST ROUND THAT HOLDS A VALID CELL: the mean of the lane's valid cells (staged cell
    // `lane` = raster column x - R: close enough for a first value), else any lane's -- the next re-centring replaces it.
    // (Until round 5 it came from the tile's first round whatever that held: a tile whose first rows lie outside the raster
    // -- every top-edge tile -- or inside a nodata region started about 0, met values around 1000 some rounds later and
    // re-centred by 1000 with those cells already in its sums; the guard then failed the tile's first output rows and the
    // whole half tile went to the exact float64 walker, ~3 ms of a lone wave.  All 128 top-edge tiles of a 16384^2 raster with
    // scattered nodata: the launch could not end before they did, 25x25 mean + var + std 2.4 instead of 1.8 ms, with the sum
    // plane 3.3; the tile row along the lower rim of a nodata region likewise.  Before the first valid cell every sum is
    // zero, so the shift is free to be anything: it is simply not chosen yet.)
    __device__ __forceinline__ void first_shift() {
        // (the HALO cells count too: a half tile whose own 64 staged columns are all nodata while its halo columns hold data
        // -- the last columns left of a region's rim -- was never "seeded" by its own cells, and this function then put c back
        // to 0 at the head of EVERY round, under sums accumulated about the shift the re-centring had moved to: garbage that
        // no guard sees.  Masked until round 6 because such a tile always held a window that failed and went to the exact
        // walker as a whole; tests/probes/rescue_debug.py found it the day single windows began to be repaired.)
        float s = 0.0f, n = 0.0f, sh = 0.0f, nh = 0.0f;
#pragma unroll
        for (int r = 0; r < U; ++r) {
            const bool ok = isfinite(pf_own[r]);
            s += ok ? pf_own[r] : 0.0f;
            n += ok ? 1.0f : 0.0f;
            const bool okh = lane < 2 * R && isfinite(pf_halo[r]);
            sh += okh ? pf_halo[r] : 0.0f;
            nh += okh ? 1.0f : 0.0f;
        }
        const float m = n > 0.0f ? s / n : nh > 0.0f ? sh / nh : 0.0f;
        const unsigned long long have = __ballot(n > 0.0f || nh > 0.0f);
        const float m_any = __shfl(m, have ? __ffsll((long long)have) - 1 : 0);      // (every lane executes the shuffle)
        c = n > 0.0f ? m : have ? m_any : 0.0f;
        seeded = have != 0;
        c_lo = c_hi = c;
    }

    // WHAT: 0 = counts, 1 = w, 2 = w^2
    template <int PHASE, int WHAT>
    __device__ __forceinline__ void pass() {
        float p[K];
        lds_cfloat *z = (lds_cfloat *)(size_t)lds_z;
        lds_cfloat *f = (lds_cfloat *)(size_t)(lds_z + 4u * STG);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (WHAT == 0) { p[k] = f[k]; continue; }
            const float w = fmaf(-c, f[k], z[k]);
            p[k] = WHAT == 1 ? w : w * w;
        }
        // running sums from the centre outwards (mom_impl.h, moment_pass: a run = the sum of ITS cells, nothing beside it)
#pragma unroll
        for (int j = R - 1; j >= 0; --j) p[j] += p[j + 1];
#pragma unroll
        for (int j = R + 2; j < K; ++j) p[j] += p[j - 1];
        auto run_sum = [&](int h) -> float { return h == 0 ? p[R] : p[R - h] + p[R + h]; };
        if constexpr (!shape_has_hole<Shape>(R)) {
#pragma unroll
            for (int h = 0; h <= R; ++h) {
                if (!C::level_used(h)) continue;
                const float S = run_sum(h);
                if (h == R) {                                  // (hw(0) == R for every shape)
                    if (WHAT == 0) snapN += S;
                    if (WHAT == 1) snapS += S;
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int dy = j - R;
                    if (Shape::hw(R, dy < 0 ? -dy : dy) != h) continue;
                    const int idx = ((PHASE - dy) % K + K) % K;
                    if (WHAT == 0) accN[idx] += S;
                    else if (WHAT == 1) accS[idx] += S;
                    else accQ[HAVE_Q ? idx : 0] += S;
                }
            }
        } else {
            // a shape with a hole (annuli): every distinct row pattern once, run(hw) - run(hwi) (compile-time tables)
            constexpr ShapeRows<R, Shape> T{};
            auto run = [&](int h) -> float { return run_sum(h); };      // the centred run of half-width h (static h)
            {                                                  // the widest run: the shift's next estimate
                const float W = run(R);
                if (WHAT == 0) snapN += W;
                if (WHAT == 1) snapS += W;
            }
#pragma unroll
            for (int d = 0; d <= R; ++d) {
                if (T.pat[d] != d) continue;
                float S = run(T.hw[d]);
                if (T.hwi[d] >= 0) S -= run(T.hwi[d] >= 0 ? T.hwi[d] : 0);
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int dy = j - R;
                    if (T.pat[dy < 0 ? -dy : dy] != d) continue;
                    const int idx = ((PHASE - dy) % K + K) % K;
                    if (WHAT == 0) accN[idx] += S;
                    else if (WHAT == 1) accS[idx] += S;
                    else accQ[HAVE_Q ? idx : 0] += S;
                }
            }
        }
    }

    template <int PHASE>
    __device__ __forceinline__ void step() {
        const int i = t + PHASE;
        if (i >= n_in) return;
        {   // ---- stage the row: the loading lane decides validity once per cell
            const float o = pf_own[PHASE], hq = pf_halo[PHASE];
            const bool vo = o == o, vh = hq == hq;
            if (!HAVE_Q) {
                a_max = fmaxf(a_max, vo ? fabsf(o - c) : 0.0f);
                if (lane < 2 * R) a_max = fmaxf(a_max, vh ? fabsf(hq - c) : 0.0f);
            }
            lds[lane] = vo ? o : 0.0f;
            lds[STG + lane] = vo ? 1.0f : 0.0f;
            if (lane < 2 * R) {
                lds[64 + lane] = vh ? hq : 0.0f;
                lds[STG + 64 + lane] = vh ? 1.0f : 0.0f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                         // (LDS serves one wave's instructions in order)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        pass<PHASE, 0>();
        pass<PHASE, 1>();
        if (HAVE_Q) pass<PHASE, 2>();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- the output row R rows up is complete
        constexpr int DONE = ((PHASE - R) % K + K) % K;
        const long yo = y0 + (i - 2 * R);
        bool bad = false;
        if (i >= 2 * R && yo < y_end && x < g.cols) {
            const float n = accN[DONE], S = accS[DONE], Q = accQ[HAVE_Q ? DONE : 0];
            float mean = nan_f32(), var = nan_f32(), sd = nan_f32(), sum = 0.0f;
            if (n > 0.0f) {
                const float ms = S / n;
                mean = c + ms;
                sum = fmaf(n, c, S);
                if (HAVE_Q) {
                    // (one valid cell: variance exactly 0, whatever the shift -- unless that cell is +-inf: the reference's nanvar
                    //  takes (inf - inf)^2 = NaN there; until round 6 a neighbouring window with n > 1 failed its guard and sent the
                    //  whole tile to the exact walker, which masked it; the differential fuzzer found it once windows stood alone)
                    const float e = n == 1.0f ? mean - mean : Q - S * ms;
                    const float B = Q + n * dqm;
                    bad = (n != 1.0f && !(e >= 0.2f * B)) || !(mean * mean * n >= gmf * B);
                    var = e / n;
                    sd = sqrtf(var);
                    if (shape_has_hole<Shape>(R)) {
                        // Annuli: a row that crosses the hole is the difference of two centred runs, and the hole's cells are under
                        // both: a cell far from the shift there -- a spike at the very centre of the ring -- is in no tap of the window,
                        // adds nothing to its Q, and leaves ~u A of rounding per term whatever the window itself holds.  ~sqrt(1.5 n)
                        // such terms behave like a random walk: accepted while that stays under 2e-6 of n |mean| (sum, mean) and of
                        // n var (squares: A^2); A = the largest |v - c| staged in the rounds under the window (wave-wide: coarse).
                        // (Solid shapes needed this too while a run was the difference of two PREFIX sums, which also hold the cells
                        // left of the run; since pass() sums from the centre outwards a run holds its own cells only.)
                        const float lim = 28.0f * sqrtf(n);        // 2e-6 / (1.2 u), u = 2^-24
                        bad = bad || !(a_span <= lim * fabsf(mean)) || (n != 1.0f && !(a_span * a_span <= lim * var));
                    }
                } else {
                    m_min = fminf(m_min, fabsf(mean));
                    bad = !(fabsf(mean) <= 3.0e38f);                    // +-inf under the window: the exact walker's business
                }
            }
            const long off = yo * g.ld_out + x;
            if (want(MOM_MEAN, a.out_mean)) a.out_mean[off] = mean;
            if (want(MOM_VAR, a.out_var)) a.out_var[off] = var;
            if (want(MOM_STD, a.out_std)) a.out_std[off] = sd;
            if (want(MOM_SUM, a.out_sum)) a.out_sum[off] = sum;
        }
        // (outside the lane-divergent block: the verdict must be the same in EVERY lane, columns beyond the raster
        // included -- the exact walker's wave-wide reductions need the whole wave to arrive together)
        const unsigned long long bm = __builtin_amdgcn_ballot_w64(bad);
        if (bm) {                                              // (rare)
            const int nb = __popcll(bm);
            if (fix_list && n_fix + nb <= fix_cap) {
                // a window that failed its guard (a few valid cells at the rim of a nodata region whose sample variance is small
                // by chance, +-inf under it): noted, the walk goes on -- one window in a thousand is no reason to redo 64 x 200
                if (bad) fix_list[n_fix + __popcll(bm & ((1ull << lane) - 1ull))] = (unsigned short)(((i - 2 * R) << 6) | lane);
                n_fix += nb;
            } else {
                badm |= bm;
            }
        }
        accN[DONE] = 0.0f; accS[DONE] = 0.0f; accQ[HAVE_Q ? DONE : 0] = 0.0f;
    }

    __device__ __forceinline__ void recentre() {
        // the lane's own estimate of the level of its columns: the mean of the round's widest runs.  Next to nodata a run
        // holds few valid cells and its mean jitters (a jittering shift trips the guard), inside nodata it holds none: such
        // lanes follow the wave's estimate instead (mean of the lanes that have one)
        const bool own = snapN >= 16.0f;
        float est = c + (snapN > 0.0f ? snapS / snapN : 0.0f);
        {
            float se = own ? est : 0.0f, sn = own ? 1.0f : 0.0f;
            se = wave_reduce<WrSum>(se);                       // (wave_reduce.h: DPP, wave-uniform results)
            sn = wave_reduce<WrSum>(sn);
            if (!own) est = sn > 0.0f ? se / sn : est;
        }
        float d = est - c;
        snapS = snapN = 0.0f;
        const float c_new = c + d;
        d = c_new - c;                                         // exact: the step the walk really takes
        c = c_new;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float S = accS[j];
            const float S2 = fmaf(-accN[j], d, S);
            if (HAVE_Q) accQ[j] -= d * (S + S2);
            accS[j] = S2;
        }
        c_lo = fminf(c_lo, c); c_hi = fmaxf(c_hi, c);
        dq_old = fmaxf(dq_last * C::HIST_FIRST, dq_old * C::HIST_DECAY);
        dq_last = d * d;
        dqm = fmaxf(dq_last, dq_old);
    }

    template <int... P>
    __device__ __forceinline__ void round(std::integer_sequence<int, P...>) {
        (load_row(t + P, pf_own[P], pf_halo[P]), ...);
        if (!seeded) first_shift();
        if (HAVE_Q && shape_has_hole<Shape>(R)) {              // (the rows of this round, before any of them is summed)
            float m = 0.0f;
#pragma unroll
            for (int r = 0; r < U; ++r) {
                const float o = fabsf(pf_own[r] - c), h = fabsf(pf_halo[r] - c);
                m = fmaxf(m, o == o ? o : 0.0f);               // (NaN: nodata, adds nothing; +-inf stays inf and flags what it touches)
                m = fmaxf(m, h == h ? h : 0.0f);
            }
            // a sliding maximum over the rounds a window in flight can reach back to: the terms were about the shift of THEIR
            // round (the re-centring moves sums by exact algebra, not the size of what was rounded)
            a_span = wave_reduce<WrMax>(m);
#pragma unroll
            for (int q = NH - 1; q > 0; --q) a_hist[q] = a_hist[q - 1];
            a_hist[0] = a_span;
#pragma unroll
            for (int q = 1; q < NH; ++q) a_span = fmaxf(a_span, a_hist[q]);
        }
        (step<P>(), ...);
        ring_rotate<K, U>(accN);
        ring_rotate<K, U>(accS);
        if (HAVE_Q) ring_rotate<HAVE_Q ? K : 1, HAVE_Q ? U : 1>(accQ);
        t += U;
        recentre();
    }

    // true: every result of the half tile is good; false: the caller redoes it with the exact float64 walker
    __device__ __forceinline__ bool run() {
        init();
        while (t < n_in) {
            round(std::make_integer_sequence<int, U>{});
            if (badm) return false;
        }
        if (!HAVE_Q) {
            // rounding of a window's S: <= ~K (2 K^2 + K) u A in its rows' prefix sums and their differences + K ntaps u A in
            // the ring, per valid cell no more than that over n -- (a few 10^4 u) A / ntaps for the mean, x 2 for the lanes'
            // different, moving shifts; accepted up to 0.9e-5 of the smallest |mean| (wide_impl.h's bound and contract)
            const float A = wave_reduce<WrMax>(a_max) + (wave_reduce<WrMax>(c_hi) - wave_reduce<WrMin>(c_lo));
            const float M = wave_reduce<WrMin>(m_min);
            constexpr float COEF = 2.0f * 5.9604645e-8f * (float)(K * (2 * K * K + K) + K * C::NTAPS) / (float)C::NTAPS;
            if (!(COEF * A <= 0.9e-5f * M)) return false;
        }
        return true;
    }
};

// The outputs MomWalkN noted (fix_list) computed directly, one window at a time by the whole wave: float64, the reference's
// own arithmetic (numba nanmean: float64 sum / count; nanvar: two passes; nansum: here the exactly rounded sum, like every
// large-window sum of this library).  Lane l takes the window's column x - R + l of each of the 2R+1 rows (the mask decides
// per row whether that cell is a tap), keeps the 2R+1 cells in registers for the second pass.  A few microseconds per window.
template <int R, typename Shape>
__device__ __forceinline__ void mom_fix_cells(const MomArgs &a, const unsigned short *list, int n, long xw, long y0, int lane) {
    constexpr int K = 2 * R + 1;
    const WalkGeom &g = a.g;
    for (int e = 0; e < n; ++e) {
        const unsigned ent = list[e];                          // (wave-uniform)
        const long yo = y0 + (long)(ent >> 6), x = xw + (long)(ent & 63u);
        const int dx = lane - R;                               // this lane's column offset (lanes >= K idle)
        const long xc = x + dx;
        const bool col_ok = lane < K && xc >= 0 && xc < g.cols;
        float v[K];
        double s = 0.0, cnt = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int dy = j - R, ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
            const long yy = yo + dy;
            const bool tap = col_ok && adx <= Shape::hw(R, ady) && adx > Shape::hwi(R, ady) && yy >= -(long)g.halo_top && yy < g.rows + g.halo_bot;
            float t = nan_f32();
            if (tap) t = g.in[yy * g.ld_in + xc];
            v[j] = t;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool ok = v[j] == v[j];
            s += ok ? (double)v[j] : 0.0;
            cnt += ok ? 1.0 : 0.0;
        }
        s = wave_reduce<WrSum>(s);
        cnt = wave_reduce<WrSum>(cnt);
        float mean = nan_f32(), var = nan_f32(), sd = nan_f32();
        if (cnt > 0.0) {
            const double m = s / cnt;
            double q = 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const double d = (double)v[j] - m;
                q += v[j] == v[j] ? d * d : 0.0;
            }
            q = wave_reduce<WrSum>(q);
            const double vr = q / cnt;
            mean = (float)m; var = (float)vr; sd = (float)sqrt(vr);
        }
#ifdef XRS_RESCUE_MARK             // (probe builds: flagged windows show up as -12345 in the mean plane)
        mean = -12345.0f;
#endif
        if (lane == 0) {
            const long off = yo * g.ld_out + x;
            if (a.out_mean) a.out_mean[off] = mean;
            if (a.out_var) a.out_var[off] = var;
            if (a.out_std) a.out_std[off] = sd;
            if (a.out_sum) a.out_sum[off] = (float)s;
        }
    }
}

}  // namespace
