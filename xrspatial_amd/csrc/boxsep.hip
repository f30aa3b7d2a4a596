// Separable statistics over BOX masks -- np.ones((k, k)), the kernels the reference's own benchmark suite runs
// (benchmarks/benchmarks/focal.py:10-34: custom_kernel(np.ones(...)) for apply / focal_stats / hotspots) -- in O(1) work
// per cell whatever k: focal var / std with the mean and the sum beside them (xrspatial/focal.py:226-258 through
// _apply_numpy :305-326).  (The mean or the sum ALONE, and convolve_2d with one weight on the box, stay on the wide row
// walker: this kernel measured no faster there -- 0.53 vs 0.55 ms at 25x25, 0.68 vs 0.47 at 11x11 -- its work per row is
// a fixed ~170 instructions per wave whatever the number of moments.)
//
// A box is the one mask whose window sum factors: sum over the window = sum over its columns of (sum over the column's
// rows).  So instead of 2R+1 ring additions per cell, row and moment (mom_impl.h, wide_impl.h: ~163 VALU instructions per
// cell for the four moments at 25x25), a wave keeps ONE running column sum per column and moment:
//   * a wave owns 64 NC columns x ~256 output rows and walks down; a lane owns NC adjacent columns.  Per output row the
//     row ENTERING the window (y + ry) is added to the column sums and the row LEAVING it (y - ry) subtracted.  The leaving
//     row was read 2 ry + 1 rows earlier: the rows of the window stay in a private LDS ring per wave (2 ry + 1 slots; the
//     entering row is loaded into registers one step ahead and written to the slot of the row that just left), so every
//     input cell is fetched ONCE.  (The first form of this kernel re-loaded the leaving row from global memory -- the chip's
//     L2s do not hold 25 rows x all resident waves, the second read went to the fabric, and the 25x25 mean ran at 0.83 ms
//     against 0.46 for the ring-adding walker it replaces.)  The ring is what limits residency (12.5 KiB per wave at 25x25:
//     8 waves per CU), so a wave does TWO rows per step, their scans and prefix arrays independent of one another: the
//     dependent chain of one row (scan, LDS round trip, float64 arithmetic: ~1000 cycles) hides behind the other's;
//   * the column sums are FLOAT64 sums of d = v - c0 and d^2 (c0 = the cell at the tile centre): d is exact, the sums of d
//     are exact, those of d^2 carry 2^-53 relative per update, so the add / subtract recurrence does not drift in any way
//     float32 results can see, and the shift keeps var = (Q - S^2 / n) / n well conditioned: Q / (n var) = 1 + m^2 / var with
//     m <= the tile's relief, against a guard at 2^24 (1 ulp of float32);
//   * the horizontal box sum of the column sums: lane-local prefix over the NC columns, a wave-wide DPP scan of the lane
//     totals (6 steps), the prefix array through LDS, and every output column is P[x + rx] - P[x - rx - 1] -- independent of
//     rx.  A wave writes the 64 NC - 2 rx columns whose windows lie inside its columns;
//   * nothing here knows the box size at compile time: one kernel for every k (LDS allowing: the ring is sized at launch);
//   * ~1 VALU instruction per cell and moment set instead of ~2.5, few registers (no accumulator ring).
// What the fast walk cannot do -- NaN / inf cells (they would stay in a running sum for ever), windows whose variance drowns
// in the cancellation (flat patches away from the shift: exact zero is the contract there) -- it does not try: the wave
// marks its tile in `todo`, a byte map over the workgroup tiles of the float32 walker that owns the mask shape
// (focal_mom_kernel / focal_wide_kernel with BoxShape), and that kernel, launched right behind this one, redoes exactly
// the marked tiles with its NaN-aware and exact paths.  Raster edges stay here (clipped windows: n = rows x cols inside,
// cells outside contribute d = 0; plain predicated loads of the entering and the leaving row, no ring).
#include "circle_walk.h"
#include "lds_dma.h"
#include "wave_reduce.h"

using namespace xrs;

namespace {

enum : int { BOX_SUM = 1, BOX_MEAN = 2, BOX_VAR = 4, BOX_STD = 8 };

#ifndef XRS_BOX_NC
#define XRS_BOX_NC 2                  // columns per lane: wave tiles of 128 columns (512-byte ring rows)
#endif
#ifndef XRS_BOX_D
#define XRS_BOX_D 4                   // rows in flight by LDS-DMA
#endif

struct BoxArgs {
    const float *in;
    long rows, cols, ld_in, ld_out;
    int halo_top, halo_bot;
    float *out_sum, *out_mean, *out_var, *out_std;
    int rx, ry;                   // half-widths of the box (columns, rows)
    int w_out;                    // output columns per wave tile: 64 NC - 2 rx, rounded down to a multiple of NC
    int tile_rows;                // output rows per wave tile
    int n_slots;                  // ring slots per wave: 2 ry + 2 + rows ahead (even)
    int wave_lds;                 // bytes of LDS per wave: the ring, then 2 rows x 2 moments of prefix arrays
    long tiles_x, tiles_y, groups_x;   // wave tiles; workgroups = 4 horizontally adjacent wave tiles
    int rim_first;
    double n_full, inv_n_full;    // cells of an unclipped window
    // the fall-back kernel's workgroup tiles (todo[ty * fb_groups_x + gx] = 1: redo rows [ty * fb_tile_rows, +fb_tile_rows)
    // x columns [gx * fb_group_cols, +fb_group_cols))
    unsigned char *todo;
    long fb_groups_x;
    int fb_tile_rows, fb_group_cols;
};

template <int NC>
struct BoxGeo {
    static constexpr int TWC = 64 * NC;                    // columns of column sums per wave
    static constexpr int ROWB = TWC * 4;                   // bytes per ring row
    static constexpr int PEVEN = 66;                       // prefix array (NC = 2): [0] = 0, [1 + l] = odd column 2 l + 1, [PEVEN + l] = even column 2 l
    static constexpr int PSLOTS = PEVEN + 64;              // float64 slots per prefix array
    static_assert(NC == 2, "one 16-byte LDS-DMA moves a pair of 512-byte rows");
};
// rows in flight by LDS-DMA (pairs of rows: one DMA instruction each).  The ring is what limits residency: with the
// squares' two extra prefix arrays 4 rows ahead keep two workgroups on a CU at 25x25, without them 8 do
constexpr int box_ahead(bool with_squares) { return with_squares ? XRS_BOX_D : 2 * XRS_BOX_D; }

// RING = true: a full tile whose whole input window lies inside the raster: rows arrive in the wave's LDS ring by LDS-DMA
// and are read from it twice (entering, leaving); RING = false: tiles at the raster / shard edge and the last tile column:
// predicated global loads of the entering and the leaving row, c0 for every cell outside (d = 0), clipped counts.
// NO: planes written per row AT LEAST (the vmcnt bookkeeping of the DMA ring: fewer than the truth is safe)
template <int OM, int NC, bool RING, int NO>
struct BoxWalk {
    static constexpr bool Q = (OM & (BOX_VAR | BOX_STD)) != 0;
    static constexpr bool EDGE = !RING;
    using G = BoxGeo<NC>;
    const BoxArgs &a;
    float *ring;                  // this wave's n_slots rows of TWC cells
    double *p1;                   // this wave's four prefix arrays of PSLOTS float64 each (across_n)
    long xs, x_out0, y0, y_end;
    int lane, rx, ry;
    double c0;
    float c0f;
    double C1[NC], C2[NC];        // (C2 unused -- and optimised away -- without the squares)
    double q_peak[NC];            // the largest window sum of squares each output column has seen on this tile (guard)
    unsigned long long failm;     // lanes with a result the fast walk must not stand for (wave-uniform)

    __device__ __forceinline__ BoxWalk(const BoxArgs &a_, unsigned char *lds, long xs_, long xo, long y0_, long ye, int lane_)
        : a(a_), xs(xs_), x_out0(xo), y0(y0_), y_end(ye), lane(lane_), rx(a_.rx), ry(a_.ry) {
        ring = reinterpret_cast<float *>(lds);
        p1 = reinterpret_cast<double *>(lds + (size_t)a_.n_slots * G::ROWB);
    }

    // EDGE: the lane's NC cells of input row yy (c0 for everything outside the raster / the shard's halo rows: d = 0)
    __device__ __forceinline__ void load_cells(long yy, float (&v)[NC]) const {
#pragma unroll
        for (int j = 0; j < NC; ++j) v[j] = c0f;
        if (yy < -(long)a.halo_top || yy >= a.rows + a.halo_bot) return;          // wave-uniform
        const float *p = a.in + yy * a.ld_in;
        const long x = xs + NC * lane;
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (x + j >= 0 && x + j < a.cols) v[j] = p[x + j];
    }
    typedef float vNC __attribute__((ext_vector_type(NC)));
    // RING: input rows yy, yy + 1 -> ring slots 2 pair, 2 pair + 1 with ONE 16-byte LDS-DMA: lanes 0..31 move the 512 bytes of
    // the first row, lanes 32..63 those of the second (`stride` bytes further; 0 past the tile: the row again, nobody reads it)
    __device__ __forceinline__ void dma_pair(long yy, unsigned stride, int pair, unsigned ring_addr) const {
        const float *p = uniform_ptr(a.in + yy * a.ld_in + xs);
        glds16_s(p, 16u * (unsigned)(lane & 31) + (lane >= 32 ? stride : 0u), ring_addr + (unsigned)pair * (2 * G::ROWB));
    }
    __device__ __forceinline__ void ring_get(int slot, float (&v)[NC]) const {
        const vNC q = *reinterpret_cast<const vNC *>(ring + (size_t)slot * G::TWC + NC * lane);
#pragma unroll
        for (int j = 0; j < NC; ++j) v[j] = q[j];
    }

    // the sums of d = v - c0 (EDGE: cells outside the raster arrive as c0, d = 0) or, in the ring walk, of v itself (float32
    // cells add exactly in float64; one subtraction fewer per cell) and of d^2
    __device__ __forceinline__ double term1(float v) const { return RING ? (double)v : (double)v - c0; }
    __device__ __forceinline__ void enter(const float (&v)[NC]) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            C1[j] += term1(v[j]);
            if (Q) { const double d = (double)v[j] - c0; C2[j] = fma(d, d, C2[j]); }
        }
    }
    __device__ __forceinline__ void leave(const float (&v)[NC]) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            C1[j] -= term1(v[j]);
            if (Q) { const double d = (double)v[j] - c0; C2[j] = fma(-d, d, C2[j]); }
        }
    }

    // Box sums of the column sums for the lane's NC OUTPUT columns x_out0 + NC lane + o (column rx + NC lane + o of the tile),
    // for N independent sets at once (the rows of a step x the moments): set i uses the prefix array pp + i * PSLOTS.
    // Layout of a prefix array (NC = 2): slot 1 + c holds the inclusive prefix up to column c, stored by column parity --
    // [0 .. 64]: slot 0 (= 0) and the odd columns (c = 2 l + 1 at 1 + l), [PEVEN ..]: the even columns (c = 2 l at PEVEN + l) -- so
    // that every access is one float64 per lane at an 8-byte lane stride: conflict free (interleaved, the two float64 of a
    // lane sat 16 bytes apart and every LDS access was a 2-way bank conflict: 46 % of the LDS cycles).
    template <int N>
    __device__ __forceinline__ void across_n(const double (&C)[N][NC], double *pp, double (&B)[N][NC], double (&H)[N][NC]) const {
        static_assert(NC == 2, "prefix layout by column parity");
        double tot[N];
#pragma unroll
        for (int i = 0; i < N; ++i) tot[i] = C[i][0] + C[i][1];
        double inc[N];
#pragma unroll
        for (int i = 0; i < N; ++i) inc[i] = tot[i];
        wave_scan_f64<N>(inc);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double *p = pp + i * G::PSLOTS;
            const double before = inc[i] - tot[i];                      // the columns of the lanes to the left
            p[G::PEVEN + lane] = before + C[i][0];                            // column 2 lane
            p[1 + lane] = inc[i];                                       // column 2 lane + 1
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                                // (LDS serves one wave's instructions in order)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int lc = NC * lane < a.w_out ? lane : 0;                  // (lanes without outputs re-read lane 0's)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double *p = pp + i * G::PSLOTS;
            // output o = 0: columns 2 lc + 2 rx (even) minus 2 lc - 1 (odd; slot 0 for lc = 0);  o = 1: 2 lc + 1 + 2 rx (odd) minus 2 lc (even)
            H[i][0] = p[G::PEVEN + lc + rx];                            // (the prefix a box sum was cut from: finish()'s guard)
            H[i][1] = p[1 + lc + rx];
            B[i][0] = H[i][0] - p[lc];
            B[i][1] = H[i][1] - p[G::PEVEN + lc];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                                // (the next step's writes come after these reads)
    }

    // one output row from the column sums (S, QQ)
    __device__ __forceinline__ void emit(long yo, const double (&S)[NC], const double (&QQ)[NC]) {
        double C[Q ? 2 : 1][NC], B[Q ? 2 : 1][NC], H[Q ? 2 : 1][NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) { C[0][j] = S[j]; if (Q) C[Q ? 1 : 0][j] = QQ[j]; }
        across_n<Q ? 2 : 1>(C, p1, B, H);
        double zero[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) zero[j] = 0.0;
        finish(yo, B[0], Q ? B[Q ? 1 : 0] : zero, Q ? H[Q ? 1 : 0] : zero);
    }
    // two output rows (yo from Sa / Qa, yo + 1 from Sb / Qb): all their scans step by step together
    __device__ __forceinline__ void emit2(long yo, const double (&Sa)[NC], const double (&Qa)[NC], const double (&Sb)[NC],
                                          const double (&Qb)[NC]) {
        constexpr int N = Q ? 4 : 2;
        double C[N][NC], B[N][NC], H[N][NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            C[0][j] = Sa[j]; C[1][j] = Sb[j];
            if (Q) { C[Q ? 2 : 0][j] = Qa[j]; C[Q ? 3 : 1][j] = Qb[j]; }
        }
        across_n<N>(C, p1, B, H);
        double zero[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) zero[j] = 0.0;
        finish(yo, B[0], Q ? B[Q ? 2 : 0] : zero, Q ? H[Q ? 2 : 0] : zero);
        finish(yo + 1, B[1], Q ? B[Q ? 3 : 1] : zero, Q ? H[Q ? 3 : 1] : zero);
    }

    __device__ __forceinline__ void finish(long yo, const double (&B1)[NC], const double (&B2)[NC], const double (&H2)[NC]) {
        const bool out_lane = NC * lane < a.w_out;
        const long xo = x_out0 + NC * lane;
        float r_sum[NC], r_mean[NC], r_var[NC], r_std[NC];
        bool bad = false;
        // EDGE: rows of the window inside the raster (wave-uniform), columns per output
        double ny = 0.0;
        bool rows_full = true;
        if (EDGE) {
            const long lo = yo - ry < -(long)a.halo_top ? -(long)a.halo_top : yo - ry;
            const long hi = yo + ry >= a.rows + a.halo_bot ? a.rows + a.halo_bot - 1 : yo + ry;
            ny = (double)(hi - lo + 1);
            rows_full = hi - lo == 2 * ry;
        }
#pragma unroll
        for (int o = 0; o < NC; ++o) {
            double n = a.n_full, inv = a.inv_n_full;
            bool full = true;
            if (EDGE) {
                const long lo = xo + o - rx < 0 ? 0 : xo + o - rx;
                const long hi = xo + o + rx >= a.cols ? a.cols - 1 : xo + o + rx;
                full = rows_full && hi - lo == 2 * rx;
                if (!full) { n = ny * (double)(hi - lo + 1); inv = 1.0 / n; }
            }
            // B1: the window sum about 0 (ring walk) or about c0 (edge walk); sd: about c0, s0: about 0
            const double sd = RING ? fma(-n, c0, B1[o]) : B1[o];
            const double s0 = RING ? B1[o] : fma(n, c0, B1[o]);
            if (Q) {
                const double e = fma(-sd, sd * inv, B2[o]);              // n * variance
                // The guard knows the column's HISTORY: the add / subtract recurrence of the column sums never re-seeds, so
                // after the window has crossed high relief every column keeps a rounding residual of ~2^-53 of its past peak
                // -- a lake at the tile's shift then has sd = 0 and e = B2 = that residual, which passes any test against
                // the current window alone (round 4: var ~1e-12 and std ~1e-6 where exact 0 is the contract).  An output
                // column's windows always hold the same columns, so the running maximum of its own B2 bounds what the
                // residual can be: below 2^-26 of it (a few hundred residuals: 2^-22 relative on e) the tile is handed on.
                q_peak[o] = fmax(q_peak[o], B2[o]);
                bad |= !(e >= 0x1p-24 * B2[o]) || !(e >= 0x1p-26 * q_peak[o]);     // (a NaN / inf anywhere fails it too)
                // ... and the box sum was the difference of two wave-wide PREFIX sums, which also hold every column to the LEFT of
                // the window: a plateau 1e7 above the shift there (d^2 = 1e14 a cell) is in no window of this lane and still
                // leaves 2^-53 of itself in B2 -- var off by 4e-5 next to such a cliff until round 6 (tests/fuzz_parity.py
                // --windows).  Below 2^-27 of the prefix it was cut from (a few roundings: 2^-25 relative on e) the tile is handed on.
                bad |= !(e >= 0x1p-27 * H2[o]);
                const double var = e * inv;
                r_var[o] = (float)var;
                r_std[o] = __builtin_amdgcn_sqrtf((float)var);               // (v_sqrt_f32: 1 ulp; sqrtf's fix-up costs 8 instructions per cell)
            } else {
                bad |= !(fabs(B1[o]) < INFINITY);
            }
            r_mean[o] = (float)(s0 * inv);
            r_sum[o] = (float)s0;
        }
        const bool live = out_lane && (!EDGE || xo < a.cols);
        failm |= __builtin_amdgcn_ballot_w64(live && bad);
        if (!live) return;
        const long off = yo * a.ld_out + xo;
        if (!EDGE) {
            typedef float stN __attribute__((ext_vector_type(NC), aligned(4)));
            auto put = [&](float *plane, const float (&r)[NC]) {
                stN v;
#pragma unroll
                for (int o = 0; o < NC; ++o) v[o] = r[o];
                __builtin_nontemporal_store(v, reinterpret_cast<stN *>(plane + off));
            };
            if ((OM & BOX_SUM) && a.out_sum) put(a.out_sum, r_sum);
            if ((OM & BOX_MEAN) && a.out_mean) put(a.out_mean, r_mean);
            if ((OM & BOX_VAR) && a.out_var) put(a.out_var, r_var);
            if ((OM & BOX_STD) && a.out_std) put(a.out_std, r_std);
        } else {
#pragma unroll
            for (int o = 0; o < NC; ++o) {
                if (xo + o >= a.cols) break;
                if ((OM & BOX_SUM) && a.out_sum) a.out_sum[off + o] = r_sum[o];
                if ((OM & BOX_MEAN) && a.out_mean) a.out_mean[off + o] = r_mean[o];
                if ((OM & BOX_VAR) && a.out_var) a.out_var[off + o] = r_var[o];
                if ((OM & BOX_STD) && a.out_std) a.out_std[off + o] = r_std[o];
            }
        }
    }

    __device__ __forceinline__ void init() {
        failm = 0;
#pragma unroll
        for (int j = 0; j < NC; ++j) { C1[j] = 0.0; C2[j] = 0.0; q_peak[j] = 0.0; }
        // the shift: the cell at the tile centre (any finite value works; a near one keeps d small)
        const long yc = y0 + (y_end - y0) / 2;
        long xc = x_out0 + a.w_out / 2;
        xc = xc < a.cols ? xc : a.cols - 1;
        const float v = a.in[yc * a.ld_in + xc];
        c0f = isfinite(v) ? v : 0.0f;
        c0 = (double)c0f;
        if (lane == 0) {
            p1[0] = 0.0; p1[G::PSLOTS] = 0.0;
            if (Q) { p1[2 * G::PSLOTS] = 0.0; p1[3 * G::PSLOTS] = 0.0; }
        }
    }

    // true: every result of the tile stands; false: the tile goes to the fall-back kernel
    __device__ __forceinline__ bool run() {
        init();
        if (RING) {
            // Input row i of the tile (raster row y0 - ry + i) lives in ring slot i mod n_slots from its DMA until it has left
            // the window; rows travel in PAIRS (2 p, 2 p + 1) -> slots (2 p, 2 p + 1) mod n_slots (n_slots even: a pair never
            // wraps).  Step s takes pair s: the first ry steps only add its rows (run-in), the others emit output rows
            // y0 + 2 (s - ry) and the next one and take pair s - ry out again.  The DMA of a step (pair s + A + 1, A = pairs
            // ahead; n_slots = 2 ry + 2 + 2 A) overwrites exactly the pair that leaves in it: issued after the step's reads.
            constexpr int A = box_ahead(Q) / 2;
            const int np = a.n_slots / 2;
            const unsigned ring_addr = lds_addr(ring);
            const long y_first = y0 - ry, y_last = y_end - 1 + ry;
            const unsigned row_bytes = (unsigned)(a.ld_in * 4);
            auto dma = [&](int pq, int pair) {
                long yy = y_first + 2 * pq;                              // (past the tile: the last row, twice; nobody reads it)
                const unsigned stride = yy + 1 <= y_last ? row_bytes : 0u;
                yy = yy <= y_last ? yy : y_last;
                dma_pair(yy, stride, pair, ring_addr);
            };
            auto next = [&](int pair) { return pair + 1 == np ? 0 : pair + 1; };
            int pair_dma = 0;
            for (int r = 0; r <= A; ++r) { dma(r, pair_dma); pair_dma = next(pair_dma); }           // pairs 0 .. A
            int pair_in = 0, pair_out = 0;
            int pr = 0;
            // ---- run-in: rows 0 .. 2 ry - 1
            for (; pr < ry; ++pr) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A) : "memory");      // pair pr landed: at most the A pairs behind it in flight
                float v0[NC], v1[NC];
                ring_get(2 * pair_in, v0);
                ring_get(2 * pair_in + 1, v1);
                pair_in = next(pair_in);
                enter(v0);
                enter(v1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma(pr + A + 1, pair_dma); pair_dma = next(pair_dma);
            }
            // ---- the walk (the tile has an even number of rows: every step emits two)
            const int n_steps = (int)(y_end - y0) / 2;
            for (int st = 0; st < n_steps; ++st, ++pr) {
                const long yo = y0 + 2 * st;
                // younger than this step's pair: A DMAs and, once A steps have emitted, the stores of the A steps in between
                if (st >= A) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A + 2 * A * NO) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A) : "memory");
                float e0[NC], e1[NC], l0[NC], l1[NC];
                ring_get(2 * pair_in, e0);
                ring_get(2 * pair_in + 1, e1);
                pair_in = next(pair_in);
                ring_get(2 * pair_out, l0);
                ring_get(2 * pair_out + 1, l1);
                pair_out = next(pair_out);
                // column sums under output row yo (Sa, Qa) and yo + 1 (Sb, Qb), and what the next step starts from
                double Sa[NC], Qa[NC], Sb[NC], Qb[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    Sa[j] = C1[j] + (double)e0[j];
                    Sb[j] = (Sa[j] - (double)l0[j]) + (double)e1[j];
                    C1[j] = Sb[j] - (double)l1[j];
                    if (Q) {
                        const double de0 = (double)e0[j] - c0, de1 = (double)e1[j] - c0;
                        const double dl0 = (double)l0[j] - c0, dl1 = (double)l1[j] - c0;
                        Qa[j] = fma(de0, de0, C2[j]);
                        Qb[j] = fma(de1, de1, fma(-dl0, dl0, Qa[j]));
                        C2[j] = fma(-dl1, dl1, Qb[j]);
                    } else {
                        Qa[j] = Qb[j] = 0.0;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the ring reads have returned: the leaving pair's slots may be refilled)
                dma(pr + A + 1, pair_dma); pair_dma = next(pair_dma);
                emit2(yo, Sa, Qa, Sb, Qb);
                if ((st & 7) == 7 && failm) return false;                // (a NaN / inf stays in the running sums: stop soon, not at once)
            }
            if (failm) return false;
            return true;
        }
        for (long yy = y0 - ry; yy < y0 + ry; ++yy) {                   // run-in: rows y0 - ry .. y0 + ry - 1
            float v[NC];
            load_cells(yy, v);
            enter(v);
        }
        for (long yo = y0; yo < y_end; ++yo) {
            float e[NC], l[NC];
            load_cells(yo + ry, e);
            load_cells(yo - ry, l);
            enter(e);
            emit(yo, C1, C2);
            leave(l);
            if (failm) return false;
        }
        return true;
    }
};

constexpr int box_planes(int om) { return (om & 1) + (om >> 1 & 1) + (om >> 2 & 1) + (om >> 3 & 1); }

template <int OM, int NC, int NO = box_planes(OM)>
__global__ void __launch_bounds__(256) box_sep_kernel(const BoxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char box_lds[];
    using G = BoxGeo<NC>;
    long ty, gx;
    if (!RimFirst(a.groups_x, a.tiles_y, a.rim_first).locate(blockIdx.x, ty, gx)) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tx = gx * 4 + wv;
    if (tx >= a.tiles_x) return;
    const long x_out0 = tx * a.w_out;
    const long xs = x_out0 - a.rx;
    const long y0 = ty * a.tile_rows;
    const long y_end = y0 + a.tile_rows < a.rows ? y0 + a.tile_rows : a.rows;
    const bool interior = xs >= 0 && xs + G::TWC <= a.cols && x_out0 + a.w_out <= a.cols && y0 - a.ry >= -(long)a.halo_top &&
                          y_end + a.ry <= a.rows + a.halo_bot && ((y_end - y0) & 1) == 0;
    unsigned char *lds = box_lds + (size_t)wv * a.wave_lds;
    bool ok;
    if (interior) {
        BoxWalk<OM, NC, true, NO> w(a, lds, xs, x_out0, y0, y_end, lane);
        ok = w.run();
    } else {
        BoxWalk<OM, NC, false, NO> w(a, lds, xs, x_out0, y0, y_end, lane);
        ok = w.run();
    }
    if (ok || lane != 0) return;
    // mark every tile of the fall-back kernel that holds cells of this one
    const long x_hi = (x_out0 + a.w_out < a.cols ? x_out0 + a.w_out : a.cols) - 1;
    for (long fy = y0 / a.fb_tile_rows; fy <= (y_end - 1) / a.fb_tile_rows; ++fy)
        for (long fx = x_out0 / a.fb_group_cols; fx <= x_hi / a.fb_group_cols; ++fx) a.todo[fy * a.fb_groups_x + fx] = 1;
}

typedef void (*BoxKernel)(const BoxArgs);

}  // namespace

namespace xrs {

// Launches the fast walk for an all-ones krows x kcols box (a set of moments with var or std in it).  0 = launched (the
// caller launches its own kernel on the tiles of `todo` behind it), -1 = not for this walk (no var / std wanted, window too
// wide for a wave tile, ring too large for the LDS), > 0 = error.
// `todo` (fb_groups_x * ceil(rows / fb_tile_rows) bytes, device) is cleared here.
int launch_box_sep(const float *in, float *out_sum, float *out_mean, float *out_var, float *out_std,
                   long rows, long cols, long ld_in, long ld_out, int krows, int kcols, int halo_top, int halo_bot,
                   unsigned char *todo, long fb_groups_x, int fb_tile_rows, int fb_group_cols, hipStream_t s) {
    constexpr int NC = XRS_BOX_NC;
    using G = BoxGeo<NC>;
    if (!todo || !(out_var || out_std) || krows < 3 || kcols < 3 || !(krows & 1) || !(kcols & 1) || kcols > 65 || 2 * (kcols / 2) > G::TWC / 2) return -1;
    BoxArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out; a.halo_top = halo_top; a.halo_bot = halo_bot;
    a.out_sum = out_sum; a.out_mean = out_mean; a.out_var = out_var; a.out_std = out_std;
    a.rx = kcols / 2; a.ry = krows / 2;
    a.w_out = ((G::TWC - 2 * a.rx) / NC) * NC;
    const bool with_q = true;
    a.n_slots = 2 * a.ry + 2 + box_ahead(with_q);
    a.wave_lds = (a.n_slots * G::ROWB + (with_q ? 4 : 2) * G::PSLOTS * (int)sizeof(double) + 15) & ~15;
    if ((unsigned long)ld_in * 4ul >= (1ul << 31)) return -1;            // (the second row of a DMA pair is addressed by a 32-bit lane offset)
    const size_t lds = 4 * (size_t)a.wave_lds;
    if (lds > 156 * 1024) return -1;
    a.n_full = (double)krows * kcols;
    a.inv_n_full = 1.0 / a.n_full;
    a.todo = todo; a.fb_groups_x = fb_groups_x; a.fb_tile_rows = fb_tile_rows; a.fb_group_cols = fb_group_cols;
    a.tiles_x = (cols + a.w_out - 1) / a.w_out;
    a.groups_x = (a.tiles_x + 3) / 4;
    const int om = (out_sum ? BOX_SUM : 0) | (out_mean ? BOX_MEAN : 0) | (out_var ? BOX_VAR : 0) | (out_std ? BOX_STD : 0);
    constexpr int ALL = BOX_SUM | BOX_MEAN | BOX_VAR | BOX_STD, MVS = BOX_MEAN | BOX_VAR | BOX_STD;
    // the instantiation: the common sets exactly, anything else through the superset that has them (absent planes are NULL)
    BoxKernel fn;
    int kind;
    if (om == MVS) { fn = &box_sep_kernel<MVS, NC>; kind = 0; }
    else if (om == ALL) { fn = &box_sep_kernel<ALL, NC>; kind = 1; }
    else { fn = &box_sep_kernel<ALL, NC, 1>; kind = 2; }               // any other subset: absent planes are NULL
    int dev = 0;
    XRS_HIP(hipGetDevice(&dev));
    if (lds > 64 * 1024) {
        // dynamic LDS beyond 64 KiB has to be allowed per kernel and device (the call synchronises: once per thread, kernel, device)
        static thread_local unsigned long long allowed[3] = {0, 0, 0};
        if (dev < 0 || dev >= 64 || !(allowed[kind] >> dev & 1)) {
            XRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
            if (dev >= 0 && dev < 64) allowed[kind] |= 1ull << dev;
        }
    }
    // tile height: whole rounds of resident workgroups, each tile paying 2 ry rows of run-in (as walk3_tile_base)
    static thread_local int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t prop;
        n_cu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    int wg_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_cu, reinterpret_cast<const void *>(fn), 256, lds) != hipSuccess || wg_cu < 1) wg_cu = 1;
    const long slots = (long)n_cu * wg_cu;
    int best = 256;
    double best_cost = 1e300;
    const char *force = ab_env("XRS_BOX_TILE_ROWS");
    for (int tr = 128; tr <= 512; tr += 32) {
        const long ty = (rows + tr - 1) / tr;
        const long rounds = (a.groups_x * ty + slots - 1) / slots;
        const double cost = (double)rounds * (double)(tr + 2 * a.ry);
        if (cost < best_cost) { best_cost = cost; best = tr; }
    }
    a.tile_rows = force && atoi(force) >= 8 ? atoi(force) : best;
    a.tiles_y = (rows + a.tile_rows - 1) / a.tile_rows;
    a.rim_first = 1;
    const long grid = RimFirst(a.groups_x, a.tiles_y, a.rim_first).grid();
    if (grid > 0x7fffffffL) return fail("box statistics: raster too large for one launch");
    const long fb_tiles_y = (rows + fb_tile_rows - 1) / fb_tile_rows;
    XRS_HIP(hipMemsetAsync(todo, 0, (size_t)(fb_groups_x * fb_tiles_y), s));
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace xrs
