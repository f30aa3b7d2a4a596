// Separable statistics over BOX masks -- np.ones((k, k)), the kernels the reference's own benchmark suite runs
// (benchmarks/benchmarks/focal.py:10-34: custom_kernel(np.ones(...)) for apply / focal_stats / hotspots) -- in O(1) work
// per cell whatever k: focal mean / sum / var / std (xrspatial/focal.py:226-258 through _apply_numpy :305-326) and
// convolve_2d with one weight value on the box (convolution.py:285-313, what focal.hotspots is fed).
//
// A box is the one mask whose window sum factors: sum over the window = sum over its columns of (sum over the column's
// rows).  So instead of 2R+1 ring additions per cell, row and moment (mom_impl.h, wide_impl.h: ~163 VALU instructions per
// cell for the four moments at 25x25), a wave keeps ONE running column sum per column and moment:
//   * a wave owns 256 columns x ~256 output rows and walks down; a lane owns 4 adjacent columns.  Per output row it loads
//     the row ENTERING the window (y + ry) and the row LEAVING it (y - ry; read 2 ry + 1 rows earlier by the same wave, so
//     it comes from L2 / the Infinity Cache, not from HBM) -- two 16-byte loads per lane, batches of 4 rows in flight;
//   * the column sums are FLOAT64 sums of d = v - c0 and d^2 (c0 = the cell at the tile centre): d is exact, the sums of d
//     are exact, those of d^2 carry 2^-53 relative per update, so the add / subtract recurrence does not drift in any way
//     float32 results can see, and the shift keeps var = (Q - S^2 / n) / n well conditioned: Q / (n var) = 1 + m^2 / var with
//     m <= the tile's relief, against a guard at 2^24 (1 ulp of float32);
//   * the horizontal box sum of the 256 column sums: lane-local prefix over the 4 columns, a wave-wide DPP scan of the lane
//     totals (6 steps), the prefix array through LDS, and every output column is P[x + rx] - P[x - rx - 1] -- independent of
//     rx.  A wave writes the 256 - 2 rx (rounded down to 4) columns whose windows lie inside its 256 columns;
//   * nothing here knows the box size at compile time: one kernel for every k (and for rectangles);
//   * ~40 VALU instructions per cell (~60 % of them float64) for all four moments, few registers (no ring), 5+ waves per
//     SIMD: the kernel is bound by the 4 B read + 4 B per plane written.
// What the fast walk cannot do -- NaN / inf cells (they would stay in a running sum for ever), windows whose variance drowns
// in the cancellation (flat patches away from the shift: exact zero is the contract there) -- it does not try: the wave
// marks its tile in `todo`, a byte map over the workgroup tiles of the float32 walker that owns the mask shape
// (focal_mom_kernel / focal_wide_kernel with BoxShape), and that kernel, launched right behind this one, redoes exactly
// the marked tiles with its NaN-aware and exact paths.  Raster edges stay here (clipped windows: n = rows x cols inside,
// cells outside contribute d = 0).
#include "circle_walk.h"
#include "lds_dma.h"

using namespace xrs;

namespace {

enum : int { BOX_SUM = 1, BOX_MEAN = 2, BOX_VAR = 4, BOX_STD = 8, BOX_CONV = 16 };

struct BoxArgs {
    const float *in;
    long rows, cols, ld_in, ld_out;
    int halo_top, halo_bot;
    float *out_sum, *out_mean, *out_var, *out_std, *out_conv;
    int rx, ry;                   // half-widths of the box (columns, rows)
    int w_out;                    // output columns per wave tile: (256 - 2 rx) rounded down to a multiple of 4
    int tile_rows;                // output rows per wave tile
    long tiles_x, tiles_y, groups_x;   // wave tiles; workgroups = 4 horizontally adjacent wave tiles
    int rim_first;
    double n_full, inv_n_full;    // cells of an unclipped window
    double wgt;                   // BOX_CONV: the weight
    // the fall-back kernel's workgroup tiles (todo[ty * fb_groups_x + gx] = 1: redo rows [ty * fb_tile_rows, +fb_tile_rows)
    // x columns [gx * fb_group_cols, +fb_group_cols))
    unsigned char *todo;
    long fb_groups_x;
    int fb_tile_rows, fb_group_cols;
};

constexpr int BOX_U_INTERIOR = 4;     // rows per batch of loads (edge tiles: 1 -- their predicated walk would otherwise set
                                      // the kernel's register count: 168 instead of 126)
constexpr int BOX_SLOTS = 328;        // prefix slots per wave and moment: 1 sentinel + 256 columns + what idle lanes read

// ---- wave-wide inclusive scan of one float64 per lane: Hillis-Steele inside the rows of 16 lanes (row_shr 1, 2, 4, 8), then
// the row totals across (row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3).  A lane without a source keeps 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_scan_f64(double v) {
    v += dpp_f64<0x111, 0xf>(v);
    v += dpp_f64<0x112, 0xf>(v);
    v += dpp_f64<0x114, 0xf>(v);
    v += dpp_f64<0x118, 0xf>(v);
    v += dpp_f64<0x142, 0xa>(v);
    v += dpp_f64<0x143, 0xc>(v);
    return v;
}

template <int OM, bool EDGE>
struct BoxWalk {
    static constexpr bool Q = (OM & (BOX_VAR | BOX_STD)) != 0;
    static constexpr int BOX_U = EDGE ? 1 : BOX_U_INTERIOR;
    const BoxArgs &a;
    double *p1, *p2;              // this wave's prefix arrays (slot 0 = 0; slot 1 + i = columns 0 .. i)
    long xs, x_out0, y0, y_end;
    int lane, rx, ry;
    double c0;
    float c0f;
    double C1[4], C2[4];          // (C2 unused -- and optimised away -- without the squares)
    unsigned long long failm;     // lanes with a result the fast walk must not stand for (wave-uniform)

    __device__ __forceinline__ BoxWalk(const BoxArgs &a_, double *p1_, double *p2_, long xs_, long xo, long y0_, long ye, int lane_)
        : a(a_), p1(p1_), p2(p2_), xs(xs_), x_out0(xo), y0(y0_), y_end(ye), lane(lane_), rx(a_.rx), ry(a_.ry) {}

    // the lane's 4 cells of input row yy (EDGE: c0 for everything outside the raster / the shard's halo rows: d = 0)
    __device__ __forceinline__ void load4(long yy, float (&v)[4]) const {
        if (!EDGE) {
            const xrs_f4u q = load_f4u(a.in + yy * a.ld_in + xs + 4 * lane);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = c0f;
        if (yy < -(long)a.halo_top || yy >= a.rows + a.halo_bot) return;          // wave-uniform
        const float *p = a.in + yy * a.ld_in;
        const long x = xs + 4 * lane;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x + j >= 0 && x + j < a.cols) v[j] = p[x + j];
    }

    __device__ __forceinline__ void enter(const float (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double d = (double)v[j] - c0;
            C1[j] += d;
            if (Q) C2[j] = fma(d, d, C2[j]);
        }
    }
    __device__ __forceinline__ void leave(const float (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double d = (double)v[j] - c0;
            C1[j] -= d;
            if (Q) C2[j] = fma(-d, d, C2[j]);
        }
    }

    // box sums of the column sums for the lane's 4 OUTPUT columns x_out0 + 4 lane + o (column index rx + 4 lane + o of the tile)
    __device__ __forceinline__ void across(const double (&C)[4], double *p, double (&B)[4]) const {
        double pre[4];
        pre[0] = C[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) pre[j] = pre[j - 1] + C[j];
        const double before = wave_scan_f64(pre[3]) - pre[3];        // the columns of the lanes to the left
#pragma unroll
        for (int j = 0; j < 4; ++j) p[1 + 4 * lane + j] = before + pre[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                                // (LDS serves one wave's instructions in order)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int o = 0; o < 4; ++o) B[o] = p[1 + 4 * lane + o + 2 * rx] - p[4 * lane + o];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                                // (the next row's writes come after these reads)
    }

    __device__ __forceinline__ void emit(long yo) {
        double B1[4], B2[4];
        across(C1, p1, B1);
        if (Q) across(C2, p2, B2);
        else { B2[0] = B2[1] = B2[2] = B2[3] = 0.0; }
        const bool out_lane = 4 * lane < a.w_out;
        const long xo = x_out0 + 4 * lane;
        float r_sum[4], r_mean[4], r_var[4], r_std[4], r_conv[4];
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
        for (int o = 0; o < 4; ++o) {
            double n = a.n_full, inv = a.inv_n_full;
            bool full = true;
            if (EDGE) {
                const long lo = xo + o - rx < 0 ? 0 : xo + o - rx;
                const long hi = xo + o + rx >= a.cols ? a.cols - 1 : xo + o + rx;
                full = rows_full && hi - lo == 2 * rx;
                if (!full) { n = ny * (double)(hi - lo + 1); inv = 1.0 / n; }
            }
            // mean of the shifted values, exact whenever it is representable (a flat window: d itself)
            double q = B1[o] * inv;
            q = fma(fma(-q, n, B1[o]), inv, q);
            if (Q) {
                const double e = fma(-B1[o], q, B2[o]);                  // n * variance
                bad |= !(e >= 0x1p-24 * B2[o]);                         // (a NaN / inf anywhere fails it too)
                const double var = e * inv;
                r_var[o] = (float)var;
                r_std[o] = sqrtf((float)var);
            } else {
                bad |= !(fabs(B1[o]) < INFINITY);
            }
            r_mean[o] = (float)(c0 + q);
            r_sum[o] = (float)fma(n, c0, B1[o]);
            if (OM & BOX_CONV) r_conv[o] = full ? (float)(a.wgt * fma(n, c0, B1[o])) : nan_f32();
        }
        const bool live = out_lane && (!EDGE || xo < a.cols);
        failm |= __builtin_amdgcn_ballot_w64(live && bad);
        if (!live) return;
        const long off = yo * a.ld_out + xo;
        if (!EDGE) {
            typedef float st4 __attribute__((ext_vector_type(4), aligned(4)));
            auto put = [&](float *plane, const float (&r)[4]) {
                st4 v; v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
                __builtin_nontemporal_store(v, reinterpret_cast<st4 *>(plane + off));
            };
            if ((OM & BOX_SUM) && a.out_sum) put(a.out_sum, r_sum);
            if ((OM & BOX_MEAN) && a.out_mean) put(a.out_mean, r_mean);
            if ((OM & BOX_VAR) && a.out_var) put(a.out_var, r_var);
            if ((OM & BOX_STD) && a.out_std) put(a.out_std, r_std);
            if ((OM & BOX_CONV) && a.out_conv) put(a.out_conv, r_conv);
        } else {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (xo + o >= a.cols) break;
                if ((OM & BOX_SUM) && a.out_sum) a.out_sum[off + o] = r_sum[o];
                if ((OM & BOX_MEAN) && a.out_mean) a.out_mean[off + o] = r_mean[o];
                if ((OM & BOX_VAR) && a.out_var) a.out_var[off + o] = r_var[o];
                if ((OM & BOX_STD) && a.out_std) a.out_std[off + o] = r_std[o];
                if ((OM & BOX_CONV) && a.out_conv) a.out_conv[off + o] = r_conv[o];
            }
        }
    }

    // true: every result of the tile stands; false: the tile goes to the fall-back kernel
    __device__ __forceinline__ bool run() {
        failm = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) C1[j] = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) C2[j] = 0.0;
        // the shift: the cell at the tile centre (any finite value works; a near one keeps d small)
        {
            const long yc = y0 + (y_end - y0) / 2;
            long xc = x_out0 + a.w_out / 2;
            xc = xc < a.cols ? xc : a.cols - 1;
            const float v = a.in[yc * a.ld_in + xc];
            c0f = isfinite(v) ? v : 0.0f;
            c0 = (double)c0f;
        }
        if (lane == 0) { p1[0] = 0.0; if (Q) p2[0] = 0.0; }
        // run-in: rows y0 - ry .. y0 + ry - 1
        for (long yy = y0 - ry; yy < y0 + ry; yy += BOX_U) {
            float v[BOX_U][4];
#pragma unroll
            for (int u = 0; u < BOX_U; ++u) {
                const long yr = yy + u < y0 + ry ? yy + u : y0 + ry - 1;     // (clamped: loaded, not used)
                load4(yr, v[u]);
            }
#pragma unroll
            for (int u = 0; u < BOX_U; ++u)
                if (yy + u < y0 + ry) enter(v[u]);
        }
        const long y_last_in = EDGE ? (long)0x7fffffffffffL : y_end - 1 + ry;      // (EDGE tests every row itself)
        for (long yo = y0; yo < y_end; yo += BOX_U) {
            float e[BOX_U][4], l[BOX_U][4];
#pragma unroll
            for (int u = 0; u < BOX_U; ++u) {
                const long ye = yo + u + ry;
                load4(ye < y_last_in ? ye : y_last_in, e[u]);
                load4(yo + u - ry, l[u]);
            }
#pragma unroll
            for (int u = 0; u < BOX_U; ++u) {
                if (yo + u < y_end) {                                    // wave-uniform
                    enter(e[u]);
                    emit(yo + u);
                    leave(l[u]);
                }
                // (one row at a time: left alone, the scheduler converts all four rows of the batch to float64 up front --
                //  64 more registers -- and interleaves the four emits: 168 VGPRs instead of ~100)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (failm) return false;
        }
        return true;
    }
};

template <int OM>
__global__ void __launch_bounds__(256) box_sep_kernel(const BoxArgs a) {
    __shared__ __attribute__((aligned(16))) double prefix[4][2][BOX_SLOTS];
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
    const bool interior = xs >= 0 && xs + 256 <= a.cols && x_out0 + a.w_out <= a.cols && y0 - a.ry >= -(long)a.halo_top &&
                          y_end + a.ry <= a.rows + a.halo_bot;
    bool ok;
    if (interior) {
        BoxWalk<OM, false> w(a, prefix[wv][0], prefix[wv][1], xs, x_out0, y0, y_end, lane);
        ok = w.run();
    } else {
        BoxWalk<OM, true> w(a, prefix[wv][0], prefix[wv][1], xs, x_out0, y0, y_end, lane);
        ok = w.run();
    }
    if (ok || lane != 0) return;
    // mark every tile of the fall-back kernel that holds cells of this one
    const long x_hi = (x_out0 + a.w_out < a.cols ? x_out0 + a.w_out : a.cols) - 1;
    for (long fy = y0 / a.fb_tile_rows; fy <= (y_end - 1) / a.fb_tile_rows; ++fy)
        for (long fx = x_out0 / a.fb_group_cols; fx <= x_hi / a.fb_group_cols; ++fx) a.todo[fy * a.fb_groups_x + fx] = 1;
}

int wg_per_cu_of(const void *fn) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, 0) != hipSuccess || n < 1) return 4;
    return n;
}

}  // namespace

namespace xrs {

// Launches the fast walk for an all-ones krows x kcols box.  0 = launched (the caller launches its own kernel on the tiles of
// `todo` behind it), -1 = not for this walk (window too wide for a 256-column tile), > 0 = error.
// `todo` (fb_groups_x * ceil(rows / fb_tile_rows) bytes, device) is cleared here.
int launch_box_sep(const float *in, float *out_sum, float *out_mean, float *out_var, float *out_std, float *out_conv, double wgt,
                   long rows, long cols, long ld_in, long ld_out, int krows, int kcols, int halo_top, int halo_bot,
                   unsigned char *todo, long fb_groups_x, int fb_tile_rows, int fb_group_cols, hipStream_t s) {
    if (!todo || krows < 3 || kcols < 3 || !(krows & 1) || !(kcols & 1) || kcols > 65 || krows > 255) return -1;
    BoxArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out; a.halo_top = halo_top; a.halo_bot = halo_bot;
    a.out_sum = out_sum; a.out_mean = out_mean; a.out_var = out_var; a.out_std = out_std; a.out_conv = out_conv;
    a.rx = kcols / 2; a.ry = krows / 2;
    a.w_out = (256 - 2 * a.rx) & ~3;
    a.n_full = (double)krows * kcols;
    a.inv_n_full = 1.0 / a.n_full;
    a.wgt = wgt;
    a.todo = todo; a.fb_groups_x = fb_groups_x; a.fb_tile_rows = fb_tile_rows; a.fb_group_cols = fb_group_cols;
    a.tiles_x = (cols + a.w_out - 1) / a.w_out;
    a.groups_x = (a.tiles_x + 3) / 4;
    const int om = (out_sum ? BOX_SUM : 0) | (out_mean ? BOX_MEAN : 0) | (out_var ? BOX_VAR : 0) | (out_std ? BOX_STD : 0) |
                   (out_conv ? BOX_CONV : 0);
    if (!om) return 0;
    constexpr int ALL = BOX_SUM | BOX_MEAN | BOX_VAR | BOX_STD, MVS = BOX_MEAN | BOX_VAR | BOX_STD;
    // the instantiation: the common sets exactly, anything else through the superset that has them (absent planes are NULL)
    const void *fn;
    int kind;
    if (om == BOX_MEAN) { fn = reinterpret_cast<const void *>(&box_sep_kernel<BOX_MEAN>); kind = 0; }
    else if (om == BOX_SUM) { fn = reinterpret_cast<const void *>(&box_sep_kernel<BOX_SUM>); kind = 1; }
    else if (om == BOX_CONV) { fn = reinterpret_cast<const void *>(&box_sep_kernel<BOX_CONV>); kind = 2; }
    else if (om == MVS) { fn = reinterpret_cast<const void *>(&box_sep_kernel<MVS>); kind = 3; }
    else if (!(om & BOX_CONV)) { fn = reinterpret_cast<const void *>(&box_sep_kernel<ALL>); kind = 4; }
    else return -1;
    // tile height: whole rounds of resident workgroups, each tile paying 2 ry rows of run-in (as walk3_tile_base)
    static thread_local int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }
    static thread_local int wg_cu[5] = {0, 0, 0, 0, 0};
    if (!wg_cu[kind]) wg_cu[kind] = wg_per_cu_of(fn);
    const long slots = (long)n_cu * wg_cu[kind];
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
    const dim3 g((unsigned)grid), b(256);
    switch (kind) {
        case 0: hipLaunchKernelGGL((box_sep_kernel<BOX_MEAN>), g, b, 0, s, a); break;
        case 1: hipLaunchKernelGGL((box_sep_kernel<BOX_SUM>), g, b, 0, s, a); break;
        case 2: hipLaunchKernelGGL((box_sep_kernel<BOX_CONV>), g, b, 0, s, a); break;
        case 3: hipLaunchKernelGGL((box_sep_kernel<MVS>), g, b, 0, s, a); break;
        default: hipLaunchKernelGGL((box_sep_kernel<ALL>), g, b, 0, s, a); break;
    }
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace xrs
