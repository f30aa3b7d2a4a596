// Fused raster pass: any subset of the 3x3 terrain products (slope, aspect, curvature, hillshade) PLUS a
// focal mean over a 3x3 / 5x5 mask, from ONE read of the raster.
//
// The reference runs every product as its own full pass over the DataArray
//   hillshade(agg)            xrspatial/hillshade.py:20-35
//   slope(agg) / aspect(agg)  xrspatial/slope.py:56-76, xrspatial/aspect.py:56-90
//   curvature(agg)            xrspatial/curvature.py:31-49
//   focal.apply(agg, kernel)  xrspatial/focal.py:305-326 with _calc_mean (:226-228)
// so a hillshade + focal-mean pipeline moves 16 B per cell (two reads, two writes).  All these products are
// functions of the same small neighbourhood, the work is HBM-bound, and the register-resident strip layout of
// the stand-alone kernels (strip.h) already holds a superset of the 3x3 neighbourhood when it holds the 5x5
// one -- so the fused kernel reads each cell once (4 B) and writes 4 B per product: 12 B per cell for
// hillshade + focal mean, 16 B for hillshade + slope + focal mean (the 65536^2 target pipeline, 24 B unfused).
// Arithmetic is the stand-alone kernels' (terrain_cells.h, focal_mean_direct_kernel): results are
// bit-identical to separate launches, which tests/test_gpu_parity.py asserts.
//
// Any width / pitch / base address (dword-aligned 16-byte accesses, ragged last lane).  Masks larger than 5x5 run as
// the separate launches -- same results, no fusion.
#include <type_traits>

#include "strip.h"
#include "terrain_cells.h"

using namespace xrs;

namespace {

struct PassArgs {
    const float *in;
    float *out[4];            // slope, aspect, curvature, hillshade (null = not requested)
    float *focal;             // focal mean
    long rows, cols, ld_in, ld_out;
    int halo_top, halo_bot;
    double inv8cx, inv8cy, curv_scale;
    float sin_alt, cos_alt, cos_az, sin_az;
    int nt_stores;            // host-side choice of the kernel variant: every output row is 16-byte aligned
    unsigned mask_rows[5];    // bit kx of entry ky: tap (ky, kx) selected
    double inv_ntaps;
    int ntaps;
    long tiles_x, n_tiles;
    // xrs_raster_pass_edges_f32: the launch covers tile rows [0, seg_tiles_y) and, after a gap of seg_skip tile rows, the rest
    // (0 / 0: every tile row, in order)
    long seg_tiles_y, seg_skip;
};

// A lane's 4 results.  NT (compile-time: every output row 16-byte aligned): one non-temporal 16-byte store -- results
// are written once and never read back, so they should not displace the halo rows / columns neighbouring strips re-read
// through L2 (same-box A/B: 2-4 % on the fused kernels; as a RUN-time flag it cost registers and 29 % on the slope
// variant).  Otherwise one 16-byte store at dword alignment (any pitch), or the first `n` cells for the last lane of a
// row whose width is not a multiple of 4.
typedef float v4f_nt __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void put4(float *p, const float (&v)[4], int n = 4) {
    if (n < 4) {
        p[0] = v[0];
        if (n > 1) p[1] = v[1];
        if (n > 2) p[2] = v[2];
    } else if (NT) {
        const v4f_nt q = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<v4f_nt *>(p));
    } else {
        store_f4u(p, v[0], v[1], v[2], v[3]);
    }
}

// OPS: compile-time superset of the terrain products this instantiation can emit (absent ones are skipped by
// wave-uniform null tests, like terrain.hip's fused kernel).
// CMASK: the window's taps as a compile-time constant (bit ky * KW + kx), 0 = read them from a.mask_rows at run time.
// With the mask known, the tap walk is straight-line code with no scalar branch per tap row (the circular 5x5 mask of
// `circle_kernel(1, 1, 2)` -- the bench's -- is instantiated this way).
#ifndef XRS_PASS_SHARED_ROWS
#define XRS_PASS_SHARED_ROWS 1
#endif
template <int OPS, int KH, int KW, int RB, bool INTERIOR, bool NT, unsigned CMASK = 0u>
__device__ __forceinline__ void pass_body(const PassArgs &a, long x_tile, long y0, int lane) {
    constexpr int RX = KW / 2, RY = KH / 2, NV = 4 + 2 * RX, NR = RB + KH - 1;
    const unsigned loff = (unsigned)lane * 4u;
    const long x0 = x_tile + lane * 4;
    float v[NR][NV];
    load_strip<KH, KW, RB, INTERIOR>(a, x_tile, y0, lane, v);

    // which rows hold a NaN / inf (wave-uniform): gates the repair in strip_focal_mean, and tells the terrain products of a
    // clean strip that their neighbourhoods are finite
    const unsigned rows_hit = strip_probe_rows(v);

    // ---- terrain products from the 3x3 centre of the registers (first: they need no accumulators)
    auto terrain = [&](auto finite) {
        constexpr bool FINITE = decltype(finite)::value;
        const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
        const float qnan = nan_f32();
        // (slope / aspect: Horn sums cell by cell here -- sharing the differences along the strip, terrain.hip's HornRoller,
        //  needs 40 more VGPRs next to the focal accumulators and costs this kernel an occupancy step: 0.99 vs 0.87 ms)
        constexpr bool HORN = (OPS & (OP_SLOPE | OP_ASPECT)) != 0;
        const SlopeK sk = slope_constants(a.inv8cx, a.inv8cy);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const long y = y0 + r;
            if (!INTERIOR && y >= a.rows) break;
            const bool row_border = !INTERIOR && ((y - 1 < y_lo) || (y + 1 >= y_hi));
            float o_slope[4], o_aspect[4], o_curv[4], o_hill[4];
            Horn hs[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const long x = x0 + o;
                const bool border = !INTERIOR && (row_border || x == 0 || x == a.cols - 1);
                const int c = RX - 1 + o, t = r + RY - 1;
                Nb q;
                q.nw = v[t][c];     q.n = v[t][c + 1];     q.ne = v[t][c + 2];
                q.w = v[t + 1][c];  q.c = v[t + 1][c + 1]; q.e = v[t + 1][c + 2];
                q.sw = v[t + 2][c]; q.s = v[t + 2][c + 1]; q.se = v[t + 2][c + 2];
                if (HORN) hs[o] = horn_cell(q);
                if ((OPS & OP_SLOPE) && a.out[0]) o_slope[o] = border ? qnan : slope_from_horn(hs[o], sk);
                if ((OPS & OP_ASPECT) && a.out[1]) o_aspect[o] = border ? qnan : aspect_from_horn(hs[o]);
                if ((OPS & OP_CURV) && a.out[2]) o_curv[o] = border ? qnan : curvature_cell(q, a.curv_scale);
                if ((OPS & OP_HILL) && a.out[3])
                    o_hill[o] = border ? qnan : hillshade_cell<FINITE>(q, a.sin_alt, a.cos_alt, a.cos_az, a.sin_az);
            }
            const long off = y * a.ld_out + x_tile;
            const int nown = INTERIOR ? 4 : (int)(a.cols - x0 < 4 ? a.cols - x0 : 4);
            if ((OPS & OP_SLOPE) && a.out[0]) put4<NT>(a.out[0] + off + loff, o_slope, nown);
            if ((OPS & OP_ASPECT) && a.out[1]) put4<NT>(a.out[1] + off + loff, o_aspect, nown);
            if ((OPS & OP_CURV) && a.out[2]) put4<NT>(a.out[2] + off + loff, o_curv, nown);
            if ((OPS & OP_HILL) && a.out[3]) put4<NT>(a.out[3] + off + loff, o_hill, nown);
        }
    };
    // (two copies only where the clean one is shorter: hillshade on interior strips)
    if (INTERIOR && (OPS & OP_HILL) && rows_hit == 0u)
        terrain(std::true_type{});
    else
        terrain(std::false_type{});

    // ---- focal mean: strip.h's strip_focal_mean (float64 sums == numba nanmean over the window; one body for clean strips,
    // strips with nodata and strips on the raster's edge).  (Round 2 tried float32 sums of values shifted by the lane's
    // centre cell: half the VALU cycles, no spills -- and the same 0.57 ms, because this kernel runs at the streaming ceiling
    // of its 1-read / 2-write traffic mix (experiments/rw_mix.hip); the float64 sums keep results independent of how a
    // raster is cut into strips, bands or shards.)
    // Shared row sums (make_row_plan): same-box A/B hillshade + 5x5 mean 0.577 -> 0.570 ms, + slope 0.808 -> 0.798; NOT for the
    // instantiations with aspect, which the row sums push from 168 to 178 VGPRs = from 3 to 2 waves per SIMD (all four
    // products + mean: 1.19 -> 1.37 ms)
    float *fout = a.focal + y0 * a.ld_out + x_tile;
    const int nown = INTERIOR ? 4 : (int)(a.cols - x0 < 4 ? a.cols - x0 : 4);
    strip_focal_mean<KH, KW, RB, CMASK, XRS_PASS_SHARED_ROWS && !(OPS & OP_ASPECT), OPS == 0, INTERIOR>(
        v, a.mask_rows, a.ntaps, a.inv_ntaps, rows_hit, [&](int r, const float (&m)[4]) {
            if (!INTERIOR && y0 + r >= a.rows) return;
            put4<NT>(fout + r * a.ld_out + loff, m, nown);
        });
}

template <int OPS, int KH, int KW, int RB, bool NT, unsigned CMASK = 0u>
// instantiations with slope / aspect: 3 workgroups per CU = 168 VGPRs (they take 155-167 without scratch; the all-four one would
// take 177 = 2 waves per SIMD if allowed to, and holds 168 with 4 spilled registers; capped at 4 workgroups per CU they spill
// 31-45 registers in the hot path: hillshade + slope + 5x5 mean 0.81 -> 1.18 ms)
#ifndef XRS_LB_PASS_HORN
#define XRS_LB_PASS_HORN 3
#endif
#ifndef XRS_LB_PASS
#define XRS_LB_PASS 4
#endif
// Waves per SIMD, capped: since the nodata handling moved out of line these kernels need 59-111 (3x3) / 85-139 (5x5) VGPRs and would
// run 4-8 waves; the streams they read and write like fewer (same-box A/B, profiles/r05/ab_wave_caps.log: hillshade + 3x3 mean
// 0.515 -> 0.489 ms at 3; 5x5 mean alone 0.368 -> 0.363 at 4; hillshade + 5x5 mean 0.548 / 0.562 / 0.546 uncapped / 3 / 4).
#ifndef XRS_PASS_WAVES3
#define XRS_PASS_WAVES3 3
#endif
#ifndef XRS_PASS_WAVES5
#define XRS_PASS_WAVES5 4
#endif
__attribute__((amdgpu_waves_per_eu(1, KH == 3 ? XRS_PASS_WAVES3 : XRS_PASS_WAVES5)))
__global__ void __launch_bounds__(256, (OPS & (OP_SLOPE | OP_ASPECT)) ? XRS_LB_PASS_HORN : XRS_LB_PASS) raster_pass_kernel(const PassArgs a) {
    const long t = xcd_tile(blockIdx.x, a.n_tiles, a.tiles_x);
    if (t < 0) return;
    const long tyl = t / a.tiles_x, tx = t - tyl * a.tiles_x;
    const long ty = tyl >= a.seg_tiles_y ? tyl + a.seg_skip : tyl;          // (two row segments in one launch: see PassArgs)
    const int lane = threadIdx.x & 63;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long x_tile = tx * 256;
    const long y0 = ty * (4 * RB) + (long)wy * RB;
    if (y0 >= a.rows) return;
    if (strip_is_interior<KH, KW, RB>(a, x_tile, y0)) {
        pass_body<OPS, KH, KW, RB, true, NT, CMASK>(a, x_tile, y0, lane);
        return;
    }
    if (x_tile + lane * 4 >= a.cols) return;
    pass_body<OPS, KH, KW, RB, false, NT, CMASK>(a, x_tile, y0, lane);
}

template <int OPS, int K>
int launch_pass(PassArgs &a, hipStream_t s) {
#ifndef XRS_PASS_RB
#define XRS_PASS_RB 4
#endif
    constexpr int RB = XRS_PASS_RB;
    a.tiles_x = (a.cols + 255) / 256;
    a.n_tiles = a.tiles_x * ((a.rows + 4 * RB - 1) / (4 * RB));
    if (a.seg_skip > 0) a.n_tiles -= a.tiles_x * a.seg_skip;               // (the caller made the segments whole tile rows)
    const long grid = xcd_grid(a.n_tiles, a.tiles_x);
    if (grid > 0x7fffffffL) return fail("raster pass: raster too large for one launch");
    // circle_kernel(1, 1, 2): rows 00100 / 01110 / 11111 / 01110 / 00100
    constexpr unsigned CIRCLE5 = 4u | 14u << 5 | 31u << 10 | 14u << 15 | 4u << 20;
    unsigned mask = 0;
    for (int ky = 0; ky < K; ++ky) mask |= a.mask_rows[ky] << (ky * K);
    constexpr unsigned BOX = (1u << (K * K)) - 1u;                                              // np.ones((k, k))
    if (K == 5 && mask == CIRCLE5) {
        if (a.nt_stores)
            hipLaunchKernelGGL((raster_pass_kernel<OPS, K, K, RB, true, K == 5 ? CIRCLE5 : 0u>), dim3((unsigned)grid), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((raster_pass_kernel<OPS, K, K, RB, false, K == 5 ? CIRCLE5 : 0u>), dim3((unsigned)grid), dim3(256), 0, s, a);
    } else if (K == 3 && mask == BOX) {
        if (a.nt_stores)
            hipLaunchKernelGGL((raster_pass_kernel<OPS, K, K, RB, true, K == 3 ? BOX : 0u>), dim3((unsigned)grid), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((raster_pass_kernel<OPS, K, K, RB, false, K == 3 ? BOX : 0u>), dim3((unsigned)grid), dim3(256), 0, s, a);
    } else if (a.nt_stores)
        hipLaunchKernelGGL((raster_pass_kernel<OPS, K, K, RB, true>), dim3((unsigned)grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((raster_pass_kernel<OPS, K, K, RB, false>), dim3((unsigned)grid), dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

// Instantiated product sets.  The focal mean fuses well with every terrain product (measured on 16384^2:
// hillshade+focal 0.57 ms vs 0.79 ms as two launches, hillshade+slope+focal 0.84 vs 1.19).  Round 1 left aspect to
// terrain.hip (its library atan2 on top of float64 Horn sums made the fused kernel VALU-bound); with the float32 Horn
// sums and the octant-folded arc tangent of terrain_cells.h it rides along (own instantiations for aspect alone and
// aspect + hillshade; any other set with aspect takes the all-four instantiation).
constexpr int FUSABLE = OP_SLOPE | OP_ASPECT | OP_CURV | OP_HILL;

template <int K>
int launch_pass_ops(PassArgs &a, int ops, hipStream_t s) {
    switch (ops) {
        // aspect alone / with hillshade: their own instantiations (the all-four kernel's 168 VGPRs are slope + aspect +
        // curvature together); every other set with aspect takes the all-four kernel, absent products skipped by
        // wave-uniform tests
        case OP_ASPECT: return launch_pass<OP_ASPECT, K>(a, s);
        case OP_ASPECT | OP_HILL: return launch_pass<OP_ASPECT | OP_HILL, K>(a, s);
        case 0: return launch_pass<0, K>(a, s);                              // the focal mean alone (compile-time masks only)
        case OP_HILL: return launch_pass<OP_HILL, K>(a, s);
        case OP_SLOPE: return launch_pass<OP_SLOPE | OP_HILL, K>(a, s);     // (absent hillshade skipped by a wave-uniform test;
                                                                             //  measured faster than a slope-only instantiation)
        case OP_CURV: return launch_pass<OP_CURV, K>(a, s);
        case OP_SLOPE | OP_HILL: return launch_pass<OP_SLOPE | OP_HILL, K>(a, s);
        case OP_CURV | OP_HILL: return launch_pass<OP_CURV | OP_HILL, K>(a, s);
        case OP_SLOPE | OP_CURV: return launch_pass<OP_SLOPE | OP_CURV, K>(a, s);
        case OP_SLOPE | OP_CURV | OP_HILL: return launch_pass<OP_SLOPE | OP_CURV | OP_HILL, K>(a, s);
        default:
            if (ops & OP_ASPECT) return launch_pass<FUSABLE, K>(a, s);
            return fail("raster pass: internal error, product set %d has no fused kernel", ops);
    }
}

}  // namespace

namespace xrs {
// circle_kernel(1, 1, 2) or np.ones((3, 3)): the masks raster_pass_kernel is specialised for
bool pass_has_compile_time_mask(const double *kernel, int krows, int kcols) {
    if (!kernel || krows != kcols) return false;
    if (krows == 3) {
        for (int i = 0; i < 9; ++i)
            if (kernel[i] != 1.0) return false;
        return true;
    }
    if (krows != 5) return false;
    static const int rows5[5] = {4, 14, 31, 14, 4};
    for (int ky = 0; ky < 5; ++ky)
        for (int kx = 0; kx < 5; ++kx)
            if ((kernel[ky * 5 + kx] == 1.0) != ((rows5[ky] >> kx & 1) != 0)) return false;
    return true;
}
}  // namespace xrs

// edge_rows < 0: the whole raster.  edge_rows >= 0 (xrs_raster_pass_edges_f32): only its first and last edge_rows rows --
// in ONE launch over two segments of tile rows when the fused kernel takes the request, as two calls otherwise.
static int raster_pass(const float *in_dev, float *slope_dev, float *aspect_dev, float *curvature_dev,
                       float *hillshade_dev, float *focal_mean_dev, const double *kernel, int krows,
                       int kcols, void *work_dev, int64_t rows, int64_t cols, int64_t ld_in,
                       int64_t ld_out, double cellsize_x, double cellsize_y, double azimuth,
                       double angle_altitude, int halo_top, int halo_bot, int64_t edge_rows, void *stream) {
    if (!in_dev) return fail("xrs_raster_pass_f32: null input");
    if (focal_mean_dev && (!kernel || krows <= 0 || kcols <= 0 || !(krows & 1) || !(kcols & 1)))
        return fail("xrs_raster_pass_f32: a focal mean needs an odd-shaped kernel");
    int ops = 0;
    if (slope_dev) ops |= OP_SLOPE;
    if (aspect_dev) ops |= OP_ASPECT;
    if (curvature_dev) ops |= OP_CURV;
    if (hillshade_dev) ops |= OP_HILL;
    if (rows <= 0 || cols <= 0 || (!ops && !focal_mean_dev)) return 0;

    const int fused_ops = ops & FUSABLE;
    // The focal mean ALONE also runs here when its mask is one of the compile-time ones (round 3: the stand-alone 5x5
    // kernel of kxk.hip spills when it is given the compile-time mask and the shared row sums this kernel has; this
    // instantiation is that kernel -- kxk.hip's xrs_focal_stats_f32 routes such requests here).
    const bool focal_only = !fused_ops && focal_mean_dev && pass_has_compile_time_mask(kernel, krows, kcols);
    // (any width / pitch / base address: the strip layout's 16-byte accesses only need dword alignment)
    bool fast = (fused_ops || focal_only) && focal_mean_dev && krows == kcols && (krows == 3 || krows == 5) && ld_in >= cols &&
                ld_out >= cols && halo_top >= 0 && halo_bot >= 0;
    float *outs[4] = {slope_dev, aspect_dev, curvature_dev, hillshade_dev};
    bool nt = ld_out % 4 == 0 && aligned16(focal_mean_dev);
    for (int i = 0; i < 4; ++i)
        if (fused_ops >> i & 1) nt = nt && aligned16(outs[i]);

    int ntaps = 0;
    if (fast) {
        for (int i = 0; i < krows * kcols; ++i)
            if (kernel[i] == 1.0) ++ntaps;                    // (cells with any other value are not taps, as in kxk.hip)
        fast = fast && ntaps > 0;
    }
    // products that stay with the stand-alone terrain kernel: all of them without the fused kernel, aspect with it
    const int rest = fast ? (ops & ~FUSABLE) : ops;
    // the two segments as whole rows of tiles (16 raster rows each): [0, seg_a) and [seg_b, end); the rows of the second one above the edge
    // proper are interior rows computed a second time with the same result (the same kernel on the same cells)
    constexpr long TILE_ROWS = 4 * XRS_PASS_RB;
    const long seg_a = edge_rows >= 0 ? (edge_rows + TILE_ROWS - 1) / TILE_ROWS : 0, seg_b = edge_rows >= 0 ? (rows - edge_rows) / TILE_ROWS : 0;
    if (edge_rows == 0) return 0;
    if (edge_rows > 0 && (rest || !fast || seg_b <= seg_a)) {
        if (2 * edge_rows >= rows)
            return raster_pass(in_dev, slope_dev, aspect_dev, curvature_dev, hillshade_dev, focal_mean_dev, kernel, krows, kcols,
                               work_dev, rows, cols, ld_in, ld_out, cellsize_x, cellsize_y, azimuth, angle_altitude, halo_top,
                               halo_bot, -1, stream);
        // two calls: the rows between the edges are the halo of either (as many of them as a window can reach)
        const int64_t between = rows - 2 * edge_rows, off = rows - edge_rows;
        const int inner = (int)(between + edge_rows < 4096 ? between + edge_rows : 4096);
        auto at = [](float *p, int64_t o) { return p ? p + o : nullptr; };
        int rc = raster_pass(in_dev, slope_dev, aspect_dev, curvature_dev, hillshade_dev, focal_mean_dev, kernel, krows, kcols,
                             work_dev, edge_rows, cols, ld_in, ld_out, cellsize_x, cellsize_y, azimuth, angle_altitude, halo_top,
                             inner, -1, stream);
        if (rc) return rc;
        return raster_pass(in_dev + off * ld_in, at(slope_dev, off * ld_out), at(aspect_dev, off * ld_out),
                           at(curvature_dev, off * ld_out), at(hillshade_dev, off * ld_out), at(focal_mean_dev, off * ld_out),
                           kernel, krows, kcols, work_dev, edge_rows, cols, ld_in, ld_out, cellsize_x, cellsize_y, azimuth,
                           angle_altitude, inner, halo_bot, -1, stream);
    }
    if (rest) {
        const int rc = xrs_terrain_fused_f32(in_dev, (rest & OP_SLOPE) ? slope_dev : nullptr,
                                             (rest & OP_ASPECT) ? aspect_dev : nullptr,
                                             (rest & OP_CURV) ? curvature_dev : nullptr,
                                             (rest & OP_HILL) ? hillshade_dev : nullptr, rows, cols, ld_in, ld_out,
                                             cellsize_x, cellsize_y, azimuth, angle_altitude, halo_top, halo_bot,
                                             stream);
        if (rc) return rc;
    }
    if (!fast) {
        if (!focal_mean_dev) return 0;
        float *stat_outs[XRS_NUM_STATS] = {nullptr};
        stat_outs[XRS_STAT_MEAN] = focal_mean_dev;
        return xrs_focal_stats_f32(in_dev, stat_outs, 1u << XRS_STAT_MEAN, rows, cols, ld_in, ld_out, kernel, krows,
                                   kcols, work_dev, halo_top, halo_bot, stream);
    }

    PassArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in_dev; a.focal = focal_mean_dev;
    for (int i = 0; i < 4; ++i) a.out[i] = (fused_ops >> i & 1) ? outs[i] : nullptr;
    a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out;
    a.halo_top = halo_top; a.halo_bot = halo_bot;
    a.nt_stores = nt ? 1 : 0;
    a.inv8cx = 1.0 / (8 * cellsize_x);
    a.inv8cy = 1.0 / (8 * cellsize_y);
    const double cs = (cellsize_x + cellsize_y) / 2;       // curvature.py:241
    a.curv_scale = -200.0 / (cs * cs);
    hillshade_constants(azimuth, angle_altitude, a.sin_alt, a.cos_alt, a.cos_az, a.sin_az);
    for (int ky = 0; ky < krows; ++ky)
        for (int kx = 0; kx < kcols; ++kx)
            if (kernel[ky * kcols + kx] == 1.0) a.mask_rows[ky] |= 1u << kx;
    a.inv_ntaps = 1.0 / ntaps;
    a.ntaps = ntaps;
    if (edge_rows >= 0) { a.seg_tiles_y = seg_a; a.seg_skip = seg_b - seg_a; }
    return krows == 3 ? launch_pass_ops<3>(a, fused_ops, as_stream(stream)) : launch_pass_ops<5>(a, fused_ops, as_stream(stream));
}

extern "C" int xrs_raster_pass_f32(const float *in_dev, float *slope_dev, float *aspect_dev, float *curvature_dev,
                                   float *hillshade_dev, float *focal_mean_dev, const double *kernel, int krows,
                                   int kcols, void *work_dev, int64_t rows, int64_t cols, int64_t ld_in,
                                   int64_t ld_out, double cellsize_x, double cellsize_y, double azimuth,
                                   double angle_altitude, int halo_top, int halo_bot, void *stream) {
    return raster_pass(in_dev, slope_dev, aspect_dev, curvature_dev, hillshade_dev, focal_mean_dev, kernel, krows, kcols,
                       work_dev, rows, cols, ld_in, ld_out, cellsize_x, cellsize_y, azimuth, angle_altitude, halo_top,
                       halo_bot, -1, stream);
}

extern "C" int xrs_raster_pass_edges_f32(const float *in_dev, float *slope_dev, float *aspect_dev, float *curvature_dev,
                                         float *hillshade_dev, float *focal_mean_dev, const double *kernel, int krows,
                                         int kcols, void *work_dev, int64_t rows, int64_t cols, int64_t ld_in,
                                         int64_t ld_out, double cellsize_x, double cellsize_y, double azimuth,
                                         double angle_altitude, int halo_top, int halo_bot, int64_t edge_rows,
                                         void *stream) {
    if (edge_rows < 0) return fail("xrs_raster_pass_edges_f32: edge_rows must not be negative");
    return raster_pass(in_dev, slope_dev, aspect_dev, curvature_dev, hillshade_dev, focal_mean_dev, kernel, krows, kcols,
                       work_dev, rows, cols, ld_in, ld_out, cellsize_x, cellsize_y, azimuth, angle_altitude, halo_top,
                       halo_bot, edge_rows, stream);
}
