// 3x3 terrain family: slope, aspect, curvature, hillshade (single and fused).
//
// Reference runners replaced (CPU arithmetic is the contract):
//   slope      xrspatial/slope.py:56-76       aspect     xrspatial/aspect.py:56-90
//   curvature  xrspatial/curvature.py:31-49   hillshade  xrspatial/hillshade.py:20-35
//
// Kernel shape (HBM-bound, 4 B in + 4 B out per cell, no MFMA):
//   * a 256-thread workgroup = 4 waves stacked in y; each wave owns a strip of
//     256 columns (64 lanes x one 16-byte float4) by RB rows.
//   * every lane keeps its (RB+2) x 6 neighbourhood in VGPRs: one aligned
//     global_load_dwordx4 per row plus two single-dword loads for the columns
//     left/right of its float4 (those lines are in L1/L2 because the neighbouring
//     lane / strip fetches them as its own float4).  All (RB+2)*3 loads of a lane
//     are independent and issued before the first use, which is what keeps
//     enough bytes in flight to cover HBM latency.
//   * no LDS: a 3-wide window needs no cross-lane traffic beyond the two halo
//     dwords (measured against an LDS-tile variant).
//   * the NaN border is written by the same kernel (the reference pre-fills the
//     output with NaN and overwrites the interior: twice the write traffic).
//   * tiles are numbered row-major and dealt to the XCDs one tile ROW at a time
//     (xrs::xcd_tile): horizontally adjacent strips share an L2, and all XCDs stream
//     through the same rows of the raster together.
// A scalar one-thread-per-cell kernel handles rasters whose width / pitch /
// base address are not multiples of 16 bytes.
#include "terrain_cells.h"

#include <cstdlib>

using namespace xrs;

namespace {

struct TerrainArgs {
    const float *in;
    void *out[4];            // slope, aspect, curvature, hillshade
    long rows, cols, ld_in, ld_out;
    int halo_top, halo_bot;
    double inv8cx, inv8cy;   // slope: 1 / (8 * cellsize)
    double curv_scale;       // curvature: -200 / cellsize^2
    float sin_alt, cos_alt, cos_az, sin_az;   // hillshade: az = (360-azimuth) deg - pi/2
    long tiles_x, n_tiles;
};

template <typename OutT>
__device__ __forceinline__ void store4(OutT *p, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<float>(float *p, const float (&v)[4]) {
    store_f4u(p, v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store4<double>(double *p, const float (&v)[4]) {
    store_d2u(p, (double)v[0], (double)v[1]);
    store_d2u(p + 2, (double)v[2], (double)v[3]);
}
// the last lane of a row when cols % 4 != 0: only the first `n` results exist
template <typename OutT>
__device__ __forceinline__ void store_n(OutT *p, const float (&v)[4], int n) {
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (o < n) p[o] = (OutT)v[o];
}

#ifndef XRS_STRIP_WX
#define XRS_STRIP_WX 1
#endif
#ifndef XRS_HORN_MODE
#define XRS_HORN_MODE 0      // 0: Horn sums cell by cell (horn_cell), 1: differences shared along the strip (HornRoller)
#endif

// ---------------------------------------------------------------- fast path
// INTERIOR: wave-uniform fact that the wave's whole (RB+2) x 258 input window lies inside the raster
// (true for all but the waves along the raster edge): loads are unconditional, addressed as a scalar row
// base + one per-lane offset (global_load ... s[base], offset:imm), and no border/NaN selects exist.
template <int OPS, typename HillT, int RB, bool INTERIOR>
__device__ __forceinline__ void terrain_strip_body(const TerrainArgs &a, long x_tile, long y0, int lane) {
    const long x0 = x_tile + lane * 4;
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;   // valid input rows [y_lo, y_hi)
    const bool has_l = INTERIOR || x0 > 0, has_r = INTERIOR || x0 + 4 < a.cols;
    const unsigned loff = (unsigned)lane * 4u;
    // columns this lane owns (4, or fewer for the last lane of a row whose width is not a multiple of 4)
    const int nown = INTERIOR ? 4 : (a.cols - x0 < 4 ? (int)(a.cols - x0) : 4);

    // v[r][0..5] = columns x0-1 .. x0+4 of input row y0 + r - 1
    float v[RB + 2][6];
#pragma unroll
    for (int r = 0; r < RB + 2; ++r) {
        const long y = y0 + r - 1;
        const bool ok = INTERIOR || (y >= y_lo && y < y_hi);
        const float *rowbase = a.in + y * a.ld_in + x_tile;            // wave-uniform
        const float *p = rowbase + loff;
        float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float l = 0.f, rr = 0.f;
        if (ok) {
            if (INTERIOR || nown == 4) {
                const xrs_f4u q = load_f4u(p);
                c4 = make_float4(q.x, q.y, q.z, q.w);
            } else {
                c4.x = p[0];
                if (nown > 1) c4.y = p[1];
                if (nown > 2) c4.z = p[2];
            }
            if ((OPS & (OP_SLOPE | OP_ASPECT)) || (r >= 1 && r <= RB)) {   // diagonal-free ops need halo columns on centre rows only
                if (has_l) l = p[-1];
                if (has_r) rr = p[4];
            }
        }
        v[r][0] = l; v[r][1] = c4.x; v[r][2] = c4.y; v[r][3] = c4.z; v[r][4] = c4.w; v[r][5] = rr;
    }

    const float qnan = nan_f32();
    // slope / aspect: the strip's Horn sums, differences shared between its cells (terrain_cells.h)
    constexpr bool HORN = (OPS & (OP_SLOPE | OP_ASPECT)) != 0;
    const bool horn = HORN && ((a.out[0] && (OPS & OP_SLOPE)) || (a.out[1] && (OPS & OP_ASPECT)));   // wave-uniform
    HornRoller roll;
    if (horn && XRS_HORN_MODE == 1) roll.start(&v[0][0], &v[1][0]);
    const SlopeK sk = slope_constants(a.inv8cx, a.inv8cy);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const long y = y0 + r;
        if (!INTERIOR && y >= a.rows) break;
        const bool row_border = !INTERIOR && ((y - 1 < y_lo) || (y + 1 >= y_hi));
        float o_slope[4], o_aspect[4], o_curv[4], o_hill[4];
        Horn hs[4];
        if (horn && XRS_HORN_MODE == 1) roll.step(&v[r + 2][0], hs);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const long x = x0 + o;
            const bool border = !INTERIOR && (row_border || x == 0 || x == a.cols - 1);
            Nb q;
            q.nw = v[r][o];     q.n = v[r][o + 1];     q.ne = v[r][o + 2];
            q.w = v[r + 1][o];  q.c = v[r + 1][o + 1]; q.e = v[r + 1][o + 2];
            q.sw = v[r + 2][o]; q.s = v[r + 2][o + 1]; q.se = v[r + 2][o + 2];
            // (a.out[i] tests are wave-uniform: the fused instantiation skips absent products)
            if (HORN && XRS_HORN_MODE == 0) hs[o] = horn_cell(q);
            if ((OPS & OP_SLOPE) && a.out[0]) o_slope[o] = border ? qnan : slope_from_horn(hs[o], sk);
            if ((OPS & OP_ASPECT) && a.out[1]) o_aspect[o] = border ? qnan : aspect_from_horn(hs[o]);
            if ((OPS & OP_CURV) && a.out[2]) o_curv[o] = border ? qnan : curvature_cell(q, a.curv_scale);
            if ((OPS & OP_HILL) && a.out[3]) o_hill[o] = border ? qnan : hillshade_cell(q, a.sin_alt, a.cos_alt, a.cos_az, a.sin_az);
        }
        const long off = y * a.ld_out + x_tile;                         // wave-uniform
        if (INTERIOR || nown == 4) {
            if ((OPS & OP_SLOPE) && a.out[0]) store4(static_cast<float *>(a.out[0]) + off + loff, o_slope);
            if ((OPS & OP_ASPECT) && a.out[1]) store4(static_cast<float *>(a.out[1]) + off + loff, o_aspect);
            if ((OPS & OP_CURV) && a.out[2]) store4(static_cast<float *>(a.out[2]) + off + loff, o_curv);
            if ((OPS & OP_HILL) && a.out[3]) {
                if (INTERIOR && sizeof(HillT) == 8)     // float64 hillshade (the numpy path): whole-KiB store instructions
                    store_wave_row_f4_as_d(reinterpret_cast<double *>(a.out[3]) + off, lane, o_hill[0], o_hill[1], o_hill[2], o_hill[3]);
                else
                    store4(static_cast<HillT *>(a.out[3]) + off + loff, o_hill);
            }
        } else {
            if ((OPS & OP_SLOPE) && a.out[0]) store_n(static_cast<float *>(a.out[0]) + off + loff, o_slope, nown);
            if ((OPS & OP_ASPECT) && a.out[1]) store_n(static_cast<float *>(a.out[1]) + off + loff, o_aspect, nown);
            if ((OPS & OP_CURV) && a.out[2]) store_n(static_cast<float *>(a.out[2]) + off + loff, o_curv, nown);
            if ((OPS & OP_HILL) && a.out[3]) store_n(static_cast<HillT *>(a.out[3]) + off + loff, o_hill, nown);
        }
    }
}

// slope / aspect stand-alone, same-box A/B under the row-interleaved tile order (tools/ab_terrain.sh, 16384^2, four rounds,
// times relative to hillshade in the same process): Horn sums cell by cell, uncapped (85 VGPRs, 5 waves per SIMD, no
// scratch) slope 1.07 / aspect 1.12; the same capped at 5 workgroups per CU 1.08 / 1.14; differences shared along the strip
// (HornRoller: 7 instead of 10 float64 operations per cell, but 104 VGPRs) capped at 5 (96 VGPRs, 13 spilled) 1.09 / 1.16,
// uncapped (4 waves per SIMD) 1.10 / 1.20.  Occupancy beats operation count here: cell by cell it is.
#ifndef XRS_LB_HORN
#define XRS_LB_HORN 1
#endif
template <int OPS, typename HillT, int RB>
__global__ void __launch_bounds__(256, ((OPS == OP_SLOPE || OPS == OP_ASPECT) && RB == 4) ? XRS_LB_HORN : 1) terrain_strip_kernel(const TerrainArgs a) {
    const long tile = xcd_tile(blockIdx.x, a.n_tiles, a.tiles_x);
    if (tile < 0) return;
    const long ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int lane = threadIdx.x & 63;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // wave index as a scalar
    // the 4 waves of a workgroup: XRS_STRIP_WX side by side (256 columns each) x 4 / XRS_STRIP_WX stacked (RB rows each)
    constexpr int WX = XRS_STRIP_WX, WY = 4 / WX;
    const long x_tile = (tx * WX + (wy % WX)) * 256;
    const long y0 = (ty * WY + (wy / WX)) * RB;
    if (y0 >= a.rows || x_tile >= a.cols) return;
    const bool interior = x_tile >= 4 && x_tile + 256 + 4 <= a.cols &&
                          y0 - 1 >= -(long)a.halo_top && y0 + RB + 1 <= a.rows + a.halo_bot && y0 + RB <= a.rows;
    if (interior) {
        terrain_strip_body<OPS, HillT, RB, true>(a, x_tile, y0, lane);
    } else {
        if (x_tile + lane * 4 >= a.cols) return;
        terrain_strip_body<OPS, HillT, RB, false>(a, x_tile, y0, lane);
    }
}

// ------------------------------------------------------------- generic path
template <typename HillT>
__global__ void __launch_bounds__(256) terrain_cell_kernel(const TerrainArgs a, const int ops) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.rows * a.cols) return;
    const long y = idx / a.cols, x = idx - y * a.cols;
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    const bool border = (y - 1 < y_lo) || (y + 1 >= y_hi) || x == 0 || x == a.cols - 1;
    float r_slope, r_aspect, r_curv, r_hill;
    r_slope = r_aspect = r_curv = r_hill = nan_f32();
    if (!border) {
        const float *p = a.in + y * a.ld_in + x;
        Nb q;
        q.nw = p[-a.ld_in - 1]; q.n = p[-a.ld_in]; q.ne = p[-a.ld_in + 1];
        q.w = p[-1];            q.c = p[0];        q.e = p[1];
        q.sw = p[a.ld_in - 1];  q.s = p[a.ld_in];  q.se = p[a.ld_in + 1];
        if (ops & OP_SLOPE) r_slope = slope_cell(q, a.inv8cx, a.inv8cy);
        if (ops & OP_ASPECT) r_aspect = aspect_cell(q);
        if (ops & OP_CURV) r_curv = curvature_cell(q, a.curv_scale);
        if (ops & OP_HILL) r_hill = hillshade_cell(q, a.sin_alt, a.cos_alt, a.cos_az, a.sin_az);
    }
    const long off = y * a.ld_out + x;
    if ((ops & OP_SLOPE) && a.out[0]) static_cast<float *>(a.out[0])[off] = r_slope;
    if ((ops & OP_ASPECT) && a.out[1]) static_cast<float *>(a.out[1])[off] = r_aspect;
    if ((ops & OP_CURV) && a.out[2]) static_cast<float *>(a.out[2])[off] = r_curv;
    if ((ops & OP_HILL) && a.out[3]) static_cast<HillT *>(a.out[3])[off] = (HillT)r_hill;
}

template <int OPS, typename HillT, int RB>
int launch_strip_rb(TerrainArgs &a, hipStream_t s);

template <int OPS, typename HillT>
int launch_strip(TerrainArgs &a, hipStream_t s) {
    const char *e = ab_env("XRS_TERRAIN_RB");        // A/B knob: rows per wave (default 4)
    if (e && e[0] == '2') return launch_strip_rb<OPS, HillT, 2>(a, s);
    if (e && e[0] == '8') return launch_strip_rb<OPS, HillT, 8>(a, s);
    return launch_strip_rb<OPS, HillT, 4>(a, s);
}

template <int OPS, typename HillT, int RB>
int launch_strip_rb(TerrainArgs &a, hipStream_t s) {
    constexpr int WX = XRS_STRIP_WX, WY = 4 / WX;
    a.tiles_x = (a.cols + 256 * WX - 1) / (256 * WX);
    const long tiles_y = (a.rows + WY * RB - 1) / (WY * RB);
    a.n_tiles = a.tiles_x * tiles_y;
    const long grid = xcd_grid(a.n_tiles, a.tiles_x);
    if (grid > 0x7fffffffL) return fail("terrain: raster too large for one launch");
    hipLaunchKernelGGL((terrain_strip_kernel<OPS, HillT, RB>), dim3((unsigned)grid), dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

int terrain_dispatch(TerrainArgs &a, int ops, bool hill_f64, hipStream_t s) {
    if (a.rows <= 0 || a.cols <= 0) return 0;
    if (a.rows < 0 || a.cols < 0 || a.ld_in < a.cols || a.ld_out < a.cols)
        return fail("terrain: bad shape rows=%ld cols=%ld ld_in=%ld ld_out=%ld", a.rows, a.cols, a.ld_in, a.ld_out);
    if (a.halo_top < 0 || a.halo_bot < 0) return fail("terrain: negative halo");
    // the strip kernels take any width / pitch / base address (dword-aligned 16-byte accesses, ragged last lane);
    // XRS_TERRAIN_VARIANT=cell forces the one-cell-per-thread kernel (A/B, and the oracle of the ragged path's tests)
    const char *variant = ab_env("XRS_TERRAIN_VARIANT");
    const bool fast = !(variant && variant[0] == 'c');
    if (fast) {
        switch (ops) {
            case OP_SLOPE: return launch_strip<OP_SLOPE, float>(a, s);
            case OP_ASPECT: return launch_strip<OP_ASPECT, float>(a, s);
            case OP_CURV: return launch_strip<OP_CURV, float>(a, s);
            case OP_HILL: return hill_f64 ? launch_strip<OP_HILL, double>(a, s) : launch_strip<OP_HILL, float>(a, s);
            default: return launch_strip<15, float>(a, s);
        }
    }
    const long n = a.rows * a.cols;
    const long grid = (n + 255) / 256;
    if (grid > 0x7fffffffL) return fail("terrain: raster too large for the unaligned path");
    if (hill_f64)
        hipLaunchKernelGGL(terrain_cell_kernel<double>, dim3((unsigned)grid), dim3(256), 0, s, a, ops);
    else
        hipLaunchKernelGGL(terrain_cell_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s, a, ops);
    XRS_LAUNCH_CHECK();
    return 0;
}

TerrainArgs base_args(const float *in, long rows, long cols, long ld_in, long ld_out, int ht, int hb) {
    TerrainArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out;
    a.halo_top = ht; a.halo_bot = hb;
    return a;
}

void set_hillshade(TerrainArgs &a, double azimuth, double altitude) {
    hillshade_constants(azimuth, altitude, a.sin_alt, a.cos_alt, a.cos_az, a.sin_az);
}

}  // namespace

extern "C" {

int xrs_slope_f32(const float *in_dev, float *out_dev, int64_t rows, int64_t cols, int64_t ld_in,
                  int64_t ld_out, double cellsize_x, double cellsize_y, int halo_top, int halo_bot,
                  void *stream) {
    if (!in_dev || !out_dev) return fail("xrs_slope_f32: null pointer");
    TerrainArgs a = base_args(in_dev, rows, cols, ld_in, ld_out, halo_top, halo_bot);
    a.out[0] = out_dev;
    a.inv8cx = 1.0 / (8 * cellsize_x);
    a.inv8cy = 1.0 / (8 * cellsize_y);
    return terrain_dispatch(a, OP_SLOPE, false, as_stream(stream));
}

int xrs_aspect_f32(const float *in_dev, float *out_dev, int64_t rows, int64_t cols, int64_t ld_in,
                   int64_t ld_out, int halo_top, int halo_bot, void *stream) {
    if (!in_dev || !out_dev) return fail("xrs_aspect_f32: null pointer");
    TerrainArgs a = base_args(in_dev, rows, cols, ld_in, ld_out, halo_top, halo_bot);
    a.out[1] = out_dev;
    return terrain_dispatch(a, OP_ASPECT, false, as_stream(stream));
}

int xrs_curvature_f32(const float *in_dev, float *out_dev, int64_t rows, int64_t cols, int64_t ld_in,
                      int64_t ld_out, double cellsize, int halo_top, int halo_bot, void *stream) {
    if (!in_dev || !out_dev) return fail("xrs_curvature_f32: null pointer");
    TerrainArgs a = base_args(in_dev, rows, cols, ld_in, ld_out, halo_top, halo_bot);
    a.out[2] = out_dev;
    a.curv_scale = -200.0 / (cellsize * cellsize);
    return terrain_dispatch(a, OP_CURV, false, as_stream(stream));
}

int xrs_hillshade_f32(const float *in_dev, void *out_dev, int out_f64, int64_t rows, int64_t cols,
                      int64_t ld_in, int64_t ld_out, double azimuth, double angle_altitude,
                      int halo_top, int halo_bot, void *stream) {
    if (!in_dev || !out_dev) return fail("xrs_hillshade_f32: null pointer");
    TerrainArgs a = base_args(in_dev, rows, cols, ld_in, ld_out, halo_top, halo_bot);
    a.out[3] = out_dev;
    set_hillshade(a, azimuth, angle_altitude);
    return terrain_dispatch(a, OP_HILL, out_f64 != 0, as_stream(stream));
}

int xrs_terrain_fused_f32(const float *in_dev, float *slope_dev, float *aspect_dev, float *curvature_dev,
                          float *hillshade_dev, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out,
                          double cellsize_x, double cellsize_y, double azimuth, double angle_altitude,
                          int halo_top, int halo_bot, void *stream) {
    if (!in_dev) return fail("xrs_terrain_fused_f32: null input");
    TerrainArgs a = base_args(in_dev, rows, cols, ld_in, ld_out, halo_top, halo_bot);
    a.out[0] = slope_dev; a.out[1] = aspect_dev; a.out[2] = curvature_dev; a.out[3] = hillshade_dev;
    a.inv8cx = 1.0 / (8 * cellsize_x);
    a.inv8cy = 1.0 / (8 * cellsize_y);
    const double cs = (cellsize_x + cellsize_y) / 2;       // curvature.py:241
    a.curv_scale = -200.0 / (cs * cs);
    set_hillshade(a, azimuth, angle_altitude);
    int ops = 0;
    if (slope_dev) ops |= OP_SLOPE;
    if (aspect_dev) ops |= OP_ASPECT;
    if (curvature_dev) ops |= OP_CURV;
    if (hillshade_dev) ops |= OP_HILL;
    if (!ops) return 0;
    // single products go through their specialised kernel; any combination uses the fused one
    const int single = (ops & (ops - 1)) == 0 ? ops : 15;
    return terrain_dispatch(a, single, false, as_stream(stream));
}

}  // extern "C"
