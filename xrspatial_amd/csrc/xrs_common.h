// Shared helpers for libxrs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/xrs_hip.h"

namespace xrs {

// Per-thread last-error text: the only mutable state in the library.
inline char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return 1;
}

#define XRS_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess)                                                          \
            return ::xrs::fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                               __FILE__, __LINE__);                                    \
    } while (0)

#define XRS_LAUNCH_CHECK()                                                             \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess)                                                          \
            return ::xrs::fail("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                    \
    } while (0)

// A/B switches (XRS_FOCAL_GEN, XRS_CONV_GEN, XRS_RIM_FIRST ...) exist only in `make AB=1` builds: the default library
// reads no environment variable on a launch path -- ab_env() is a constant there and every branch behind it folds away.
#ifdef XRS_AB
inline const char *ab_env(const char *name) { return getenv(name); }
#else
constexpr const char *ab_env(const char *) { return nullptr; }
#endif

// Run-time options of the library (xrs_set_option / xrs_get_option, include/xrs_hip.h): process-wide, relaxed atomics.
int option_value(int key);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// blockIdx -> tile index.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8, observed, used for
// speed only).  Tiles are numbered row-major and handed out in groups of `unit` consecutive tiles -- one tile ROW for
// the 2-D kernels (unit = tiles_x): XCD k takes tile rows k, k + 8, ...  All eight XCDs then walk through the same
// few rows of the raster at the same time (one dense stream through DRAM), and the halo COLUMNS a tile re-reads were
// fetched into its own L2 by its left / right neighbours; the halo ROWS between tile rows are fetched by two L2s
// (2 rows in 16 for the 3x3 strip kernels: +6 % traffic at the L2 <-> fabric counters).  Round 1 gave every XCD one
// contiguous band of the raster instead (all halos in one L2, traffic == algorithmic bytes) -- eight streams 1/8 of the
// raster apart, and 4-7 % slower on every strip kernel (experiments/strip_floor.hip: 0.352 vs 0.328 ms for the loading
// pattern of the 3x3 kernels; groups of 2 / 4 / 8+ tile rows: 0.332 / 0.340 / 0.35).  unit = 1: launch order.
// Returns -1 for the surplus blocks of the padded grid.
// unit = 0: the round-1 order, one contiguous band of tiles per XCD -- kept for the column walkers (circle_walk.h,
// wide_impl.h: tall tiles, 100+ rows each; same-box A/B 3-6 % faster in bands) and, unmeasured, the
// LDS-tile kernels.
__device__ __forceinline__ long xcd_tile(long block, long n_tiles, long unit) {
    if (unit == 0) {
        const long per_xcd = (n_tiles + 7) >> 3;
        const long t = (block & 7) * per_xcd + (block >> 3);
        return ((block >> 3) < per_xcd && t < n_tiles) ? t : -1;
    }
    const unsigned xcd = (unsigned)block & 7u, j = (unsigned)(block >> 3), u = (unsigned)unit;
    const unsigned grp = j / u, within = j - grp * u;
    const long t = ((long)grp * 8 + xcd) * unit + within;
    return t < n_tiles ? t : -1;
}
inline long xcd_grid(long n_tiles, long unit) {
    if (unit == 0) return ((n_tiles + 7) >> 3) << 3;
    const long groups = (n_tiles + unit - 1) / unit;
    return (((groups + 7) >> 3) << 3) * unit;
}
// which order a kernel family uses (compile-time, the same expression on the host and in the kernel)
#ifndef XRS_XCD_WALK
#define XRS_XCD_WALK 0          // column walkers: bands
#endif
#ifndef XRS_XCD_LDS
#define XRS_XCD_LDS 0           // LDS-tile window kernels, run kernels, geodesic: bands (as measured in round 1)
#endif
#define XCD_UNIT(rows, tiles_x) ((rows) ? (long)(tiles_x) : 0L)

__device__ __forceinline__ float nan_f32() { return __int_as_float(0x7fc00000); }

// max(a, |b|, |c|) in one v_max3_f32 (no canonicalising moves; a NaN operand is skipped, which is what the guard below
// wants: NaN cells are caught through the sums they poison)
__device__ __forceinline__ float amax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}


// 16-byte global accesses at DWORD alignment (global_load/store_dwordx4 only need 4-byte alignment on gfx9+): rasters
// whose width or pitch is not a multiple of 4 cells -- an SRTM tile is 3601 wide -- keep the one-instruction-per-lane
// row accesses of the strip kernels instead of dropping to one-cell-per-thread kernels.
struct __attribute__((packed, aligned(4))) xrs_f4u { float x, y, z, w; };
struct __attribute__((packed, aligned(8))) xrs_d2u { double x, y; };
__device__ __forceinline__ xrs_f4u load_f4u(const float *p) { return *reinterpret_cast<const xrs_f4u *>(p); }

// Streaming ("nt") cache policy for planes that are read or written exactly once per launch (copy, per-cell indices,
// zonal reductions).  On a 1 GiB + 1 GiB copy on this chip nt loads + nt stores measured 6 290 GB/s against 5 870 with
// the default policy (experiments/copy_sweep.hip); ndvi / evi / savi gain 4-5 %.  NOT for the stencil kernels' input:
// their halo rows and columns are re-read by neighbouring strips out of L2, and streaming loads cost them 4-35 %
// (profiles/r01/nt_ab_r01.log).  XRS_NT=0 builds the default-policy library for A/B runs.
#ifndef XRS_NT
#define XRS_NT 1
#endif
typedef float xrs_v4f __attribute__((ext_vector_type(4)));
typedef float xrs_v4fu __attribute__((ext_vector_type(4), aligned(4)));
typedef int xrs_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg_stream(const float4 *p) {
#if XRS_NT
    const xrs_v4f v = __builtin_nontemporal_load(reinterpret_cast<const xrs_v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ int4 ldg_stream(const int4 *p) {
#if XRS_NT
    const xrs_v4i v = __builtin_nontemporal_load(reinterpret_cast<const xrs_v4i *>(p));
    return make_int4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void stg_stream(float4 *p, const float4 v) {
#if XRS_NT
    xrs_v4f q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
    __builtin_nontemporal_store(q, reinterpret_cast<xrs_v4f *>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ xrs_f4u load_f4u_stream(const float *p) {       // dword-aligned 16 bytes
#if XRS_NT
    const xrs_v4fu v = __builtin_nontemporal_load(reinterpret_cast<const xrs_v4fu *>(p));
    xrs_f4u r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
#else
    return load_f4u(p);
#endif
}
// Results are written once and not read again by the launch that produces them: streaming ("nt") stores.  Same-box A/B
// (profiles/r01/ntst_ab_r01.log): hillshade 0.386 -> 0.366 ms, slope 0.422 -> 0.400, curvature 0.388 -> 0.361,
// 5x5 convolve 0.556 -> 0.438, four terrain products 1.147 -> 1.070.  XRS_NT_STENCIL_STORES=0: default policy.
#ifndef XRS_NT_STENCIL_STORES
#define XRS_NT_STENCIL_STORES 1
#endif
typedef double xrs_v2du_st __attribute__((ext_vector_type(2), aligned(8)));
template <typename T>
__device__ __forceinline__ void st_stream(T *p, const T v) {          // one scalar result per lane
#if XRS_NT_STENCIL_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void store_d2u(double *p, double x, double y) {    // two doubles at 8-byte alignment
#if XRS_NT_STENCIL_STORES
    xrs_v2du_st q; q.x = x; q.y = y;
    __builtin_nontemporal_store(q, reinterpret_cast<xrs_v2du_st *>(p));
#else
    xrs_d2u q; q.x = x; q.y = y;
    *reinterpret_cast<xrs_d2u *>(p) = q;
#endif
}
// A wave's 256 adjacent float64 results (4 per lane, lane l owns columns 4l .. 4l+3) as TWO store instructions that
// each write 1 KiB of consecutive bytes: lane pairs are transposed through ds_bpermute first.  Every lane storing its own
// 32 bytes as two 16-byte halves makes each store instruction touch 64 x 16 B at a 32-byte stride -- two instructions
// writing alternate halves of the same 2 KiB -- and costs a third of the bandwidth (experiments/f64_store.hip: 0.856 ->
// 0.584 ms for a float32 -> float64 plane).  `row` = address of the wave's first result; ALL 64 lanes must be active
// (wave-uniform control flow).
__device__ __forceinline__ void store_wave_row_d4(double *row, int lane, double x, double y, double z, double w) {
    const bool odd = lane & 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        // store k writes doubles [128 k + 2 lane, + 2): columns (2 lane, 2 lane + 1) of half k, owned by lane 32 k + lane / 2
        const int src = 32 * k + (lane >> 1);
        const double a0 = __shfl(x, src), a1 = __shfl(y, src), a2 = __shfl(z, src), a3 = __shfl(w, src);
        store_d2u(row + 128 * k + 2 * lane, odd ? a2 : a0, odd ? a3 : a1);
    }
}
// the same for float32 results that leave as float64 (converted after the exchange: half the shuffles)
__device__ __forceinline__ void store_wave_row_f4_as_d(double *row, int lane, float x, float y, float z, float w) {
    const bool odd = lane & 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int src = 32 * k + (lane >> 1);
        const float a0 = __shfl(x, src), a1 = __shfl(y, src), a2 = __shfl(z, src), a3 = __shfl(w, src);
        store_d2u(row + 128 * k + 2 * lane, (double)(odd ? a2 : a0), (double)(odd ? a3 : a1));
    }
}
typedef float xrs_v4fu_st __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void store_f4u(float *p, float x, float y, float z, float w) {
#if XRS_NT_STENCIL_STORES
    xrs_v4fu_st q; q.x = x; q.y = y; q.z = z; q.w = w;
    __builtin_nontemporal_store(q, reinterpret_cast<xrs_v4fu_st *>(p));
#else
    xrs_f4u q; q.x = x; q.y = y; q.z = z; q.w = w;
    *reinterpret_cast<xrs_f4u *>(p) = q;
#endif
}

// kxk_runs.hip: prefix-sum focal mean for large run-structured masks.  0 = launched, -1 = mask not
// suitable (caller uses the tap kernels), > 0 = error.
int try_launch_focal_mean_runs(const float *in, float *out, long rows, long cols, long ld_in, long ld_out,
                               const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                               hipStream_t s);
// same tiles, plus variance / standard deviation from a second prefix array (any output may be null)
int try_launch_focal_meanvar_runs(const float *in, float *out_mean, float *out_var, float *out_std, long rows,
                                  long cols, long ld_in, long ld_out, const double *kernel, int krows, int kcols,
                                  int halo_top, int halo_bot, hipStream_t s);

// kxk_circle.hip: float32 sum / max / min / range for circular masks of radius 2..12 cells (column walker).
// 0 = launched, -1 = not such a circle (caller walks the taps), > 0 = error.  Null outputs are skipped.
// With any of out_mean / out_var / out_std non-null all seven statistics come from one kernel (radius 2, 3 only).
int try_launch_focal_circle_f32(const float *in, float *out_sum, float *out_max, float *out_min, float *out_range,
                                float *out_mean, float *out_var, float *out_std, long rows, long cols, long ld_in,
                                long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                                hipStream_t s);

int try_launch_focal_box_f32(const float *in, float *out_sum, float *out_max, float *out_min, float *out_range,
                             float *out_mean, float *out_var, float *out_std, long rows, long cols, long ld_in,
                             long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                             hipStream_t s);      // kxk_box.hip: the same for np.ones((k, k)) masks
// kxk_circle64.hip / kxk_box64.hip: mean / var / std for the same shapes (float64 moments of shifted values, guarded).
int try_launch_focal_box_f64(const float *in, float *out_mean, float *out_var, float *out_std, long rows, long cols,
                             long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top,
                             int halo_bot, hipStream_t s);
int try_launch_focal_circle_f64(const float *in, float *out_mean, float *out_var, float *out_std, long rows, long cols,
                                long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top,
                                int halo_bot, hipStream_t s);

// kxk_wide_circle.hip / kxk_wide_box.hip: focal mean / window sum, radius 3..12 cells, float32 on shifted values with a
// guarded fall-back to the float64 walker (wide_impl.h).  0 = launched, -1 = not such a mask, > 0 = error.
int try_launch_focal_wide_circle(const float *in, float *out_mean, float *out_sum, long rows, long cols, long ld_in,
                                 long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                                 hipStream_t s);
int try_launch_focal_wide_box(const float *in, float *out_mean, float *out_sum, long rows, long cols, long ld_in,
                              long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                              hipStream_t s);
// the same TUs: convolve_2d with one weight value on a circle / box of radius 3..12 cells (WIDE_CONV)
int try_launch_conv_wide_circle(const float *in, float *out, long rows, long cols, long ld_in, long ld_out, const double *kernel,
                                const double *weights_dev, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s);
int try_launch_conv_wide_box(const float *in, float *out, long rows, long cols, long ld_in, long ld_out, const double *kernel,
                             const double *weights_dev, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s);
// kxk_big.hip: focal statistics / convolve_2d for windows beyond the tiled kernels' 63 x 63 (one thread per cell, window from
// global memory, mask / weights from a device copy of the kernel in `work_dev`)
int launch_window_any_size(bool conv, const float *in, float *const *outs, long rows, long cols, long ld_in, long ld_out,
                           const double *kernel, int krows, int kcols, void *work_dev, int halo_top, int halo_bot,
                           hipStream_t s);
// pass.hip: is this mask one of the compile-time masks of raster_pass_kernel (circle_kernel(1, 1, 2), np.ones((3, 3)))?
bool pass_has_compile_time_mask(const double *kernel, int krows, int kcols);
// kxk_ext_circle.hip / kxk_ext_box.hip (ext_impl.h): max / min / range, radius 4..12 cells, two input rows per step.
int try_launch_focal_ext_circle(const float *in, float *out_max, float *out_min, float *out_range, long rows, long cols,
                                long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top,
                                int halo_bot, hipStream_t s);
int try_launch_focal_ext_box(const float *in, float *out_max, float *out_min, float *out_range, long rows, long cols,
                             long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                             hipStream_t s);
// kxk_mom_circle.hip / kxk_mom_box.hip (mom_impl.h): mean / var / std / sum, radius 4..12 cells, float32 sums about a
// shift that trails the walk, guarded; exact float64 walker for the tiles that fail the guard.
int try_launch_focal_mom_circle(const float *in, float *out_sum, float *out_mean, float *out_var, float *out_std, long rows,
                                long cols, long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top,
                                int halo_bot, hipStream_t s, unsigned char *todo_dev = nullptr);
int try_launch_focal_mom_box(const float *in, float *out_sum, float *out_mean, float *out_var, float *out_std, long rows,
                             long cols, long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top,
                             int halo_bot, hipStream_t s, unsigned char *todo_dev = nullptr);
// kxk_ext_ann_*.hip / kxk_mom_ann*.hip: the same two walkers for annulus_kernel(1, 1, R, RI), 4 <= R <= 12, 1 <= RI < R (one
// instantiation per pair; the moments walkers one translation unit per outer radius).  0 = launched, -1 = not such an annulus.
int try_launch_focal_ext_annulus_a(const float *in, float *out_max, float *out_min, float *out_range, long rows, long cols,
                                   long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                                   hipStream_t s);
int try_launch_focal_ext_annulus_b(const float *in, float *out_max, float *out_min, float *out_range, long rows, long cols,
                                   long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                                   hipStream_t s);
int try_launch_focal_ext_annulus_c(const float *in, float *out_max, float *out_min, float *out_range, long rows, long cols,
                                   long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                                   hipStream_t s);
#define XRS_DECL_MOM_ANNULUS(RR)                                                                                              \
    int try_launch_focal_mom_annulus##RR(const float *in, float *out_sum, float *out_mean, float *out_var, float *out_std,   \
                                         long rows, long cols, long ld_in, long ld_out, const double *kernel, int krows,     \
                                         int kcols, int halo_top, int halo_bot, hipStream_t s);
XRS_DECL_MOM_ANNULUS(4) XRS_DECL_MOM_ANNULUS(5) XRS_DECL_MOM_ANNULUS(6) XRS_DECL_MOM_ANNULUS(7) XRS_DECL_MOM_ANNULUS(8)
XRS_DECL_MOM_ANNULUS(9) XRS_DECL_MOM_ANNULUS(10) XRS_DECL_MOM_ANNULUS(11) XRS_DECL_MOM_ANNULUS(12)
#undef XRS_DECL_MOM_ANNULUS
// kxk_wide_ann4.hip .. kxk_wide_ann12.hip (wide_impl.h): the mean (out_mean) or the uniform-weight convolution (out_conv; exactly
// one of the two) over annulus_kernel(1, 1, RR, RI).  0 = launched, -1 = not such a mask, > 0 = error.
#define XRS_DECL_WIDE_ANNULUS(RR)                                                                                             \
    int try_launch_wide_annulus##RR(const float *in, float *out_mean, float *out_conv, long rows, long cols, long ld_in,      \
                                    long ld_out, const double *kernel, const double *weights_dev, int krows, int kcols,       \
                                    int halo_top, int halo_bot, hipStream_t s);
XRS_DECL_WIDE_ANNULUS(4) XRS_DECL_WIDE_ANNULUS(5) XRS_DECL_WIDE_ANNULUS(6) XRS_DECL_WIDE_ANNULUS(7) XRS_DECL_WIDE_ANNULUS(8)
XRS_DECL_WIDE_ANNULUS(9) XRS_DECL_WIDE_ANNULUS(10) XRS_DECL_WIDE_ANNULUS(11) XRS_DECL_WIDE_ANNULUS(12)
#undef XRS_DECL_WIDE_ANNULUS
// kxk_sw_circle.hip / kxk_sw_box.hip (sw_impl.h): any of the seven statistics over circles / boxes of radius 2, 3 cells from one
// pass of the strip walker (outs: XRS_STAT_* order, NULL = not wanted).  0 = launched, -1 = not such a mask, > 0 = error.
int try_launch_focal_sw_circle(const float *in, float *const *outs, long rows, long cols, long ld_in, long ld_out,
                               const double *kernel, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s);
int try_launch_focal_sw_box(const float *in, float *const *outs, long rows, long cols, long ld_in, long ld_out,
                            const double *kernel, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s);
// boxsep.hip: the separable fast walk for np.ones((krows, kcols)) -- mean / var / std / sum sets WITH var or std (for the
// mean or the sum alone it measured no faster than the wide row walker: 0.53 vs 0.55 ms at 25x25, 0.68 vs 0.47 at 11x11) --
// in front of a fall-back kernel whose workgroup tiles (fb_group_cols x fb_tile_rows cells, fb_groups_x per tile row) it
// marks in `todo` where it cannot stand for its results.  0 = launched, -1 = not for this walk (the caller runs its kernel
// everywhere), > 0 = error.
int launch_box_sep(const float *in, float *out_sum, float *out_mean, float *out_var, float *out_std,
                   long rows, long cols, long ld_in, long ld_out, int krows, int kcols, int halo_top, int halo_bot,
                   unsigned char *todo, long fb_groups_x, int fb_tile_rows, int fb_group_cols, hipStream_t s);
// bytes of `todo` that is always enough for a rows x cols raster (host side of xrs_focal_workspace_bytes)
inline size_t box_todo_bytes(long rows, long cols) { return (size_t)(cols / 512 + 2) * (size_t)(rows / 64 + 2); }
// the moments kernels' work-list of wave tiles handed on to focal_mom_rescue_kernel (mom_impl.h): [0] count, [2..] tiles;
// wave tiles are at least 64 columns x 16 rows
// behind it (mom_exact_offset): the bands the rescue launch hands on to the float64 walker's own launch, 16 bytes each
inline size_t mom_rescue_tiles(long rows, long cols) { return (size_t)(cols / 64 + 2) * (size_t)(rows / 16 + 2); }
inline size_t mom_exact_offset(long rows, long cols) { return 256 + 4 * mom_rescue_tiles(rows, cols); }
inline size_t mom_exact_cap(long rows, long cols) { return 2 * mom_rescue_tiles(rows, cols) + 8192; }
inline size_t mom_rescue_bytes(long rows, long cols) { return mom_exact_offset(rows, cols) + 16 + 16 * mom_exact_cap(rows, cols); }
// ... handed from xrs_focal_stats_f32 (kxk.hip, which owns the caller's workspace) to launch_mom (mom_impl.h, nine
// translation units) without widening every entry point in between: set around the call, NULL otherwise
inline unsigned *&mom_rescue_slot() {
    static thread_local unsigned *p = nullptr;
    return p;
}

}  // namespace xrs
