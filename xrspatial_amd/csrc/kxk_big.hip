// Focal statistics and convolve_2d for windows of ANY size and shape (round 3: nothing is refused any more).
//
// The tiled kernels of kxk.hip stage a (tile + window) region in LDS and carry the mask as one 64-bit word per kernel row:
// they end at 63 x 63.  The reference takes any odd-shaped kernel (xrspatial/convolution.py:285-313, focal.py:305-326), at
// O(window) work per cell; so does this fall-back: one thread per output cell, the window read straight from global
// memory (consecutive lanes read consecutive cells: coalesced, and neighbouring windows overlap in L1 / L2), the mask or
// the weights read from a float64 copy of the kernel in device memory (wave-uniform addresses: scalar loads).  Same
// arithmetic as the tiled kernels: taps in row-major order, float64 sum / count for the mean, two-pass float64 variance,
// float32 sequential sum (numba's nansum keeps the array dtype), NaN cells skipped, window clipped at the raster / shard
// edge; convolve_2d: float64 multiply-adds over the full window, NaN within k//2 cells of the edge.
// A 101 x 101 window costs 10 201 loads per cell and statistic pass -- minutes on a 16384^2 raster, as in the reference;
// circles, boxes and everything up to 63 x 63 never come here.
#include "xrs_common.h"

using namespace xrs;

namespace {

struct BigArgs {
    const float *in;
    float *out[XRS_NUM_STATS];    // focal: per statistic; convolve: out[0]
    long rows, cols, ld_in, ld_out;
    int halo_top, halo_bot, krows, kcols;
    const double *kern;           // device copy of the kernel: taps are the cells == 1.0 (focal) / weights (convolve)
};

__global__ void __launch_bounds__(256) focal_big_kernel(const BigArgs a) {
    const long x = (long)blockIdx.x * 256 + threadIdx.x;
    const long y = blockIdx.y;
    if (x >= a.cols) return;
    const int ry = a.krows / 2, rx = a.kcols / 2;
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    const bool want_var = a.out[XRS_STAT_VAR] || a.out[XRS_STAT_STD];
    double sum64 = 0.0;
    float sum32 = 0.0f, mn = INFINITY, mx = -INFINITY;
    int cnt = 0;
    for (int ky = 0; ky < a.krows; ++ky) {
        const long yy = y + ky - ry;
        if (yy < y_lo || yy >= y_hi) continue;                   // (block-uniform)
        const float *row = a.in + yy * a.ld_in;
        const double *krow = a.kern + (long)ky * a.kcols;
        for (int kx = 0; kx < a.kcols; ++kx) {
            if (krow[kx] != 1.0) continue;                       // (wave-uniform: `kernel == 1` selects a tap, focal.py:323)
            const long xx = x + kx - rx;
            if (xx < 0 || xx >= a.cols) continue;
            const float v = row[xx];
            if (isnan(v)) continue;
            sum64 += (double)v;
            sum32 += v;
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
            ++cnt;
        }
    }
    const double mean = sum64 / (double)cnt;                     // (0 / 0 = NaN for a window without a valid cell)
    const long off = y * a.ld_out + x;
    if (a.out[XRS_STAT_MEAN]) a.out[XRS_STAT_MEAN][off] = (float)mean;
    if (a.out[XRS_STAT_MAX]) a.out[XRS_STAT_MAX][off] = cnt ? mx : nan_f32();
    if (a.out[XRS_STAT_MIN]) a.out[XRS_STAT_MIN][off] = cnt ? mn : nan_f32();
    if (a.out[XRS_STAT_RANGE]) a.out[XRS_STAT_RANGE][off] = cnt ? mx - mn : nan_f32();
    if (a.out[XRS_STAT_SUM]) a.out[XRS_STAT_SUM][off] = sum32;
    if (!want_var) return;
    double ssd = 0.0;
    for (int ky = 0; ky < a.krows; ++ky) {
        const long yy = y + ky - ry;
        if (yy < y_lo || yy >= y_hi) continue;
        const float *row = a.in + yy * a.ld_in;
        const double *krow = a.kern + (long)ky * a.kcols;
        for (int kx = 0; kx < a.kcols; ++kx) {
            if (krow[kx] != 1.0) continue;
            const long xx = x + kx - rx;
            if (xx < 0 || xx >= a.cols) continue;
            const float v = row[xx];
            if (isnan(v)) continue;
            const double d = (double)v - mean;
            ssd += d * d;
        }
    }
    const double var = ssd / (double)cnt;
    if (a.out[XRS_STAT_VAR]) a.out[XRS_STAT_VAR][off] = (float)var;
    if (a.out[XRS_STAT_STD]) a.out[XRS_STAT_STD][off] = (float)sqrt(var);
}

__global__ void __launch_bounds__(256) convolve_big_kernel(const BigArgs a) {
    const long x = (long)blockIdx.x * 256 + threadIdx.x;
    const long y = blockIdx.y;
    if (x >= a.cols) return;
    const int ry = a.krows / 2, rx = a.kcols / 2;
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    float res = nan_f32();
    // convolution.py:296-309: only cells whose whole window lies inside the raster are computed, the border stays NaN
    if (y - ry >= y_lo && y + ry < y_hi && x - rx >= 0 && x + rx < a.cols) {
#pragma clang fp contract(off)       // num += kernel * data: a multiply and an add, as numba emits them
        double acc = 0.0;
        for (int ky = 0; ky < a.krows; ++ky) {
            const float *row = a.in + (y + ky - ry) * a.ld_in + (x - rx);
            const double *krow = a.kern + (long)ky * a.kcols;
            for (int kx = 0; kx < a.kcols; ++kx) acc += krow[kx] * (double)row[kx];
        }
        res = (float)acc;
    }
    a.out[0][y * a.ld_out + x] = res;
}

int launch_big(const BigArgs &a, bool conv, hipStream_t s) {
    if (a.rows > 0x7fffffffL) return fail("window kernels: too many rows for one launch");
    const dim3 grid((unsigned)((a.cols + 255) / 256), (unsigned)a.rows);
    if (conv) hipLaunchKernelGGL(convolve_big_kernel, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(focal_big_kernel, grid, dim3(256), 0, s, a);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace xrs {

// `kernel`: host float64 (krows x kcols); `work_dev`: device workspace of krows * kcols * 8 bytes (xrs_kxk_workspace_bytes)
// that receives its copy.  focal: `outs` per statistic (null = not wanted); convolve: outs[0].
int launch_window_any_size(bool conv, const float *in, float *const *outs, long rows, long cols, long ld_in, long ld_out,
                           const double *kernel, int krows, int kcols, void *work_dev, int halo_top, int halo_bot,
                           hipStream_t s) {
    if (!work_dev)
        return fail("a %dx%d window needs a device workspace of xrs_kxk_workspace_bytes(%d, %d) bytes", krows, kcols, krows, kcols);
    XRS_HIP(hipMemcpyAsync(work_dev, kernel, (size_t)krows * kcols * sizeof(double), hipMemcpyHostToDevice, s));
    BigArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in;
    for (int i = 0; i < (conv ? 1 : XRS_NUM_STATS); ++i) a.out[i] = outs[i];
    a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out;
    a.halo_top = halo_top; a.halo_bot = halo_bot; a.krows = krows; a.kcols = kcols;
    a.kern = static_cast<const double *>(work_dev);
    return launch_big(a, conv, s);
}

}  // namespace xrs
