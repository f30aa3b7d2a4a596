// focal.apply with a user callable: the windows themselves, gathered on the device.
//
// Reference: _apply_numpy (xrspatial/focal.py:305-326) fills a kernel-shaped float32 array per cell -- NaN everywhere,
// then data[ky, kx] where the kernel is 1 and (ky, kx) is inside the raster -- and hands it to `func`.  With the seven
// built-in reducers that loop is xrs_focal_stats_f32; an arbitrary Python callable can only run on the host, so this
// entry materialises exactly those arrays for a band of rows, [band_rows][cols][krows][kcols] float32, and the host
// calls `func` on each.  HBM-bound on the write (4 * krows * kcols bytes per cell; the taps re-read the band from L2).
#include "xrs_common.h"

using namespace xrs;

namespace {

constexpr int MASK_WORDS = 64;                  // up to 2048 taps (45 x 45)
struct TapMask { unsigned w[MASK_WORDS]; };

__global__ void __launch_bounds__(256) windows_kernel(const float *in, float *out, long rows, long cols, long ld, long y0,
                                                      long n_out, int kr, int kc, TapMask m) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const int ntaps = kr * kc;
    const long cell = i / ntaps;
    const int t = (int)(i - cell * ntaps);
    const long yl = cell / cols, x = cell - yl * cols;
    const int ky = t / kc, kx = t - ky * kc;
    const long sy = y0 + yl + ky - kr / 2, sx = x + kx - kc / 2;
    const bool on = (m.w[t >> 5] >> (t & 31)) & 1u;
    float v = nan_f32();
    if (on && sy >= 0 && sy < rows && sx >= 0 && sx < cols) v = in[sy * ld + sx];
    __builtin_nontemporal_store(v, out + i);
}

}  // namespace

extern "C" int xrs_focal_windows_f32(const float *in_dev, float *windows_dev, int64_t rows, int64_t cols, int64_t ld_in,
                                     int64_t y0, int64_t band_rows, const double *kernel, int krows, int kcols,
                                     void *stream) {
    if (rows < 0 || cols < 0 || band_rows < 0 || y0 < 0 || y0 + band_rows > rows)
        return fail("xrs_focal_windows_f32: band [%ld, %ld) outside the %ld rows", (long)y0, (long)(y0 + band_rows), (long)rows);
    if (krows < 1 || kcols < 1 || !(krows & 1) || !(kcols & 1)) return fail("xrs_focal_windows_f32: kernel sides must be odd");
    if ((long)krows * kcols > 32L * MASK_WORDS)
        return fail("xrs_focal_windows_f32: at most %d taps (got %d x %d)", 32 * MASK_WORDS, krows, kcols);
    if (band_rows == 0 || cols == 0) return 0;
    if (!in_dev || !windows_dev || !kernel) return fail("xrs_focal_windows_f32: null pointer");
    if (ld_in < cols) return fail("xrs_focal_windows_f32: ld_in < cols");
    TapMask m;
    memset(&m, 0, sizeof m);
    for (int t = 0; t < krows * kcols; ++t)
        if (kernel[t] == 1.0) m.w[t >> 5] |= 1u << (t & 31);          // `kernel[kyidx, kxidx] == 1` (focal.py:323)
    const long n_out = band_rows * cols * krows * kcols;
    const long blocks = (n_out + 255) / 256;
    if (blocks >= (1L << 31)) return fail("xrs_focal_windows_f32: band too large (%ld values); use fewer rows per call", n_out);
    hipLaunchKernelGGL(windows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), in_dev, windows_dev, rows, cols,
                       ld_in, y0, n_out, krows, kcols, m);
    XRS_LAUNCH_CHECK();
    return 0;
}
