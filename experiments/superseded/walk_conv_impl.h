// convolve_2d with one weight value on a circle / box through the column walker of circle_walk.h, for one mask shape
// (XRS_WALK_SHAPE) and radius 3..12 cells.  Included by kxk_circle_conv.hip and kxk_box_conv.hip.
#include "circle_walk.h"

using namespace xrs;

namespace {

template <int R>
__global__ void __launch_bounds__(256) XRS_WALK_KERNEL(const WalkGeom g, float *out, double w, const double *weights) {
    walk_conv_tile<R, XRS_WALK_SHAPE>(g, out, w, weights);
}

template <int R>
int launch_conv(WalkGeom &g, float *out, const double *kernel, const double *weights_dev, hipStream_t s) {
    double w;
    if (!is_uniform_shape<R, XRS_WALK_SHAPE>(kernel, &w)) return -1;
    long grid;
    if (int rc = walk_grid(g, &grid)) return rc;
    hipLaunchKernelGGL((XRS_WALK_KERNEL<R>), dim3((unsigned)grid), dim3(256), 0, s, g, out, w, weights_dev);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace xrs {

// 0 = launched, -1 = not one weight on this shape / radius (caller uses the tap kernels), > 0 = error.
// `weights_dev`: the kernel as float64 in device memory (already uploaded by the caller), for the rare exact path.
int XRS_WALK_ENTRY(const float *in, float *out, long rows, long cols, long ld_in, long ld_out, const double *kernel,
                   const double *weights_dev, int krows, int kcols, int halo_top, int halo_bot, hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    WalkGeom g;
    memset(&g, 0, sizeof(g));
    g.in = in; g.rows = rows; g.cols = cols; g.ld_in = ld_in; g.ld_out = ld_out;
    g.halo_top = halo_top; g.halo_bot = halo_bot;
    switch (krows / 2) {
        case 3: return launch_conv<3>(g, out, kernel, weights_dev, s);
        case 4: return launch_conv<4>(g, out, kernel, weights_dev, s);
        case 5: return launch_conv<5>(g, out, kernel, weights_dev, s);
        case 6: return launch_conv<6>(g, out, kernel, weights_dev, s);
        case 7: return launch_conv<7>(g, out, kernel, weights_dev, s);
        case 8: return launch_conv<8>(g, out, kernel, weights_dev, s);
        case 9: return launch_conv<9>(g, out, kernel, weights_dev, s);
        case 10: return launch_conv<10>(g, out, kernel, weights_dev, s);
        case 11: return launch_conv<11>(g, out, kernel, weights_dev, s);
        case 12: return launch_conv<12>(g, out, kernel, weights_dev, s);
        default: return -1;
    }
}

}  // namespace xrs
