// Float64 statistics (mean, var, std) of focal_stats / focal.apply through the column walker of circle_walk.h, for
// one mask shape (XRS_WALK_SHAPE) and radius 1..12 cells.  Included by kxk_circle64.hip and kxk_box64.hip.
#include "circle_walk.h"

using namespace xrs;

namespace {

template <int R, bool WANT_VAR>
__global__ void __launch_bounds__(256) XRS_WALK_KERNEL(const WalkGeom g, const WalkOuts o) {
    walk_tile<R, XRS_WALK_SHAPE, false, false, false, true, WANT_VAR>(g, o);
}

template <int R>
int launch64(WalkGeom &g, const WalkOuts &o, const double *kernel, hipStream_t s) {
    if (!is_shape<R, XRS_WALK_SHAPE>(kernel)) return -1;
    long grid;
    if (int rc = walk_grid(g, &grid)) return rc;
    if (o.var || o.std)
        hipLaunchKernelGGL((XRS_WALK_KERNEL<R, true>), dim3((unsigned)grid), dim3(256), 0, s, g, o);
    else
        hipLaunchKernelGGL((XRS_WALK_KERNEL<R, false>), dim3((unsigned)grid), dim3(256), 0, s, g, o);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

namespace xrs {

// 0 = launched, -1 = not a circle this file is instantiated for, > 0 = error
int XRS_WALK_ENTRY(const float *in, float *out_mean, float *out_var, float *out_std, long rows, long cols,
                                long ld_in, long ld_out, const double *kernel, int krows, int kcols, int halo_top,
                                int halo_bot, hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    if (!out_mean && !out_var && !out_std) return 0;
    WalkGeom g;
    memset(&g, 0, sizeof(g));
    g.in = in; g.rows = rows; g.cols = cols; g.ld_in = ld_in; g.ld_out = ld_out;
    g.halo_top = halo_top; g.halo_bot = halo_bot;
    const WalkOuts o = {nullptr, nullptr, nullptr, nullptr, out_mean, out_var, out_std};
    switch (krows / 2) {
        case 1: return launch64<1>(g, o, kernel, s);
        case 2: return launch64<2>(g, o, kernel, s);
        case 3: return launch64<3>(g, o, kernel, s);
        case 4: return launch64<4>(g, o, kernel, s);
        case 5: return launch64<5>(g, o, kernel, s);
        case 6: return launch64<6>(g, o, kernel, s);
        case 7: return launch64<7>(g, o, kernel, s);
        case 8: return launch64<8>(g, o, kernel, s);
        case 9: return launch64<9>(g, o, kernel, s);
        case 10: return launch64<10>(g, o, kernel, s);
        case 11: return launch64<11>(g, o, kernel, s);
        case 12: return launch64<12>(g, o, kernel, s);
        default: return -1;
    }
}

}  // namespace xrs
