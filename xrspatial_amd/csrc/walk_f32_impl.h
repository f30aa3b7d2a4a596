// Float32 statistics (row-major sum, max, min, range) of focal_stats / focal.apply through the column walker of
// circle_walk.h, for one mask shape (XRS_WALK_SHAPE) and radius 1..12 cells.  For radius <= 3 the float64 moments ride
// along in the same kernel (one read of the raster for all seven statistics).  The tap-by-tap walk of kxk.hip this
// replaces spends 6 VALU + 6 SALU instructions per tap on a 25x25 mask (profiles/r01/pmc_focal25_sum.json).
// Included by kxk_circle.hip and kxk_box.hip, which define XRS_WALK_SHAPE / XRS_WALK_KERNEL / XRS_WALK_ENTRY.
#include "circle_walk.h"

using namespace xrs;

namespace {

template <int R, bool WANT_SUM, bool WANT_MM, bool F64>
__global__ void __launch_bounds__(256) XRS_WALK_KERNEL(const WalkGeom g, const WalkOuts o) {
    walk_tile<R, XRS_WALK_SHAPE, true, WANT_SUM, WANT_MM, F64>(g, o);
}

template <int R, bool WANT_SUM, bool WANT_MM, bool F64>
int launch(WalkGeom &g, const WalkOuts &o, hipStream_t s) {
    long grid;
    if (int rc = walk_grid(g, &grid)) return rc;
    hipLaunchKernelGGL((XRS_WALK_KERNEL<R, WANT_SUM, WANT_MM, F64>), dim3((unsigned)grid), dim3(256), 0, s, g, o);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <int R>
int dispatch(WalkGeom &g, const WalkOuts &o, const double *kernel, bool with_moments, hipStream_t s) {
    if (!is_shape<R, XRS_WALK_SHAPE>(kernel)) return -1;
    const bool ws = o.sum, wm = o.max || o.min || o.range;
    if (with_moments) {
        if constexpr (R <= 3) return launch<R, true, true, true>(g, o, s);
        else return -1;
    }
    if (ws && wm) return launch<R, true, true, false>(g, o, s);
    if (ws) return launch<R, true, false, false>(g, o, s);
    return launch<R, false, true, false>(g, o, s);
}

}  // namespace

namespace xrs {

// 0 = launched, -1 = not the shape / a radius this file is instantiated for (caller walks the taps), > 0 = error.
// out_mean / out_var / out_std non-null: all seven statistics in one kernel (radius 2 and 3 only).
int XRS_WALK_ENTRY(const float *in, float *out_sum, float *out_max, float *out_min, float *out_range,
                                float *out_mean, float *out_var, float *out_std, long rows, long cols, long ld_in,
                                long ld_out, const double *kernel, int krows, int kcols, int halo_top, int halo_bot,
                                hipStream_t s) {
    if (krows != kcols || !(krows & 1)) return -1;
    const bool moments = out_mean || out_var || out_std;
    if (!out_sum && !out_max && !out_min && !out_range && !moments) return 0;
    WalkGeom g;
    memset(&g, 0, sizeof(g));
    g.in = in; g.rows = rows; g.cols = cols; g.ld_in = ld_in; g.ld_out = ld_out;
    g.halo_top = halo_top; g.halo_bot = halo_bot;
    const WalkOuts o = {out_sum, out_max, out_min, out_range, out_mean, out_var, out_std};
    switch (krows / 2) {
        case 1: return dispatch<1>(g, o, kernel, moments, s);
        case 2: return dispatch<2>(g, o, kernel, moments, s);
        case 3: return dispatch<3>(g, o, kernel, moments, s);
        case 4: return dispatch<4>(g, o, kernel, moments, s);
        case 5: return dispatch<5>(g, o, kernel, moments, s);
        case 6: return dispatch<6>(g, o, kernel, moments, s);
        case 7: return dispatch<7>(g, o, kernel, moments, s);
        case 8: return dispatch<8>(g, o, kernel, moments, s);
        case 9: return dispatch<9>(g, o, kernel, moments, s);
        case 10: return dispatch<10>(g, o, kernel, moments, s);
        case 11: return dispatch<11>(g, o, kernel, moments, s);
        case 12: return dispatch<12>(g, o, kernel, moments, s);
        default: return -1;
    }
}

}  // namespace xrs
