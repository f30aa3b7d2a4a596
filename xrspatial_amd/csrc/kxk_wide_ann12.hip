// Focal mean and uniform-weight convolve_2d over annulus_kernel(1, 1, 12, RI), RI = 1 .. 11: the wide row walker
// (a row with a hole is the difference of two centred runs of the lane's prefix sums).
// (2 waves per SIMD: at 3 the two-run rows on top of the radius-12 ring spill 3 .. 31 registers INTO the round loop -- 47 scratch
//  accesses per 5 rows, tools/loopscan.py -- and the 25x25 ring's mean takes 1.40 ms instead of 0.65: profiles/r04/ab_annulus_wide.log)
#define XRS_WIDE_WAVES 2
#define XRS_WIDE_ANNULUS_R 12
#define XRS_WIDE_ENTRY try_launch_wide_annulus12
#include "wide_impl.h"
