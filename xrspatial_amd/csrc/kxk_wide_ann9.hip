// Focal mean and uniform-weight convolve_2d over annulus_kernel(1, 1, 9, RI), RI = 1 .. 8: the wide row walker
// (a row with a hole is the difference of two centred runs of the lane's prefix sums).
#define XRS_WIDE_ANNULUS_R 9
#define XRS_WIDE_ENTRY try_launch_wide_annulus9
#include "wide_impl.h"
