// max / min / range over annulus_kernel(1, 1, R, RI), R = 10 .. 11, RI = 1 .. R - 1: the two-rows-per-step extrema walker.
#define XRS_EXT_ANNULUS_RMIN 10
#define XRS_EXT_ANNULUS_RMAX 11
#define XRS_EXT_ENTRY try_launch_focal_ext_annulus_b
#include "ext_impl.h"
