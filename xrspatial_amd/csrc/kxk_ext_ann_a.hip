// max / min / range over annulus_kernel(1, 1, R, RI), R = 4 .. 9, RI = 1 .. R - 1: the two-rows-per-step extrema walker.
#define XRS_EXT_ANNULUS_RMIN 4
#define XRS_EXT_ANNULUS_RMAX 9
#define XRS_EXT_ENTRY try_launch_focal_ext_annulus_a
#include "ext_impl.h"
