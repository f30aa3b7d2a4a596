// max / min / range over annulus_kernel(1, 1, R, RI), R = 12 .. 12, RI = 1 .. R - 1: the two-rows-per-step extrema walker.
#define XRS_EXT_ANNULUS_RMIN 12
#define XRS_EXT_ANNULUS_RMAX 12
#define XRS_EXT_ENTRY try_launch_focal_ext_annulus_c
#include "ext_impl.h"
