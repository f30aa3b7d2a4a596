// mean / var / std / sum over annulus_kernel(1, 1, 11, RI), RI = 1 .. 10: the float32 trailing-shift moments walker.
// (one column per lane: with two, the row-pattern differences on top of the radius-11 rings spill ~100 scratch accesses into the
//  round loop, each reload draining the DMA ring -- 3.4 ms instead of 1.45 for the circle; one column has none)
#define XRS_MOM_NC 1
#define XRS_MOM_ANNULUS_R 11
#define XRS_MOM_ENTRY try_launch_focal_mom_annulus11
#include "mom_impl.h"
