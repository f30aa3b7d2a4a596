// mean / var / std / sum over annulus_kernel(1, 1, 5, RI), RI = 1 .. 4: the float32 trailing-shift moments walker.
#define XRS_MOM_ANNULUS_R 5
#define XRS_MOM_ENTRY try_launch_focal_mom_annulus5
#include "mom_impl.h"
