// mean / var / std / sum over annulus_kernel(1, 1, 6, RI), RI = 1 .. 5: the float32 trailing-shift moments walker.
#define XRS_MOM_ANNULUS_R 6
#define XRS_MOM_ENTRY try_launch_focal_mom_annulus6
#include "mom_impl.h"
