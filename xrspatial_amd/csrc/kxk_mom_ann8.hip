// mean / var / std / sum over annulus_kernel(1, 1, 8, RI), RI = 1 .. 7: the float32 trailing-shift moments walker.
#define XRS_MOM_ANNULUS_R 8
#define XRS_MOM_ENTRY try_launch_focal_mom_annulus8
#include "mom_impl.h"
