// mean / var / std / sum over annulus_kernel(1, 1, 9, RI), RI = 1 .. 8: the float32 trailing-shift moments walker.
#define XRS_MOM_ANNULUS_R 9
#define XRS_MOM_ENTRY try_launch_focal_mom_annulus9
#include "mom_impl.h"
