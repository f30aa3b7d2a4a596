// mean / var / std / sum over annulus_kernel(1, 1, 4, RI), RI = 1 .. 3: the float32 trailing-shift moments walker.
#define XRS_MOM_ANNULUS_R 4
#define XRS_MOM_ENTRY try_launch_focal_mom_annulus4
#include "mom_impl.h"
