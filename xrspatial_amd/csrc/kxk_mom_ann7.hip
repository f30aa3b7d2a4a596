// mean / var / std / sum over annulus_kernel(1, 1, 7, RI), RI = 1 .. 6: the float32 trailing-shift moments walker.
#define XRS_MOM_ANNULUS_R 7
#define XRS_MOM_ENTRY try_launch_focal_mom_annulus7
#include "mom_impl.h"
