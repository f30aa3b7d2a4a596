// mean / var / std / sum over circular masks (circle_kernel, radius 4..12 cells): the float32 trailing-shift moments walker.
#define XRS_MOM_SHAPE CircleShape
#define XRS_MOM_ENTRY try_launch_focal_mom_circle
#include "mom_impl.h"
