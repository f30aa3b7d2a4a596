// mean / var / std / sum only over circle masks (radius 4..12 cells): one pass of the second-generation walker.
#define XRS_WALK_SHAPE CircleShape
#define XRS_WALK_KERNEL focal_circle2_mom_kernel
#define XRS_WALK_ENTRY try_launch_focal_circle2_mom
#define XRS_WALK2_MM 0
#define XRS_WALK2_MOM 1
#include "walk2_impl.h"
