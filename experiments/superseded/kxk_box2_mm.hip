// max / min / range only over box masks (radius 4..12 cells): one pass of the second-generation walker.
#define XRS_WALK_SHAPE BoxShape
#define XRS_WALK_KERNEL focal_box2_mm_kernel
#define XRS_WALK_ENTRY try_launch_focal_box2_mm
#define XRS_WALK2_MM 1
#define XRS_WALK2_MOM 0
#include "walk2_impl.h"
