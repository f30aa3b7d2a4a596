// Box masks: float64 moments through the column walker.
#define XRS_WALK_SHAPE BoxShape
#define XRS_WALK_KERNEL focal_box_f64_kernel
#define XRS_WALK_ENTRY try_launch_focal_box_f64
#include "walk_f64_impl.h"
