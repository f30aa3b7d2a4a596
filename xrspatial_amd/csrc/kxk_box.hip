// Box masks (np.ones((k, k)), k = 3..25): float32 statistics through the column walker.
#define XRS_WALK_SHAPE BoxShape
#define XRS_WALK_KERNEL focal_box_kernel
#define XRS_WALK_ENTRY try_launch_focal_box_f32
#include "walk_f32_impl.h"
