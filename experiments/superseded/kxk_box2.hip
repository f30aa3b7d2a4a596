// Box masks (np.ones((k, k)), 9x9 .. 25x25): all seven statistics in one pass, second-generation walker.
#define XRS_WALK_SHAPE BoxShape
#define XRS_WALK_KERNEL focal_box2_kernel
#define XRS_WALK_ENTRY try_launch_focal_box2
#include "walk2_impl.h"
