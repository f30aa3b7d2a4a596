// Circular masks (circle_kernel, radius 4..12 cells): all seven statistics in one pass, second-generation walker.
#define XRS_WALK_SHAPE CircleShape
#define XRS_WALK_KERNEL focal_circle2_kernel
#define XRS_WALK_ENTRY try_launch_focal_circle2
#include "walk2_impl.h"
