// Circular masks (circle_kernel, radius 1..12 cells): float32 statistics through the column walker.
#define XRS_WALK_SHAPE CircleShape
#define XRS_WALK_KERNEL focal_circle_kernel
#define XRS_WALK_ENTRY try_launch_focal_circle_f32
#include "walk_f32_impl.h"
