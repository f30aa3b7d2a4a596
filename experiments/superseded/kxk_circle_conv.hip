// convolve_2d with a normalised circle_kernel: column walker.
#define XRS_WALK_SHAPE CircleShape
#define XRS_WALK_KERNEL conv_circle_kernel
#define XRS_WALK_ENTRY try_launch_conv_circle
#include "walk_conv_impl.h"
