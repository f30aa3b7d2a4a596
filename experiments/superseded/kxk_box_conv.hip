// convolve_2d with a constant box kernel (np.ones((k, k)) / k**2): column walker.
#define XRS_WALK_SHAPE BoxShape
#define XRS_WALK_KERNEL conv_box_kernel
#define XRS_WALK_ENTRY try_launch_conv_box
#include "walk_conv_impl.h"
