// Focal mean / window sum over circular masks (circle_kernel, radius 3..12 cells): the wide row walker.
#define XRS_WIDE_SHAPE CircleShape
#define XRS_WIDE_ENTRY try_launch_focal_wide_circle
#define XRS_WIDE_CONV_ENTRY try_launch_conv_wide_circle
#include "wide_impl.h"
