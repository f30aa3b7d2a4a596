// max / min / range over circular masks (circle_kernel, radius 4..12 cells): the two-rows-per-step extrema walker.
#define XRS_EXT_SHAPE CircleShape
#define XRS_EXT_ENTRY try_launch_focal_ext_circle
#include "ext_impl.h"
