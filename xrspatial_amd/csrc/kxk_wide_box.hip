// Focal mean / window sum over box masks (np.ones((k, k)), 7x7 .. 25x25): the wide row walker.
#define XRS_WIDE_SHAPE BoxShape
#define XRS_WIDE_ENTRY try_launch_focal_wide_box
#define XRS_WIDE_CONV_ENTRY try_launch_conv_wide_box
#include "wide_impl.h"
