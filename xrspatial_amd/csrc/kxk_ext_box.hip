// max / min / range over box masks (np.ones((k, k)), k = 9..25): the two-rows-per-step extrema walker.
#define XRS_EXT_SHAPE BoxShape
#define XRS_EXT_ENTRY try_launch_focal_ext_box
#include "ext_impl.h"
