// All seven statistics over small box masks (np.ones((5, 5)), np.ones((7, 7))): the strip walker.
#define XRS_SW_SHAPE BoxShape
#define XRS_SW_ENTRY try_launch_focal_sw_box
#include "sw_impl.h"
