// All seven statistics over small circular masks (circle_kernel radius 2, 3 cells: 5x5, 7x7): the strip walker.
#define XRS_SW_SHAPE CircleShape
#define XRS_SW_ENTRY try_launch_focal_sw_circle
#include "sw_impl.h"
