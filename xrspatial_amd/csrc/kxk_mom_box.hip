// mean / var / std / sum over box masks (np.ones((k, k)), k = 9..25): the float32 trailing-shift moments walker.
#define XRS_MOM_SHAPE BoxShape
#define XRS_MOM_ENTRY try_launch_focal_mom_box
#include "mom_impl.h"
