// TEST INFRASTRUCTURE — stand-in for <opencv2/calib3d/calib3d.hpp>: see ../cv_standin.h
#include "../../cv_standin.h"
