// TEST INFRASTRUCTURE — stand-in for <opencv2/video/tracking.hpp>: see ../cv_standin.h
#include "../../cv_standin.h"
