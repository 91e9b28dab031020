// TEST INFRASTRUCTURE — stand-in for <opencv2/imgproc/imgproc.hpp>: see ../cv_standin.h
#include "../../cv_standin.h"
