// TEST INFRASTRUCTURE — stand-in for <opencv2/highgui/highgui.hpp>: see ../cv_standin.h
#include "../../cv_standin.h"
