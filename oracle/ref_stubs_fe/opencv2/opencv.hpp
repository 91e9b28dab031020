// TEST INFRASTRUCTURE — stand-in for <opencv2/opencv.hpp>: see ../cv_standin.h
#include "../cv_standin.h"
