// TEST INFRASTRUCTURE — stand-in for <opencv2/core/core.hpp>: see ../cv_standin.h
#include "../../cv_standin.h"
