// TEST INFRASTRUCTURE (diff kit): see ../opencv.hpp
#include "../../../ref_stubs/opencv2/core/eigen.hpp"
