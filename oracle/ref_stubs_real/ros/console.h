// TEST INFRASTRUCTURE (diff kit, `make -C oracle ref_real`): the ROS / OpenCV NAME stand-ins of ../../ref_stubs without its Eigen / Ceres stand-ins
#include "../../ref_stubs/ros/console.h"
