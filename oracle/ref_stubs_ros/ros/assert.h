// TEST INFRASTRUCTURE — <ros/assert.h> for the front-end builds: ROS_ASSERT / ROS_BREAK of ../../ref_stubs (active: they abort)
#include "../../ref_stubs/ros/assert.h"
