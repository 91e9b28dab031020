// TEST INFRASTRUCTURE — stand-in for <sensor_msgs/Imu.h> (feature_tracker_node.cpp includes it and uses nothing of it)
#ifndef VINS_REF_FE_SENSOR_MSGS_IMU_H
#define VINS_REF_FE_SENSOR_MSGS_IMU_H
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct Imu { std_msgs::Header header; };
}  // namespace sensor_msgs
#endif
