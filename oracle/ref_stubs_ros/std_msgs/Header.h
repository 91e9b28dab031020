// TEST INFRASTRUCTURE — stand-in for <std_msgs/Header.h>
#ifndef VINS_REF_FE_STD_MSGS_HEADER_H
#define VINS_REF_FE_STD_MSGS_HEADER_H
#include <cstdint>
#include <string>
#include <ros/ros.h>
namespace std_msgs {
struct Header {
    uint32_t seq = 0;
    ros::Time stamp;
    std::string frame_id;
};
}  // namespace std_msgs
#endif
