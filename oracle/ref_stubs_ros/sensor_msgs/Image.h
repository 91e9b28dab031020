// TEST INFRASTRUCTURE — stand-in for <sensor_msgs/Image.h>
#ifndef VINS_REF_FE_SENSOR_MSGS_IMAGE_H
#define VINS_REF_FE_SENSOR_MSGS_IMAGE_H
#include <cstdint>
#include <string>
#include <vector>
#include <boost/shared_ptr.hpp>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct Image {
    std_msgs::Header header;
    uint32_t height = 0, width = 0;
    std::string encoding;
    uint8_t is_bigendian = 0;
    uint32_t step = 0;
    std::vector<uint8_t> data;
};
typedef boost::shared_ptr<Image> ImagePtr;
typedef boost::shared_ptr<const Image> ImageConstPtr;
}  // namespace sensor_msgs
#endif
