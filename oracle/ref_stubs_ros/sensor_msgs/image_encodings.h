// TEST INFRASTRUCTURE — stand-in for <sensor_msgs/image_encodings.h>
#ifndef VINS_REF_FE_SENSOR_MSGS_IMAGE_ENCODINGS_H
#define VINS_REF_FE_SENSOR_MSGS_IMAGE_ENCODINGS_H
#include <string>
namespace sensor_msgs {
namespace image_encodings {
const std::string MONO8 = "mono8";
const std::string BGR8 = "bgr8";
}  // namespace image_encodings
}  // namespace sensor_msgs
#endif
