// TEST INFRASTRUCTURE — stand-in for <geometry_msgs/Point32.h>
#ifndef VINS_REF_FE_GEOMETRY_MSGS_POINT32_H
#define VINS_REF_FE_GEOMETRY_MSGS_POINT32_H
namespace geometry_msgs {
struct Point32 { float x = 0, y = 0, z = 0; };
}  // namespace geometry_msgs
#endif
