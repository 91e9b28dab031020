// TEST INFRASTRUCTURE — stand-in for <sensor_msgs/PointCloud.h> (+ ChannelFloat32)
#ifndef VINS_REF_FE_SENSOR_MSGS_POINTCLOUD_H
#define VINS_REF_FE_SENSOR_MSGS_POINTCLOUD_H
#include <string>
#include <vector>
#include <boost/shared_ptr.hpp>
#include <geometry_msgs/Point32.h>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct ChannelFloat32 {
    std::string name;
    std::vector<float> values;
};
struct PointCloud {
    std_msgs::Header header;
    std::vector<geometry_msgs::Point32> points;
    std::vector<ChannelFloat32> channels;
};
typedef boost::shared_ptr<PointCloud> PointCloudPtr;
typedef boost::shared_ptr<const PointCloud> PointCloudConstPtr;
}  // namespace sensor_msgs
#endif
