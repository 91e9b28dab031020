// TEST INFRASTRUCTURE — stand-in for <ros/ros.h>: ros::Time / ros::NodeHandle as far as the estimator sources name them.
#ifndef VINS_REF_STUB_ROS_ROS_H
#define VINS_REF_STUB_ROS_ROS_H
#include <cstdint>
#include <map>
#include <string>
#include "console.h"
#include "assert.h"
namespace ros {
struct Time {
    uint32_t sec = 0, nsec = 0;
    Time() {}
    explicit Time(double t) { fromSec(t); }
    Time &fromSec(double t) {
        sec = static_cast<uint32_t>(t);
        nsec = static_cast<uint32_t>((t - sec) * 1e9 + 0.5);
        if (nsec >= 1000000000u) { sec++; nsec -= 1000000000u; }
        return *this;
    }
    double toSec() const { return static_cast<double>(sec) + 1e-9 * static_cast<double>(nsec); }
};
class NodeHandle {
  public:
    NodeHandle() {}
    explicit NodeHandle(const std::string &) {}
    // private parameters of the node (the launch files pass config_file / vins_folder this way)
    std::map<std::string, std::string> params;
    bool getParam(const std::string &name, std::string &v) const {
        auto it = params.find(name);
        if (it == params.end()) return false;
        v = it->second;
        return true;
    }
    void shutdown() {}
};
}  // namespace ros
#endif
