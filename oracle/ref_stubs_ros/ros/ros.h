// TEST INFRASTRUCTURE — stand-in for <ros/ros.h> as far as feature_tracker/src/{feature_tracker_node,parameters}.cpp use it.
// Publishers do not publish: every message handed to ros::Publisher::publish() is kept in a process-wide capture list
// (topic, type-erased shared_ptr) that the driver (../ref_fe_driver.cpp) reads back — that list is the "topic contract" of
// SURVEY.md Appendix E as observed from the reference's own img_callback().
#ifndef VINS_REF_FE_ROS_ROS_H
#define VINS_REF_FE_ROS_ROS_H
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>
#include <boost/shared_ptr.hpp>      // (the stand-in of ref_stubs_fe/boost or the real one: messages travel as boost::shared_ptr, like in ROS)
#include <ros/console.h>      // logging macros compile to nothing
#include <ros/assert.h>
#define ROSCONSOLE_DEFAULT_NAME "ros"
namespace ros {
struct Time {
    uint32_t sec = 0, nsec = 0;
    Time() {}
    explicit Time(double t) { fromSec(t); }
    Time& fromSec(double t) {
        sec = static_cast<uint32_t>(t);
        nsec = static_cast<uint32_t>((t - sec) * 1e9 + 0.5);
        if (nsec >= 1000000000u) { sec++; nsec -= 1000000000u; }
        return *this;
    }
    double toSec() const { return static_cast<double>(sec) + 1e-9 * static_cast<double>(nsec); }
    static Time now() { return Time(); }
};
struct CapturedMessage {
    std::string topic;
    boost::shared_ptr<const void> msg;
};
inline std::vector<CapturedMessage>& captured() {
    static std::vector<CapturedMessage> list;
    return list;
}
class Publisher {
  public:
    Publisher() {}
    explicit Publisher(const std::string& t) : topic_(t) {}
    template <typename M> void publish(const boost::shared_ptr<M>& m) const { captured().push_back({topic_, boost::shared_ptr<const void>(boost::shared_ptr<const M>(m))}); }
    template <typename M> void publish(const M& m) const { captured().push_back({topic_, boost::shared_ptr<const void>(boost::shared_ptr<const M>(new M(m)))}); }
    const std::string& getTopic() const { return topic_; }
  private:
    std::string topic_;
};
class Subscriber {};
class NodeHandle {
  public:
    NodeHandle() {}
    explicit NodeHandle(const std::string&) {}
    static std::map<std::string, std::string>& params() {      // private parameters of the node (config_file, vins_folder)
        static std::map<std::string, std::string> p;
        return p;
    }
    bool getParam(const std::string& name, std::string& v) const {
        auto it = params().find(name);
        if (it == params().end()) return false;
        v = it->second;
        return true;
    }
    void shutdown() {}
    template <typename M> Publisher advertise(const std::string& topic, int) { return Publisher(topic); }
    template <typename F> Subscriber subscribe(const std::string&, int, F) { return Subscriber(); }
};
inline void init(int&, char**, const std::string&) {}
inline void spin() {}
namespace console {
namespace levels { enum Level { Debug, Info, Warn, Error, Fatal }; }
inline bool set_logger_level(const std::string&, levels::Level) { return true; }
}  // namespace console
}  // namespace ros
#endif
