// TEST INFRASTRUCTURE — stand-in for <ros/assert.h>: assertions stay ACTIVE (a failed ROS_ASSERT aborts, like in ROS).
#ifndef VINS_REF_STUB_ROS_ASSERT_H
#define VINS_REF_STUB_ROS_ASSERT_H
#include <cstdio>
#include <cstdlib>
#define ROS_BREAK() do { std::fprintf(stderr, "ROS_BREAK at %s:%d\n", __FILE__, __LINE__); std::abort(); } while (0)
#define ROS_ASSERT(cond) do { if (!(cond)) { std::fprintf(stderr, "ROS_ASSERT(%s) failed at %s:%d\n", #cond, __FILE__, __LINE__); std::abort(); } } while (0)
#define ROS_ASSERT_MSG(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "ROS_ASSERT_MSG(%s) failed at %s:%d: ", #cond, __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); std::abort(); } } while (0)
#endif
