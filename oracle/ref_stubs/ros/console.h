// TEST INFRASTRUCTURE — stand-in for <ros/console.h>: the logging macros of the reference compile to nothing
// (arguments are type-checked by the compiler but never evaluated), so the reference sources build without ROS.
#ifndef VINS_REF_STUB_ROS_CONSOLE_H
#define VINS_REF_STUB_ROS_CONSOLE_H
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <map>      // the real ros/console.h pulls these in; feature_manager.h relies on it
#include <vector>
#include <string>
#define VINS_REF_NOLOG(...) do { if (0) { std::printf(__VA_ARGS__); } } while (0)
#define VINS_REF_NOLOG_STREAM(x) do { if (0) { std::stringstream vins_ref_ss; vins_ref_ss << x; } } while (0)
#define ROS_DEBUG(...) VINS_REF_NOLOG(__VA_ARGS__)
#define ROS_INFO(...) VINS_REF_NOLOG(__VA_ARGS__)
#define ROS_WARN(...) VINS_REF_NOLOG(__VA_ARGS__)
#define ROS_ERROR(...) VINS_REF_NOLOG(__VA_ARGS__)
#define ROS_DEBUG_STREAM(x) VINS_REF_NOLOG_STREAM(x)
#define ROS_INFO_STREAM(x) VINS_REF_NOLOG_STREAM(x)
#define ROS_WARN_STREAM(x) VINS_REF_NOLOG_STREAM(x)
#define ROS_ERROR_STREAM(x) VINS_REF_NOLOG_STREAM(x)
#endif
