// TEST INFRASTRUCTURE — stand-in for <std_msgs/Float32.h>.
#ifndef VINS_REF_STUB_STD_MSGS_FLOAT32_H
#define VINS_REF_STUB_STD_MSGS_FLOAT32_H
namespace std_msgs { struct Float32 { float data = 0.f; }; }
#endif
