// TEST INFRASTRUCTURE — stand-in for <std_msgs/Bool.h>
#ifndef VINS_REF_FE_STD_MSGS_BOOL_H
#define VINS_REF_FE_STD_MSGS_BOOL_H
namespace std_msgs {
struct Bool { bool data = false; };
}  // namespace std_msgs
#endif
