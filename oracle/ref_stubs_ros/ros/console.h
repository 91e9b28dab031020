// TEST INFRASTRUCTURE — <ros/console.h> for the front-end builds: the logging macros of ../../ref_stubs (they compile to nothing)
#include "../../ref_stubs/ros/console.h"
