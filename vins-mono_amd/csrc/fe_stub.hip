#include "vg_handle.h"
struct FeState { int dummy; };
extern "C" void fe_state_destroy(FeState* s) { delete s; }
