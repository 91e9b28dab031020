// placeholder until the marginalization kernel lands
#include <hip/hip_runtime.h>
#include "ba_layout.h"
extern "C" hipError_t ba_launch_marg(const BaLayout& L, const BaPtrs& P, hipStream_t stream) { return hipSuccess; }
