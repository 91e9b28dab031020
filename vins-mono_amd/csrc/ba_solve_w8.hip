// ba_solve_w8.hip — the 8-wavefront build of the per-round solve kernel (ba_solve_w8_kernel).
//
// Same source, same phase functions, same LDS carve as ba_solve_kernel (ba_pipeline.hip); only the number of threads a window's
// workgroup runs differs.  The 4-wavefront form exists so that TWO windows share a CU when a batch fills the chip (round 5); a call
// with a few windows -- Estimator::optimization() is ONE window per call (vins_estimator/src/estimator_node.cpp:314 ->
// estimator.cpp:482) -- leaves 255 CUs idle, and then the second half of the CU's wavefront slots is better spent on the same
// window: the MFMA tile phases, the assembly and every strided loop run over twice the threads (single window 92.7 -> 86.5 us per
// round, pipeline 0.985 -> 0.937 ms on the same box; profiles/r06sv8_*).  The host picks the form per batch (BaLayout::sv_w8,
// ba_host.hip); sums that are strided over the threads of the workgroup are grouped differently in the two forms, so results agree
// to rounding, not bit for bit.
#define SV_NT 512
#define BA_SOLVE_W8_TU
#include "ba_pipeline.hip"
