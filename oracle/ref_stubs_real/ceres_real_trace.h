// TEST INFRASTRUCTURE (diff kit, `make -C oracle ref_real`) — what the driver reads of the last ceres::Solve when the reference's
// translation units are built against the REAL Ceres: the fields of the stand-in's Solver::Summary that tests compare
// (oracle/ref_stubs/ceres/ceres.h), filled from ceres::IterationSummary by ceres_real_trace.cc.
#ifndef VINS_REF_CERES_REAL_TRACE_H
#define VINS_REF_CERES_REAL_TRACE_H
#include <vector>
namespace vins_ref_real {
struct Iter {
    int iteration = 0;
    bool step_is_valid = false, step_is_successful = false;
    double cost = 0, cost_change = 0, gradient_max_norm = 0, step_norm = 0, relative_decrease = 0, trust_region_radius = 0;
    double candidate_cost = 0, model_cost_change = 0, mu = 0;   // derived: cost - cost_change, cost_change / relative_decrease, (not exposed by Ceres: NaN)
    int exit_reason = 0;                                         // not exposed per iteration by Ceres: 0
};
struct Summary {
    std::vector<Iter> iterations;
    double initial_cost = 0, final_cost = 0;
    int termination_type = 0;                                    // 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE (ceres::TerminationType)
};
extern Summary last;
}  // namespace vins_ref_real
#endif
