// TEST INFRASTRUCTURE (diff kit) — records the per-iteration summary of the REAL ceres::Solve that the reference's unchanged
// Estimator::optimization() calls (vins_estimator/src/estimator.cpp:803-818: DENSE_SCHUR, DOGLEG, max_num_iterations =
// NUM_ITERATIONS, max_solver_time_in_seconds = SOLVER_TIME).  The reference keeps its Solver::Summary in a local variable, so the
// call is wrapped at LINK time (oracle/Makefile: -Wl,--wrap=<mangled ceres::Solve>): the wrapper copies the options, adds an
// IterationCallback and forwards to the real function.  Nothing of the reference is edited.
// NOT BUILT IN THE GRAFT IMAGE (no Eigen / Ceres there): written against the public API of Ceres 1.14 (solver.h, iteration_callback.h).
// By default the wall-clock cap is lifted (max_solver_time_in_seconds = 1e9) so that goldens do not depend on the machine's speed —
// oracle/ASSUMPTIONS.md C7; set VINS_REF_KEEP_TIME_CAP=1 in the environment to keep the reference's 0.04 s.
#include <cmath>
#include <cstdlib>
#include <limits>

#include <ceres/ceres.h>

#include "ceres_real_trace.h"

namespace vins_ref_real {
Summary last;
namespace {
class Recorder : public ceres::IterationCallback {
  public:
    ceres::CallbackReturnType operator()(const ceres::IterationSummary& s) override {
        Iter it;
        it.iteration = s.iteration;
        it.step_is_valid = s.step_is_valid;
        it.step_is_successful = s.step_is_successful;
        // Ceres: s.cost = cost of the point the iteration ends at, s.cost_change = (cost it started from) - (candidate's cost).
        // The rows the tests compare (stand-in convention, oracle/ref_stubs/ceres/ceres.h): cost = where the iteration STARTED,
        // candidate_cost next to it.
        it.cost = s.iteration == 0 ? s.cost : prev_cost;
        it.cost_change = s.cost_change;
        it.gradient_max_norm = s.gradient_max_norm;
        it.step_norm = s.step_norm;
        it.relative_decrease = s.relative_decrease;
        it.trust_region_radius = s.trust_region_radius;
        it.candidate_cost = s.iteration == 0 ? 0.0 : prev_cost - s.cost_change;
        prev_cost = s.cost;
        it.model_cost_change = s.relative_decrease != 0.0 ? s.cost_change / s.relative_decrease : 0.0;
        it.mu = std::numeric_limits<double>::quiet_NaN();
        rows.push_back(it);
        return ceres::SOLVER_CONTINUE;
    }
    std::vector<Iter> rows;
    double prev_cost = 0;
};
}  // namespace
}  // namespace vins_ref_real

// ceres::Solve(const Solver::Options&, Problem*, Solver::Summary*)
extern "C" void __real__ZN5ceres5SolveERKNS_6Solver7OptionsEPNS_7ProblemEPNS0_7SummaryE(const ceres::Solver::Options&, ceres::Problem*, ceres::Solver::Summary*);
extern "C" void __wrap__ZN5ceres5SolveERKNS_6Solver7OptionsEPNS_7ProblemEPNS0_7SummaryE(const ceres::Solver::Options& options, ceres::Problem* problem,
                                                                                      ceres::Solver::Summary* summary) {
    ceres::Solver::Options o = options;
    vins_ref_real::Recorder rec;
    o.callbacks.push_back(&rec);
    const char* keep = std::getenv("VINS_REF_KEEP_TIME_CAP");
    if (!(keep && keep[0] == '1')) o.max_solver_time_in_seconds = 1e9;
    __real__ZN5ceres5SolveERKNS_6Solver7OptionsEPNS_7ProblemEPNS0_7SummaryE(o, problem, summary);
    vins_ref_real::last.iterations = rec.rows;
    vins_ref_real::last.initial_cost = summary->initial_cost;
    vins_ref_real::last.final_cost = summary->final_cost;
    vins_ref_real::last.termination_type = summary->termination_type == ceres::CONVERGENCE ? 0 : summary->termination_type == ceres::NO_CONVERGENCE ? 1 : 2;
}
