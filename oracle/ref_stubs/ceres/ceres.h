// TEST INFRASTRUCTURE — header stand-in for <ceres/ceres.h>, written from scratch for this repo (Ceres is not installed).
//
// It lets the reference's estimator / factor sources compile UNCHANGED into oracle/_ref/libvins_ref.so:
//  * the modelling API is the real one (CostFunction / SizedCostFunction / LocalParameterization / LossFunction /
//    Problem with ownership semantics) so that `Estimator::optimization()` (estimator.cpp:670-1003) builds its problem
//    exactly as it does against Ceres;
//  * `ceres::Solve` (solver_stub.cc) is this repo's RESTATEMENT of Ceres' trust-region minimiser in the configuration
//    estimator.cpp:803-815 selects (DENSE_SCHUR + traditional DOGLEG, Jacobi scaling, monotonic steps; every default
//    listed in oracle/ASSUMPTIONS.md C1-C8).  It is generic over the Problem — it knows nothing about VINS.
// So oracle/_ref pins: the factors, the problem construction, the marginalization and the window bookkeeping (the
// reference's own code); it does NOT pin the third-party minimiser (restated here, as everywhere else in oracle/).
#ifndef VINS_REF_STUB_CERES_H
#define VINS_REF_STUB_CERES_H

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <map>
#include <numeric>
#include <string>
#include <vector>

namespace ceres {

typedef int32_t int32;

class CostFunction {
  public:
    CostFunction() : num_residuals_(0) {}
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
    const std::vector<int32> &parameter_block_sizes() const { return parameter_block_sizes_; }
    int num_residuals() const { return num_residuals_; }

  protected:
    std::vector<int32> *mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
    void set_num_residuals(int n) { num_residuals_ = n; }

  private:
    std::vector<int32> parameter_block_sizes_;
    int num_residuals_;
};

template <int kNumResiduals, int... Ns> class SizedCostFunction : public CostFunction {
  public:
    SizedCostFunction() {
        set_num_residuals(kNumResiduals);
        *mutable_parameter_block_sizes() = std::vector<int32>{Ns...};
    }
    virtual ~SizedCostFunction() {}
};

// Named by initial/initial_sfm.h only (global SfM — out of scope); never evaluated in this build.
template <typename Functor, int kNumResiduals, int... Ns> class AutoDiffCostFunction : public SizedCostFunction<kNumResiduals, Ns...> {
  public:
    explicit AutoDiffCostFunction(Functor *f) : functor_(f) {}
    virtual ~AutoDiffCostFunction() { delete functor_; }
    virtual bool Evaluate(double const *const *, double *, double **) const {
        std::fprintf(stderr, "oracle/_ref: AutoDiffCostFunction::Evaluate is not provided by the ceres stand-in\n");
        std::abort();
    }

  private:
    Functor *functor_;
};

class LocalParameterization {
  public:
    virtual ~LocalParameterization() {}
    virtual bool Plus(const double *x, const double *delta, double *x_plus_delta) const = 0;
    virtual bool ComputeJacobian(const double *x, double *jacobian) const = 0;   // GlobalSize x LocalSize, row-major
    virtual int GlobalSize() const = 0;
    virtual int LocalSize() const = 0;
};

class LossFunction {
  public:
    virtual ~LossFunction() {}
    virtual void Evaluate(double sq_norm, double out[3]) const = 0;   // rho, rho', rho''
};
class TrivialLoss : public LossFunction {
  public:
    virtual void Evaluate(double s, double rho[3]) const { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
};
// rho(s) = b log(1 + s / b), b = a^2   (loss_function.h)
class CauchyLoss : public LossFunction {
  public:
    explicit CauchyLoss(double a) : b_(a * a), c_(1.0 / b_) {}
    virtual void Evaluate(double s, double rho[3]) const {
        const double sum = 1.0 + s * c_;
        const double inv = 1.0 / sum;
        rho[0] = b_ * std::log(sum);
        rho[1] = std::max(std::numeric_limits<double>::min(), inv);
        rho[2] = -c_ * (inv * inv);
    }

  private:
    const double b_, c_;
};
class HuberLoss : public LossFunction {
  public:
    explicit HuberLoss(double a) : a_(a), b_(a * a) {}
    virtual void Evaluate(double s, double rho[3]) const {
        if (s > b_) {
            const double r = std::sqrt(s);
            rho[0] = 2.0 * a_ * r - b_;
            rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
            rho[2] = -rho[1] / (2.0 * s);
        } else {
            rho[0] = s;
            rho[1] = 1.0;
            rho[2] = 0.0;
        }
    }

  private:
    const double a_, b_;
};

enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
enum DoglegType { TRADITIONAL_DOGLEG, SUBSPACE_DOGLEG };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };

struct IterationSummary {
    int iteration = 0;
    bool step_is_valid = false, step_is_nonmonotonic = false, step_is_successful = false;
    double cost = 0, cost_change = 0, gradient_max_norm = 0, gradient_norm = 0, step_norm = 0, relative_decrease = 0,
           trust_region_radius = 0, eta = 0, step_size = 0;
    // extras of the stand-in (per-iteration trace compared by tests/test_ref_parity.py)
    double candidate_cost = 0, model_cost_change = 0, mu = 0;
    int exit_reason = 0;   // 0 none, 1 parameter tolerance, 2 function tolerance, 3 gradient tolerance
};

enum CallbackReturnType { SOLVER_CONTINUE, SOLVER_ABORT, SOLVER_TERMINATE_SUCCESSFULLY };
class IterationCallback {       // (the stand-in's Solve() reports every iteration summary to the callbacks after the fact)
  public:
    virtual ~IterationCallback() {}
    virtual CallbackReturnType operator()(const IterationSummary &summary) = 0;
};

class Problem {
  public:
    struct Options {};
    Problem() {}
    explicit Problem(const Options &) {}
    ~Problem();
    void AddParameterBlock(double *values, int size);
    void AddParameterBlock(double *values, int size, LocalParameterization *local_parameterization);
    void SetParameterBlockConstant(double *values);
    void SetParameterBlockVariable(double *values);
    void AddResidualBlock(CostFunction *cost, LossFunction *loss, const std::vector<double *> &parameter_blocks);
    template <typename... Ts> void AddResidualBlock(CostFunction *cost, LossFunction *loss, double *x0, Ts *...xs) {
        AddResidualBlock(cost, loss, std::vector<double *>{x0, xs...});
    }
    int NumParameterBlocks() const { return static_cast<int>(blocks_.size()); }
    int NumResidualBlocks() const { return static_cast<int>(residuals_.size()); }

    // ---- internals shared with solver_stub.cc
    struct ParameterBlock {
        double *user;
        int size;
        LocalParameterization *lp;
        bool constant;
        int local_size() const { return lp ? lp->LocalSize() : size; }
    };
    struct ResidualBlock {
        CostFunction *cost;
        LossFunction *loss;
        std::vector<int> blocks;   // indices into blocks_
    };
    std::vector<ParameterBlock> blocks_;   // in order of first mention (AddParameterBlock or AddResidualBlock)
    std::map<double *, int> index_;
    std::vector<ResidualBlock> residuals_;

  private:
    Problem(const Problem &);
    void operator=(const Problem &);
    int intern(double *values, int size);
};

class Solver {
  public:
    struct Options {
        LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
        TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;
        DoglegType dogleg_type = TRADITIONAL_DOGLEG;
        int max_num_iterations = 50;
        double max_solver_time_in_seconds = 1e9;   // see ASSUMPTIONS C7: read, never enforced (non-reproducible)
        int num_threads = 1;
        bool minimizer_progress_to_stdout = false;
        bool use_nonmonotonic_steps = false;
        bool use_explicit_schur_complement = false;
        bool jacobi_scaling = true;
        double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
        double min_relative_decrease = 1e-3;
        double min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
        int max_num_consecutive_invalid_steps = 5;
        double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
        std::vector<IterationCallback *> callbacks;
    };
    struct Summary {
        TerminationType termination_type = NO_CONVERGENCE;
        double initial_cost = 0, final_cost = 0;
        int num_successful_steps = 0, num_unsuccessful_steps = 0;
        int num_parameters_reduced = 0, num_effective_parameters_reduced = 0, num_residuals_reduced = 0;
        std::vector<IterationSummary> iterations;
        std::string message;
        std::string BriefReport() const;
        std::string FullReport() const { return BriefReport(); }
    };
};

void Solve(const Solver::Options &options, Problem *problem, Solver::Summary *summary);

}  // namespace ceres
#endif
