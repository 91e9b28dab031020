// TEST INFRASTRUCTURE — ceres::Problem / ceres::Solve for the header stand-in ceres.h (see its header comment).
//
// ceres::Solve here is a generic dense restatement of Ceres' TrustRegionMinimizer + DoglegStrategy (traditional
// dogleg) + DENSE_SCHUR as configured by estimator.cpp:803-815; the same algorithm as oracle/ba_numpy.py `solve`
// (ASSUMPTIONS C1-C8), but driven by the Problem the REFERENCE code builds, through the reference's own
// CostFunction::Evaluate / LocalParameterization::Plus.  The third-party minimiser itself stays "restated from its
// published algorithm"; what this file adds to the oracle is that nothing VINS-specific is restated any more.
#include "ceres.h"

#include <algorithm>
#include <cassert>
#include <cstring>
#include <set>

namespace ceres {

// ------------------------------------------------------------------------------------------ Problem
Problem::~Problem() {
    std::set<CostFunction *> costs;
    std::set<LossFunction *> losses;
    std::set<LocalParameterization *> lps;
    for (auto &r : residuals_) {
        costs.insert(r.cost);
        if (r.loss) losses.insert(r.loss);
    }
    for (auto &b : blocks_)
        if (b.lp) lps.insert(b.lp);
    for (auto c : costs) delete c;
    for (auto l : losses) delete l;
    for (auto l : lps) delete l;
}
int Problem::intern(double *values, int size) {
    auto it = index_.find(values);
    if (it != index_.end()) {
        if (blocks_[it->second].size != size) {
            std::fprintf(stderr, "ceres stand-in: parameter block %p re-added with size %d (was %d)\n", (void *)values, size, blocks_[it->second].size);
            std::abort();
        }
        return it->second;
    }
    blocks_.push_back(ParameterBlock{values, size, nullptr, false});
    index_[values] = static_cast<int>(blocks_.size()) - 1;
    return index_[values];
}
void Problem::AddParameterBlock(double *values, int size) { intern(values, size); }
void Problem::AddParameterBlock(double *values, int size, LocalParameterization *lp) {
    int b = intern(values, size);
    if (lp) {
        if (lp->GlobalSize() != size) {
            std::fprintf(stderr, "ceres stand-in: local parameterization size mismatch\n");
            std::abort();
        }
        blocks_[b].lp = lp;
    }
}
void Problem::SetParameterBlockConstant(double *values) {
    auto it = index_.find(values);
    if (it == index_.end()) {
        std::fprintf(stderr, "ceres stand-in: SetParameterBlockConstant on an unknown block\n");
        std::abort();
    }
    blocks_[it->second].constant = true;
}
void Problem::SetParameterBlockVariable(double *values) {
    auto it = index_.find(values);
    if (it != index_.end()) blocks_[it->second].constant = false;
}
void Problem::AddResidualBlock(CostFunction *cost, LossFunction *loss, const std::vector<double *> &pb) {
    const std::vector<int32> &sizes = cost->parameter_block_sizes();
    if (sizes.size() != pb.size()) {
        std::fprintf(stderr, "ceres stand-in: cost function expects %zu parameter blocks, got %zu\n", sizes.size(), pb.size());
        std::abort();
    }
    ResidualBlock rb{cost, loss, {}};
    for (size_t i = 0; i < pb.size(); i++) rb.blocks.push_back(intern(pb[i], sizes[i]));
    residuals_.push_back(rb);
}

std::string Solver::Summary::BriefReport() const {
    char buf[256];
    std::snprintf(buf, sizeof buf, "Ceres stand-in: iterations %zu, initial cost %.6e, final cost %.6e, termination %d", iterations.size(),
                  initial_cost, final_cost, static_cast<int>(termination_type));
    return buf;
}

// ------------------------------------------------------------------------------------------ evaluator
namespace {
struct Program {
    Problem *p;
    std::vector<int> col;      // tangent column offset per block, -1 = constant or unused
    std::vector<int> active;   // blocks in the reduced program, program order
    int ncols = 0, nrows = 0;
    std::vector<std::vector<double>> x;   // ambient values per block (working copy)

    explicit Program(Problem *pp) : p(pp) {
        const int nb = static_cast<int>(p->blocks_.size());
        std::vector<char> used(nb, 0);
        for (auto &r : p->residuals_) {
            nrows += r.cost->num_residuals();
            for (int b : r.blocks) used[b] = 1;
        }
        col.assign(nb, -1);
        x.resize(nb);
        for (int b = 0; b < nb; b++) {
            auto &pb = p->blocks_[b];
            x[b].assign(pb.user, pb.user + pb.size);
            if (pb.constant || !used[b]) continue;
            col[b] = ncols;
            ncols += pb.local_size();
            active.push_back(b);
        }
    }
    // cost, residuals (loss-corrected), dense row-major J (tangent columns, loss-corrected) at `vals`
    bool evaluate(const std::vector<std::vector<double>> &vals, double *cost, std::vector<double> *r, std::vector<double> *J) const {
        double total = 0;
        if (r) r->assign(nrows, 0.0);
        if (J) J->assign(static_cast<size_t>(nrows) * ncols, 0.0);
        int row = 0;
        std::vector<double> res, lpj, tmp;
        std::vector<std::vector<double>> jac;
        for (auto &rb : p->residuals_) {
            const int nr = rb.cost->num_residuals(), np = static_cast<int>(rb.blocks.size());
            std::vector<const double *> par(np);
            std::vector<double *> jp(np, nullptr);
            res.assign(nr, 0.0);
            jac.resize(np);
            for (int i = 0; i < np; i++) {
                const int b = rb.blocks[i];
                par[i] = vals[b].data();
                if (J && col[b] >= 0) {
                    jac[i].assign(static_cast<size_t>(nr) * p->blocks_[b].size, 0.0);
                    jp[i] = jac[i].data();
                }
            }
            if (!rb.cost->Evaluate(par.data(), res.data(), J ? jp.data() : nullptr)) return false;
            double sq = 0;
            for (int k = 0; k < nr; k++) sq += res[k] * res[k];
            double sqrt_rho1 = 1.0, residual_scaling = 1.0, alpha_sq_norm = 0.0;
            if (rb.loss) {   // corrector.cc
                double rho[3];
                rb.loss->Evaluate(sq, rho);
                total += 0.5 * rho[0];
                sqrt_rho1 = std::sqrt(rho[1]);
                if (sq == 0.0 || rho[2] <= 0.0) {
                    residual_scaling = sqrt_rho1;
                    alpha_sq_norm = 0.0;
                } else {
                    const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
                    const double alpha = 1.0 - std::sqrt(D);
                    residual_scaling = sqrt_rho1 / (1 - alpha);
                    alpha_sq_norm = alpha / sq;
                }
            } else {
                total += 0.5 * sq;
            }
            if (J) {
                for (int i = 0; i < np; i++) {
                    const int b = rb.blocks[i];
                    if (col[b] < 0) continue;
                    auto &pb = p->blocks_[b];
                    const int gs = pb.size, ls = pb.local_size();
                    std::vector<double> &Jg = jac[i];
                    if (rb.loss) {
                        if (alpha_sq_norm == 0.0) {
                            for (auto &v : Jg) v *= sqrt_rho1;
                        } else {   // J = sqrt_rho1 (J - alpha_sq_norm r (r^T J))
                            for (int c = 0; c < gs; c++) {
                                double rtj = 0;
                                for (int k = 0; k < nr; k++) rtj += res[k] * Jg[k * gs + c];
                                for (int k = 0; k < nr; k++) Jg[k * gs + c] = sqrt_rho1 * (Jg[k * gs + c] - alpha_sq_norm * res[k] * rtj);
                            }
                        }
                    }
                    double *dst = J->data() + static_cast<size_t>(row) * ncols + col[b];
                    if (pb.lp) {
                        lpj.assign(static_cast<size_t>(gs) * ls, 0.0);
                        pb.lp->ComputeJacobian(vals[b].data(), lpj.data());
                        for (int k = 0; k < nr; k++)
                            for (int c = 0; c < ls; c++) {
                                double s = 0;
                                for (int g = 0; g < gs; g++) s += Jg[k * gs + g] * lpj[g * ls + c];
                                dst[static_cast<size_t>(k) * ncols + c] += s;
                            }
                    } else {
                        for (int k = 0; k < nr; k++)
                            for (int c = 0; c < gs; c++) dst[static_cast<size_t>(k) * ncols + c] += Jg[k * gs + c];
                    }
                }
            }
            if (r)
                for (int k = 0; k < nr; k++) (*r)[row + k] = res[k] * residual_scaling;
            row += nr;
        }
        *cost = total;
        return true;
    }
    void plus(const std::vector<std::vector<double>> &from, const std::vector<double> &delta, std::vector<std::vector<double>> *to) const {
        *to = from;
        for (int b : active) {
            auto &pb = p->blocks_[b];
            if (pb.lp) pb.lp->Plus(from[b].data(), delta.data() + col[b], (*to)[b].data());
            else
                for (int k = 0; k < pb.size; k++) (*to)[b][k] = from[b][k] + delta[col[b] + k];
        }
    }
    double ambient_norm(const std::vector<std::vector<double>> &a, const std::vector<std::vector<double>> *b) const {
        double s = 0;
        for (int blk : active)
            for (size_t k = 0; k < a[blk].size(); k++) {
                const double d = a[blk][k] - (b ? (*b)[blk][k] : 0.0);
                s += d * d;
            }
        return std::sqrt(s);
    }
};

struct Dense {   // row-major J with a per-row list of structurally non-zero columns
    int nrows, ncols;
    const std::vector<double> *J;
    std::vector<std::vector<int>> nz;
    void index() {
        nz.assign(nrows, {});
        for (int i = 0; i < nrows; i++)
            for (int c = 0; c < ncols; c++)
                if ((*J)[static_cast<size_t>(i) * ncols + c] != 0.0) nz[i].push_back(c);
    }
    double at(int i, int c) const { return (*J)[static_cast<size_t>(i) * ncols + c]; }
};

// DENSE_SCHUR: min |J y - r|^2 + |D y|^2; the columns in `elim` (independent 1-wide blocks) are eliminated first, the
// reduced system is factorised by dense LL^T.  false = factorisation failed (non-positive pivot / non-finite result).
bool dense_schur_solve(const Dense &A, const std::vector<double> &r, const std::vector<double> &D, const std::vector<char> &elim, std::vector<double> *y) {
    const int n = A.ncols;
    std::vector<double> H(static_cast<size_t>(n) * n, 0.0), g(n, 0.0);
    for (int i = 0; i < A.nrows; i++) {
        const auto &nz = A.nz[i];
        for (size_t a = 0; a < nz.size(); a++) {
            const double va = A.at(i, nz[a]);
            g[nz[a]] += va * r[i];
            for (size_t b = 0; b <= a; b++) H[static_cast<size_t>(nz[a]) * n + nz[b]] += va * A.at(i, nz[b]);
        }
    }
    for (int a = 0; a < n; a++) {
        H[static_cast<size_t>(a) * n + a] += D[a] * D[a];
        for (int b = 0; b < a; b++) H[static_cast<size_t>(b) * n + a] = H[static_cast<size_t>(a) * n + b];
    }
    std::vector<int> pi, li;
    for (int c = 0; c < n; c++) (elim[c] ? li : pi).push_back(c);
    const int R = static_cast<int>(pi.size());
    for (int l : li) {
        const double h = H[static_cast<size_t>(l) * n + l];
        if (!(h > 0) || !std::isfinite(h)) return false;
    }
    std::vector<double> S(static_cast<size_t>(R) * R), gr(R);
    for (int a = 0; a < R; a++) {
        double s = g[pi[a]];
        for (int l : li) s -= H[static_cast<size_t>(pi[a]) * n + l] * (g[l] / H[static_cast<size_t>(l) * n + l]);
        gr[a] = s;
        for (int b = 0; b <= a; b++) {
            double v = H[static_cast<size_t>(pi[a]) * n + pi[b]];
            for (int l : li) {
                const double wa = H[static_cast<size_t>(pi[a]) * n + l];
                if (wa != 0.0) v -= wa / H[static_cast<size_t>(l) * n + l] * H[static_cast<size_t>(pi[b]) * n + l];
            }
            S[static_cast<size_t>(a) * R + b] = v;
        }
    }
    for (int j = 0; j < R; j++) {   // LL^T in place (lower)
        double d = S[static_cast<size_t>(j) * R + j];
        for (int k = 0; k < j; k++) d -= S[static_cast<size_t>(j) * R + k] * S[static_cast<size_t>(j) * R + k];
        if (!(d > 0) || !std::isfinite(d)) return false;
        const double ljj = std::sqrt(d);
        S[static_cast<size_t>(j) * R + j] = ljj;
        for (int i = j + 1; i < R; i++) {
            double s = S[static_cast<size_t>(i) * R + j];
            for (int k = 0; k < j; k++) s -= S[static_cast<size_t>(i) * R + k] * S[static_cast<size_t>(j) * R + k];
            S[static_cast<size_t>(i) * R + j] = s / ljj;
        }
    }
    std::vector<double> yp(gr);
    for (int i = 0; i < R; i++) {
        double s = yp[i];
        for (int k = 0; k < i; k++) s -= S[static_cast<size_t>(i) * R + k] * yp[k];
        yp[i] = s / S[static_cast<size_t>(i) * R + i];
    }
    for (int i = R - 1; i >= 0; i--) {
        double s = yp[i];
        for (int k = i + 1; k < R; k++) s -= S[static_cast<size_t>(k) * R + i] * yp[k];
        yp[i] = s / S[static_cast<size_t>(i) * R + i];
    }
    y->assign(n, 0.0);
    for (int a = 0; a < R; a++) (*y)[pi[a]] = yp[a];
    for (int l : li) {
        double s = g[l];
        for (int a = 0; a < R; a++) s -= H[static_cast<size_t>(pi[a]) * n + l] * yp[a];
        (*y)[l] = s / H[static_cast<size_t>(l) * n + l];
    }
    for (double v : *y)
        if (!std::isfinite(v)) return false;
    return true;
}
}  // namespace

// ------------------------------------------------------------------------------------------ Solve
static void SolveImpl(const Solver::Options &opt, Problem *problem, Solver::Summary *summary) {
    *summary = Solver::Summary();
    Program prog(problem);
    const int n = prog.ncols, m = prog.nrows;
    summary->num_parameters_reduced = 0;
    for (int b : prog.active) summary->num_parameters_reduced += problem->blocks_[b].size;
    summary->num_effective_parameters_reduced = n;
    summary->num_residuals_reduced = m;
    auto writeback = [&](const std::vector<std::vector<double>> &x) {
        for (int b : prog.active) std::memcpy(problem->blocks_[b].user, x[b].data(), sizeof(double) * x[b].size());
    };
    if (n == 0 || m == 0) {
        summary->termination_type = CONVERGENCE;
        return;
    }
    // e-blocks of DENSE_SCHUR: a greedy independent set among the 1-wide blocks, cheapest (lowest degree) first
    std::vector<char> elim(n, 0);
    if (opt.linear_solver_type == DENSE_SCHUR) {
        const int nb = static_cast<int>(problem->blocks_.size());
        std::vector<int> degree(nb, 0);
        for (auto &r : problem->residuals_)
            for (int b : r.blocks) degree[b]++;
        std::vector<int> cand;
        for (int b : prog.active)
            if (problem->blocks_[b].local_size() == 1) cand.push_back(b);
        std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return degree[a] < degree[b]; });
        std::vector<char> chosen(nb, 0);
        std::vector<std::vector<int>> res_of(nb);
        for (size_t ri = 0; ri < problem->residuals_.size(); ri++)
            for (int b : problem->residuals_[ri].blocks) res_of[b].push_back(static_cast<int>(ri));
        for (int b : cand) {
            bool ok = true;
            for (int ri : res_of[b])
                for (int o : problem->residuals_[ri].blocks)
                    if (o != b && chosen[o]) ok = false;
            if (ok) {
                chosen[b] = 1;
                elim[prog.col[b]] = 1;
            }
        }
    }

    std::vector<std::vector<double>> x = prog.x, x_cand;
    double cost = 0;
    std::vector<double> r, J;
    if (!prog.evaluate(x, &cost, &r, &J)) {
        summary->termination_type = FAILURE;
        summary->message = "initial evaluation failed";
        return;
    }
    summary->initial_cost = summary->final_cost = cost;
    std::vector<double> scale(n, 1.0);
    if (opt.jacobi_scaling) {
        for (int c = 0; c < n; c++) {
            double s = 0;
            for (int i = 0; i < m; i++) s += J[static_cast<size_t>(i) * n + c] * J[static_cast<size_t>(i) * n + c];
            scale[c] = 1.0 / (1.0 + std::sqrt(s));
        }
    }
    auto scale_jacobian = [&]() {
        for (int i = 0; i < m; i++)
            for (int c = 0; c < n; c++) J[static_cast<size_t>(i) * n + c] *= scale[c];
    };
    scale_jacobian();
    Dense A{m, n, &J, {}};
    A.index();
    auto gradient = [&](std::vector<double> *g) {   // scaled-space J^T r
        g->assign(n, 0.0);
        for (int i = 0; i < m; i++)
            for (int c : A.nz[i]) (*g)[c] += A.at(i, c) * r[i];
    };
    auto grad_max_unscaled = [&](const std::vector<double> &g) {
        double mx = 0;
        for (int c = 0; c < n; c++) mx = std::max(mx, std::fabs(g[c] / scale[c]));
        return mx;
    };
    std::vector<double> g;
    gradient(&g);
    {
        IterationSummary it0;
        it0.iteration = 0;
        it0.cost = cost;
        it0.gradient_max_norm = grad_max_unscaled(g);
        it0.trust_region_radius = opt.initial_trust_region_radius;
        it0.step_is_valid = it0.step_is_successful = true;
        summary->iterations.push_back(it0);
    }
    if (grad_max_unscaled(g) <= opt.gradient_tolerance) {
        summary->termination_type = CONVERGENCE;
        summary->iterations.back().exit_reason = 3;
        return;
    }

    double radius = opt.initial_trust_region_radius;
    const double min_mu = 1e-8, max_mu = 1.0;
    double mu = min_mu;
    bool reuse = false;
    double x_norm = prog.ambient_norm(x, nullptr);
    int num_invalid = 0;
    std::vector<double> Dg(n), gt(n), gn(n), step(n), s(n), y;
    double alpha = 0, dogleg_norm = 0;
    int iter = 0;
    summary->termination_type = NO_CONVERGENCE;
    while (true) {
        if (iter >= opt.max_num_iterations) break;
        iter++;
        IterationSummary is;
        is.iteration = iter;
        is.cost = cost;
        is.trust_region_radius = radius;
        bool ok = true;
        double model_change = 0;
        if (!reuse) {   // DoglegStrategy::ComputeStep, fresh linearisation
            reuse = true;
            for (int c = 0; c < n; c++) Dg[c] = 0;
            for (int i = 0; i < m; i++)
                for (int c : A.nz[i]) Dg[c] += A.at(i, c) * A.at(i, c);
            for (int c = 0; c < n; c++) Dg[c] = std::sqrt(std::min(std::max(Dg[c], opt.min_lm_diagonal), opt.max_lm_diagonal));
            gradient(&g);
            for (int c = 0; c < n; c++) gt[c] = g[c] / Dg[c];
            double gtgt = 0, JgJg = 0;
            for (int c = 0; c < n; c++) gtgt += gt[c] * gt[c];
            for (int i = 0; i < m; i++) {
                double v = 0;
                for (int c : A.nz[i]) v += A.at(i, c) * (gt[c] / Dg[c]);
                JgJg += v * v;
            }
            alpha = gtgt / JgJg;
            bool solved = false;
            while (mu < max_mu) {
                std::vector<double> Dm(n);
                for (int c = 0; c < n; c++) Dm[c] = Dg[c] * std::sqrt(mu);
                if (dense_schur_solve(A, r, Dm, elim, &y)) {
                    solved = true;
                    break;
                }
                mu *= 10.0;
            }
            if (!solved) ok = false;
            else
                for (int c = 0; c < n; c++) gn[c] = -(y[c] * Dg[c]);
        }
        is.mu = mu;
        if (ok) {
            double gnorm = 0, gnn = 0, gtgn = 0;
            for (int c = 0; c < n; c++) {
                gnorm += gt[c] * gt[c];
                gnn += gn[c] * gn[c];
                gtgn += gt[c] * gn[c];
            }
            gnorm = std::sqrt(gnorm);
            gnn = std::sqrt(gnn);
            if (gnn <= radius) {
                s = gn;
                dogleg_norm = gnn;
            } else if (gnorm * alpha >= radius) {
                for (int c = 0; c < n; c++) s[c] = -(radius / gnorm) * gt[c];
                dogleg_norm = radius;
            } else {
                const double b_dot_a = -alpha * gtgn;
                const double a_sq = (alpha * gnorm) * (alpha * gnorm);
                const double bma_sq = a_sq - 2 * b_dot_a + gnn * gnn;
                const double c0 = b_dot_a - a_sq;
                const double d = std::sqrt(c0 * c0 + bma_sq * (radius * radius - a_sq));
                const double beta = (c0 <= 0) ? (d - c0) / bma_sq : (radius * radius - a_sq) / (d + c0);
                double nn = 0;
                for (int c = 0; c < n; c++) {
                    s[c] = (-alpha * (1.0 - beta)) * gt[c] + beta * gn[c];
                    nn += s[c] * s[c];
                }
                dogleg_norm = std::sqrt(nn);
            }
            for (int c = 0; c < n; c++) step[c] = s[c] / Dg[c];
            for (int i = 0; i < m; i++) {
                double v = 0;
                for (int c : A.nz[i]) v += A.at(i, c) * step[c];
                model_change += -v * (r[i] + v / 2.0);
            }
        }
        is.model_cost_change = model_change;
        is.step_norm = dogleg_norm;
        if (!ok || !(model_change > 0)) {   // invalid step
            num_invalid++;
            is.step_is_valid = false;
            summary->iterations.push_back(is);
            if (num_invalid >= opt.max_num_consecutive_invalid_steps) {
                summary->termination_type = FAILURE;
                break;
            }
            mu *= 10.0;
            reuse = false;
            continue;
        }
        num_invalid = 0;
        is.step_is_valid = true;
        std::vector<double> delta(n);
        for (int c = 0; c < n; c++) delta[c] = step[c] * scale[c];
        prog.plus(x, delta, &x_cand);
        double cost_cand = 0;
        if (!prog.evaluate(x_cand, &cost_cand, nullptr, nullptr)) cost_cand = std::numeric_limits<double>::max();
        is.candidate_cost = cost_cand;
        const double step_norm = prog.ambient_norm(x, &x_cand);
        if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) {
            is.exit_reason = 1;
            summary->iterations.push_back(is);
            summary->termination_type = CONVERGENCE;
            break;
        }
        is.cost_change = cost - cost_cand;
        if (std::fabs(cost - cost_cand) <= opt.function_tolerance * cost) {
            is.exit_reason = 2;
            summary->iterations.push_back(is);
            summary->termination_type = CONVERGENCE;
            break;
        }
        const double rho = (cost - cost_cand) / model_change;
        is.relative_decrease = rho;
        if (rho > opt.min_relative_decrease) {
            x = x_cand;
            x_norm = prog.ambient_norm(x, nullptr);
            prog.evaluate(x, &cost, &r, &J);
            scale_jacobian();
            A.index();
            is.step_is_successful = true;
            summary->num_successful_steps++;
            if (rho < 0.25) radius *= 0.5;
            if (rho > 0.75) radius = std::max(radius, 3.0 * dogleg_norm);
            mu = std::max(min_mu, 2.0 * mu / 10.0);
            reuse = false;
            gradient(&g);
            is.gradient_max_norm = grad_max_unscaled(g);
            summary->iterations.push_back(is);
            if (is.gradient_max_norm <= opt.gradient_tolerance) {
                summary->iterations.back().exit_reason = 3;
                summary->termination_type = CONVERGENCE;
                break;
            }
        } else {
            summary->num_unsuccessful_steps++;
            summary->iterations.push_back(is);
            radius *= 0.5;
            reuse = true;
        }
    }
    summary->final_cost = cost;
    writeback(x);
}

Solver::Summary vins_ref_last_summary;   // read by oracle/ref_stubs/ref_driver.cpp (Estimator::optimization keeps its summary local)
void Solve(const Solver::Options &opt, Problem *problem, Solver::Summary *summary) {
    SolveImpl(opt, problem, summary);
    // callbacks see Ceres' convention: `cost` is the cost of the point the iteration ENDS at (the candidate's if the step was
    // accepted), while the stand-in's own rows keep the cost the iteration STARTED from next to `candidate_cost`
    for (IterationCallback *cb : opt.callbacks)
        for (IterationSummary it : summary->iterations) {
            if (it.iteration > 0 && it.step_is_valid && it.step_is_successful) it.cost = it.candidate_cost;
            (*cb)(it);
        }
    vins_ref_last_summary = *summary;
}

}  // namespace ceres
