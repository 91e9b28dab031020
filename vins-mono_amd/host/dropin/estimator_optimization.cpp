// estimator_optimization.cpp — THE drop-in for the back end: `void Estimator::optimization()` of the REFERENCE class
// (vins_estimator/src/estimator.h:47, body at estimator.cpp:670-1003) on the MI355X path.
//
// How it is used.  This file includes the reference's own "estimator.h" and defines exactly one member function of the
// reference's `class Estimator`.  In a catkin workspace: add this file and include/vinsgpu.h to vins_estimator, link
// libvinsgpu.so, and remove (or `objcopy --weaken-symbol=_ZN9Estimator12optimizationEv`) the definition in estimator.cpp —
// nothing else of the package changes: processIMU / processImage / solveOdometry / slideWindow / vector2double /
// double2vector / FeatureManager / MarginalizationInfo are the reference's, and they call this body where they called
// Ceres.  Here (no Eigen / Ceres / ROS in the image) oracle/Makefile target `ref_gpu` does precisely that against the
// header stand-ins of oracle/ref_stubs and produces oracle/_ref/libvins_ref_gpu.so; tests/test_dropin_gpu.py drives the
// reference's processIMU / processImage loop through it and through the all-reference build side by side.
//
// What the body does instead of estimator.cpp:670-1003:
//   :672-701  ceres::Problem + parameter blocks      -> vector2double() (the reference's), then plain tables
//   :703-801  AddResidualBlock (prior, IMU, projection, relocalisation factors) -> vg_ba_problem (same selection rules)
//   :803-818  ceres::Solve(DENSE_SCHUR, DOGLEG)      -> vg_ba_optimize_begin: solve + gauge fix on the device, states back
//   :823      double2vector()                        -> the reference's own (its yaw / position fix is idempotent on the
//                                                       already fixed states; it also forms the relocalisation by-products)
//   :825-1000 MarginalizationInfo machinery          -> runs on the device BEHIND the state download; picked up at the next
//                                                       call (or by vins_gpu_collect_prior) and stored into the reference's
//                                                       own MarginalizationInfo / last_marginalization_parameter_blocks
// State the reference class has no member for (device handle, pending prior) lives in a side table keyed by `this`.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "estimator.h"
#include "vinsgpu.h"

namespace {

struct VgSide {
    vg_handle* vg = nullptr;
    bool prior_pending = false;   // optimization() has returned, its marginalization result is still on the device
    bool solver_failed = false;   // the device reported a non-finite solve
    // What the window looked like when the result was left on the device: the pending prior belongs to the window that
    // CONTINUES that one.  Both slides leave the newest frame of then at index WINDOW_SIZE - 1 (estimator.cpp:1037-1041,
    // :1078-1079), and optimization() itself is the only writer of last_marginalization_info besides clearState() (:70-76).
    double pending_stamp = 0;                               // Headers[WINDOW_SIZE].stamp at that time
    const MarginalizationInfo* pending_lmi = nullptr;       // last_marginalization_info at that time (not yet replaced)
    vg_ba_summary last;           // trace of the last solve (the reference only logs Summary::BriefReport)
    // run-time behaviour switches (vins_gpu_set_option)
    bool solver_time_cap = false; // forward SOLVER_TIME as max_solver_time_in_seconds (estimator.cpp:812-815)
    bool marg_eigen = false;      // the reference's eigen form of the prior instead of the pivoted-Cholesky square root
    bool marg_mode_dirty = false;
    bool imu_info_reference = false, imu_info_dirty = false;
};
std::mutex g_mu;
std::unordered_map<const Estimator*, VgSide> g_side;
VgSide& side_of(const Estimator* e) {
    std::lock_guard<std::mutex> lock(g_mu);
    return g_side[e];             // (references into an unordered_map stay valid across inserts)
}

bool block_of(Estimator& e, const double* addr, int& kind, int& index) {
    for (int i = 0; i <= WINDOW_SIZE; ++i) {
        if (addr == e.para_Pose[i]) { kind = VG_BLK_POSE; index = i; return true; }
        if (addr == e.para_SpeedBias[i]) { kind = VG_BLK_SPEEDBIAS; index = i; return true; }
    }
    if (addr == e.para_Ex_Pose[0]) { kind = VG_BLK_EXPOSE; index = 0; return true; }
    if (addr == e.para_Td[0]) { kind = VG_BLK_TD; index = 0; return true; }
    return false;
}

// The marginalization result of the last optimization() -> last_marginalization_info / ..._parameter_blocks
// (estimator.cpp:926-929 / :992-996), as the reference's own types.
void collect_prior(Estimator& e, VgSide& s) {
    if (!s.prior_pending) return;
    s.prior_pending = false;
    // (before the caller's slideWindow() the frame is still at WINDOW_SIZE; afterwards both slides leave it at WINDOW_SIZE - 1)
    const bool continues = e.Headers[WINDOW_SIZE - 1].stamp.toSec() == s.pending_stamp || e.Headers[WINDOW_SIZE].stamp.toSec() == s.pending_stamp;
    if (e.last_marginalization_info != s.pending_lmi || !continues) {
        // clearState() ran in between (failureDetection() in processImage, estimator.cpp:190-200, or the restart callback,
        // estimator_node.cpp:182-198): it freed last_marginalization_info and the window now on the host is a fresh one.  The
        // result on the device describes the trajectory before the reset -- the reference has no prior here, so neither do we.
        // (Nothing to fetch: the next upload is ordered behind the marginalization kernel on the handle's stream.)
        return;
    }
    const int K = WINDOW_SIZE + 1;
    const int cap = 6 * K + 32, capb = K + 8;
    std::vector<int> nkind(capb), nindex(capb);
    std::vector<double> nJ0((size_t)cap * cap), nr0(cap), nx0(9 * capb);
    vg_ba_prior pr;
    memset(&pr, 0, sizeof(pr));
    pr.cap = cap; pr.cap_blocks = capb; pr.block_kind = nkind.data(); pr.block_index = nindex.data();
    pr.J0 = nJ0.data(); pr.r0 = nr0.data(); pr.x0 = nx0.data();
    if (vg_ba_optimize_prior(s.vg, &pr) != VG_OK) throw std::runtime_error(std::string("vg_ba_optimize_prior: ") + vg_last_error(s.vg));
    if (s.solver_failed) {
        // no new prior exists, and the old one must not survive the caller's slideWindow() (it names un-shifted blocks)
        if (e.last_marginalization_info) delete e.last_marginalization_info;
        e.last_marginalization_info = nullptr;
        e.last_marginalization_parameter_blocks.clear();
        return;
    }
    if (!pr.valid) return;        // MARGIN_SECOND_NEW without Pose[WINDOW_SIZE-1] in the prior: the old prior stays (:935-936)
    MarginalizationInfo* mi = new MarginalizationInfo();
    mi->n = pr.n; mi->m = pr.m;
    mi->linearized_jacobians.resize(pr.n, pr.n);
    mi->linearized_residuals.resize(pr.n);
    for (int r = 0; r < pr.n; ++r) {
        mi->linearized_residuals(r) = nr0[r];
        for (int c = 0; c < pr.n; ++c) mi->linearized_jacobians(r, c) = nJ0[(size_t)r * pr.n + c];
    }
    std::vector<double*> blocks;
    int off = 0, x0o = 0;
    for (int b = 0; b < pr.nblocks; ++b) {
        const int kind = nkind[b], idx = nindex[b];
        const int gs = kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 7), ls = kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 6);
        double* d = new double[gs];
        memcpy(d, nx0.data() + x0o, sizeof(double) * gs);
        // the block's address in the SLID window (addr_shift, :913-924 / :969-990: the device already re-labelled it)
        double* addr = kind == VG_BLK_POSE ? e.para_Pose[idx] : kind == VG_BLK_SPEEDBIAS ? e.para_SpeedBias[idx] : kind == VG_BLK_EXPOSE ? e.para_Ex_Pose[0] : e.para_Td[0];
        mi->parameter_block_data[reinterpret_cast<long>(addr)] = d;      // ~MarginalizationInfo frees it (marginalization_factor.cpp:71-87)
        mi->parameter_block_size[reinterpret_cast<long>(addr)] = gs;
        mi->parameter_block_idx[reinterpret_cast<long>(addr)] = pr.m + off;
        mi->keep_block_size.push_back(gs);
        mi->keep_block_idx.push_back(pr.m + off);
        mi->keep_block_data.push_back(d);
        blocks.push_back(addr);
        off += ls; x0o += gs;
    }
    mi->sum_block_size = x0o;
    if (e.last_marginalization_info) delete e.last_marginalization_info;
    e.last_marginalization_info = mi;
    e.last_marginalization_parameter_blocks = blocks;
}

}  // namespace

void Estimator::optimization() {
    VgSide& s = side_of(this);
    if (vg_abi_version() != VG_ABI_VERSION) throw std::runtime_error("libvinsgpu.so was built from another include/vinsgpu.h (ABI version mismatch)");
    if (!s.vg) {
        if (vg_create(&s.vg) != VG_OK) throw std::runtime_error("vg_create failed: no MI355X / libvinsgpu (no CPU fallback)");
    }
    // form of the prior factor the marginalization hands to the next frame: the pivoted-Cholesky square root (default) or the
    // reference's eigen form (marginalization_factor.cpp:285-296) -- vins_gpu_set_option(e, VINS_GPU_OPT_MARG_EIGEN, 1); include/vinsgpu.h
    if (s.marg_mode_dirty) { vg_ba_set_marg_mode(s.vg, s.marg_eigen ? VG_MARG_EIGEN : VG_MARG_SQRT); s.marg_mode_dirty = false; }
    if (s.imu_info_dirty) { vg_ba_set_imu_info_mode(s.vg, s.imu_info_reference ? VG_IMU_INFO_REFERENCE : VG_IMU_INFO_FACTOR); s.imu_info_dirty = false; }
    collect_prior(*this, s);                                    // the previous frame's marginalization result, if still on the device
    vector2double();                                            // estimator.cpp:701
    const int K = WINDOW_SIZE + 1;
    // ---- factor tables instead of problem.AddResidualBlock (:719-764), same filter
    std::vector<int> lm_start, lm_nobs, lm_off;
    std::vector<double> obs;
    for (auto& it : f_manager.feature) {
        it.used_num = it.feature_per_frame.size();
        if (!(it.used_num >= 2 && it.start_frame < WINDOW_SIZE - 2)) continue;
        lm_start.push_back(it.start_frame);
        lm_nobs.push_back((int)it.feature_per_frame.size());
        lm_off.push_back((int)obs.size() / 7);
        for (auto& f : it.feature_per_frame) {
            const double row[7] = {f.point.x(), f.point.y(), f.uv.x(), f.uv.y(), f.velocity.x(), f.velocity.y(), f.cur_td};
            obs.insert(obs.end(), row, row + 7);
        }
    }
    const int L = (int)lm_start.size();
    // ---- relocalisation factors (:769-801)
    std::vector<int> relo_lm;
    std::vector<double> relo_xy;
    if (relocalization_info) {
        size_t retrive = 0;
        int feature_index = -1;
        for (auto& it : f_manager.feature) {
            if (!(it.used_num >= 2 && it.start_frame < WINDOW_SIZE - 2)) continue;
            ++feature_index;
            if (it.start_frame > relo_frame_local_index) continue;
            while (retrive < match_points.size() && (int)match_points[retrive].z() < it.feature_id) ++retrive;
            if (retrive < match_points.size() && (int)match_points[retrive].z() == it.feature_id) {
                relo_lm.push_back(feature_index);
                relo_xy.push_back(match_points[retrive].x());
                relo_xy.push_back(match_points[retrive].y());
                ++retrive;
            }
        }
    }
    // ---- IMUFactor(pre_integrations[j]), j = i + 1 (:711-718)
    std::vector<vg_imu_preint> imu(K - 1);
    for (int i = 0; i < WINDOW_SIZE; i++) {
        const IntegrationBase* p = pre_integrations[i + 1];
        vg_imu_preint& m = imu[i];
        memset(&m, 0, sizeof(m));
        if (!p) continue;
        m.valid = 1; m.sum_dt = p->sum_dt;
        for (int k = 0; k < 3; ++k) { m.delta_p[k] = p->delta_p(k); m.delta_v[k] = p->delta_v(k); m.linearized_ba[k] = p->linearized_ba(k); m.linearized_bg[k] = p->linearized_bg(k); }
        m.delta_q[0] = p->delta_q.x(); m.delta_q[1] = p->delta_q.y(); m.delta_q[2] = p->delta_q.z(); m.delta_q[3] = p->delta_q.w();
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c) { m.jacobian[r * 15 + c] = p->jacobian(r, c); m.covariance[r * 15 + c] = p->covariance(r, c); }
    }
    vg_ba_problem pb;
    memset(&pb, 0, sizeof(pb));
    pb.K = K; pb.L = L; pb.n_obs = (int)obs.size() / 7;
    pb.pose = &para_Pose[0][0]; pb.speedbias = &para_SpeedBias[0][0]; pb.ex_pose = &para_Ex_Pose[0][0]; pb.td = ESTIMATE_TD ? para_Td[0][0] : td;
    pb.inv_depth = &para_Feature[0][0];
    pb.lm_start = lm_start.data(); pb.lm_nobs = lm_nobs.data(); pb.lm_obs_off = lm_off.data(); pb.obs = obs.data(); pb.imu = imu.data();
    // ---- MarginalizationFactor(last_marginalization_info) on last_marginalization_parameter_blocks (:703-709)
    std::vector<int> pkind, pindex;
    std::vector<double> pJ0, pr0v, px0;
    if (last_marginalization_info) {
        MarginalizationInfo* mi = last_marginalization_info;
        const int n = mi->n, nb = (int)mi->keep_block_size.size();
        const int ncols = (int)mi->linearized_jacobians.cols();
        // columns of linearized_jacobians are addressed through keep_block_idx - m (marginalization_factor.cpp:343-378): gather
        // them into the block order of last_marginalization_parameter_blocks
        pJ0.assign((size_t)n * n, 0.0);
        int col = 0;
        for (int b = 0; b < nb; ++b) {
            int kind, index;
            if (!block_of(*this, last_marginalization_parameter_blocks[b], kind, index)) throw std::runtime_error("prior block address outside the para_* arrays");
            pkind.push_back(kind); pindex.push_back(index);
            const int gs = mi->keep_block_size[b], ls = mi->localSize(gs), src = mi->keep_block_idx[b] - mi->m;
            px0.insert(px0.end(), mi->keep_block_data[b], mi->keep_block_data[b] + gs);
            if (src < 0 || src + ls > ncols || col + ls > n) throw std::runtime_error("prior block outside linearized_jacobians");
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < ls; ++c) pJ0[(size_t)r * n + col + c] = mi->linearized_jacobians(r, src + c);
            col += ls;
        }
        if (col != n) throw std::runtime_error("prior: kept blocks do not cover the n columns");
        pr0v.resize(n);
        for (int r = 0; r < n; ++r) pr0v[r] = mi->linearized_residuals(r);
        pb.prior_n = n; pb.prior_nblocks = nb;
        pb.prior_block_kind = pkind.data(); pb.prior_block_index = pindex.data();
        pb.prior_J0 = pJ0.data(); pb.prior_r0 = pr0v.data(); pb.prior_x0 = px0.data();
    }
    const bool relo_in_problem = relocalization_info && !relo_lm.empty();
    if (relo_in_problem) {
        pb.relo_n = (int)relo_lm.size(); pb.relo_pose = relo_Pose; pb.relo_lm = relo_lm.data(); pb.relo_xy = relo_xy.data();
    }
    pb.estimate_extrinsic = ESTIMATE_EXTRINSIC ? 1 : 0; pb.estimate_td = ESTIMATE_TD ? 1 : 0; pb.max_iters = NUM_ITERATIONS;
    pb.focal = FOCAL_LENGTH; pb.tr = TR; pb.row = ROW; pb.g_norm = G.z();
    // options.max_solver_time_in_seconds (:812-815): a wall-clock cap makes the result depend on timing (oracle/ASSUMPTIONS.md C7)
    // and 40 ms is 30 x what a solve takes here, so it is NOT forwarded unless the caller asks for the reference's contract to the
    // letter: vins_gpu_set_option(e, VINS_GPU_OPT_SOLVER_TIME_CAP, 1) (a run-time switch)
    if (s.solver_time_cap) pb.max_solver_time_s = marginalization_flag == MARGIN_OLD ? SOLVER_TIME * 4.0 / 5.0 : SOLVER_TIME;
    // ---- outputs
    std::vector<double> lam(L > 0 ? L : 1);
    double td_out = td;
    vg_ba_state st;
    st.pose = &para_Pose[0][0]; st.speedbias = &para_SpeedBias[0][0]; st.ex_pose = &para_Ex_Pose[0][0]; st.td = ESTIMATE_TD ? &para_Td[0][0] : &td_out;
    st.inv_depth = lam.data(); st.relo_pose = pb.relo_n ? relo_Pose : nullptr;
    const int flag = marginalization_flag == MARGIN_OLD ? VG_MARGIN_OLD : VG_MARGIN_SECOND_NEW;
    const int rc = vg_ba_optimize_begin(s.vg, &pb, flag, &st, &s.last);
    if (rc != VG_OK && rc != VG_ERR_NUMERIC) throw std::runtime_error(std::string("vg_ba_optimize_begin: ") + vg_last_error(s.vg));
    for (int l = 0; l < L; ++l) para_Feature[l][0] = lam[l];
    if (relocalization_info && !relo_in_problem) {
        // relo_Pose carried no factor: the reference leaves it untouched by the solve and gauge-fixes it in double2vector
        // (:598-603) with the rot_diff / para_Pose[0] it derives there; the device has applied that transform to the window
        // already, so it is applied to relo_Pose here (vg_ba_summary::gauge_*) and double2vector's own transform is the identity
        Matrix3d rot;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) rot(r, c) = s.last.gauge_rot[3 * r + c];
        const Matrix3d Rr = rot * Quaterniond(relo_Pose[6], relo_Pose[3], relo_Pose[4], relo_Pose[5]).normalized().toRotationMatrix();
        const Vector3d tr = rot * Vector3d(relo_Pose[0] - s.last.gauge_p0[0], relo_Pose[1] - s.last.gauge_p0[1], relo_Pose[2] - s.last.gauge_p0[2])
                            + Vector3d(para_Pose[0][0], para_Pose[0][1], para_Pose[0][2]);
        const Quaterniond q(Rr);
        relo_Pose[0] = tr.x(); relo_Pose[1] = tr.y(); relo_Pose[2] = tr.z();
        relo_Pose[3] = q.x(); relo_Pose[4] = q.y(); relo_Pose[5] = q.z(); relo_Pose[6] = q.w();
    }
    double2vector();                                            // :823 (the reference's)
    s.solver_failed = rc == VG_ERR_NUMERIC;
    s.prior_pending = true;
    s.pending_stamp = Headers[WINDOW_SIZE].stamp.toSec();
    s.pending_lmi = last_marginalization_info;
    if (s.solver_failed) collect_prior(*this, s);              // (drops the priors, see there)
}

// ---- C entry points for the surrounding code (the reference class has no member to hang these on)
extern "C" {
// fetch the marginalization result of the last optimization() now (tests, serialisation of last_marginalization_info)
void vins_gpu_collect_prior(Estimator* e) { collect_prior(*e, side_of(e)); }
// forget a pending marginalization result: for callers that reset the estimator by other means than clearState()
void vins_gpu_reset(Estimator* e) {
    VgSide& s = side_of(e);
    s.prior_pending = false; s.solver_failed = false;
}
// behaviour switches of one Estimator (run time, no environment variables): VINS_GPU_OPT_SOLVER_TIME_CAP forwards SOLVER_TIME as
// max_solver_time_in_seconds (estimator.cpp:812-815; default off), VINS_GPU_OPT_MARG_EIGEN asks for the reference's eigen form of
// the prior instead of the square root (default off), VINS_GPU_OPT_IMU_INFO_REFERENCE (3) for the IMU factors' sqrt_info formed as
// imu_factor.h:64 spells it, inverse() then LLT (vg_ba_set_imu_info_mode; default off).  Takes effect at the next optimization().  Returns 0, -1 for an unknown option.
int vins_gpu_set_option(Estimator* e, int option, int value) {
    VgSide& s = side_of(e);
    if (option == 1) { s.solver_time_cap = value != 0; return 0; }                                       // VINS_GPU_OPT_SOLVER_TIME_CAP
    if (option == 2) { s.marg_eigen = value != 0; s.marg_mode_dirty = true; return 0; }                  // VINS_GPU_OPT_MARG_EIGEN
    if (option == 3) { s.imu_info_reference = value != 0; s.imu_info_dirty = true; return 0; }           // VINS_GPU_OPT_IMU_INFO_REFERENCE
    return -1;
}
// trace of the last solve
const vg_ba_summary* vins_gpu_last_summary(Estimator* e) { return &side_of(e).last; }
int vins_gpu_last_iterations(Estimator* e) { return side_of(e).last.num_iterations; }
// release the device handle of an Estimator that is about to be destroyed
void vins_gpu_release(Estimator* e) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_side.find(e);
    if (it == g_side.end()) return;
    if (it->second.vg) vg_destroy(it->second.vg);
    g_side.erase(it);
}
}
