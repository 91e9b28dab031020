// host_test_api.cpp — C entry point used by tests/ to drive the drop-in Estimator from a vg_ba_problem:
// it fills the Estimator members the way the surrounding reference code would have left them (Ps/Rs/..., the
// f_manager.feature list incl. features the filter must skip, pre_integrations[], last_marginalization_info), calls
// Estimator::optimization() and returns the members it wrote.  Test plumbing only.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include "estimator.h"
#include "feature_tracker.h"

extern "C" int vins_host_estimator_roundtrip(const vg_ba_problem* p, int margin_flag, double* pose_out /*K*7*/, double* sb_out /*K*9*/,
                                             double* depth_out /*L*/, int* solve_flag_out /*L*/, int* prior_n, int* prior_nblocks,
                                             int* prior_kind, int* prior_index, double* prior_J0, double* prior_r0, int* iters) {
    if (p->K != WINDOW_SIZE + 1) return -1;
    Estimator est;
    ESTIMATE_EXTRINSIC = p->estimate_extrinsic; ESTIMATE_TD = p->estimate_td; NUM_ITERATIONS = p->max_iters;
    TR = p->tr; ROW_D = p->row; FOCAL_LENGTH_D = p->focal; G_NORM = p->g_norm;
    for (int i = 0; i <= WINDOW_SIZE; ++i) {
        est.Ps[i] = Vector3d(p->pose[7 * i], p->pose[7 * i + 1], p->pose[7 * i + 2]);
        est.Rs[i] = Quaterniond(p->pose[7 * i + 6], p->pose[7 * i + 3], p->pose[7 * i + 4], p->pose[7 * i + 5]).toRotationMatrix();
        est.Vs[i] = Vector3d(p->speedbias[9 * i], p->speedbias[9 * i + 1], p->speedbias[9 * i + 2]);
        est.Bas[i] = Vector3d(p->speedbias[9 * i + 3], p->speedbias[9 * i + 4], p->speedbias[9 * i + 5]);
        est.Bgs[i] = Vector3d(p->speedbias[9 * i + 6], p->speedbias[9 * i + 7], p->speedbias[9 * i + 8]);
    }
    est.tic[0] = Vector3d(p->ex_pose[0], p->ex_pose[1], p->ex_pose[2]);
    est.ric[0] = Quaterniond(p->ex_pose[6], p->ex_pose[3], p->ex_pose[4], p->ex_pose[5]).toRotationMatrix();
    est.td = p->td;
    IntegrationBase pre[WINDOW_SIZE + 1];
    for (int k = 0; k < WINDOW_SIZE; ++k) {
        const vg_imu_preint& m = p->imu[k];
        IntegrationBase& q = pre[k + 1];
        q.sum_dt = m.sum_dt;
        q.delta_p = Vector3d(m.delta_p[0], m.delta_p[1], m.delta_p[2]); q.delta_v = Vector3d(m.delta_v[0], m.delta_v[1], m.delta_v[2]);
        q.linearized_ba = Vector3d(m.linearized_ba[0], m.linearized_ba[1], m.linearized_ba[2]);
        q.linearized_bg = Vector3d(m.linearized_bg[0], m.linearized_bg[1], m.linearized_bg[2]);
        q.delta_q = Quaterniond(m.delta_q[3], m.delta_q[0], m.delta_q[1], m.delta_q[2]);
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c) { q.jacobian(r, c) = m.jacobian[r * 15 + c]; q.covariance(r, c) = m.covariance[r * 15 + c]; }
        est.pre_integrations[k + 1] = m.valid ? &q : nullptr;
    }
    // features: the real ones interleaved with decoys the used_num / start_frame filter must drop
    for (int l = 0; l < p->L; ++l) {
        FeaturePerId f;
        f.feature_id = l; f.start_frame = p->lm_start[l]; f.estimated_depth = 1.0 / p->inv_depth[l];
        for (int k = 0; k < p->lm_nobs[l]; ++k) {
            const double* o = p->obs + 7 * (p->lm_obs_off[l] + k);
            FeaturePerFrame fr;
            fr.point = Vector3d(o[0], o[1], 1.0); fr.uv.x() = o[2]; fr.uv.y() = o[3]; fr.velocity.x() = o[4]; fr.velocity.y() = o[5]; fr.cur_td = o[6];
            f.feature_per_frame.push_back(fr);
        }
        est.f_manager.feature.push_back(f);
        if (l % 3 == 0) {       // decoys: a single-observation track and a track starting too late
            FeaturePerId d1; d1.feature_id = 100000 + l; d1.start_frame = 2; d1.estimated_depth = 5.0; d1.feature_per_frame.push_back(f.feature_per_frame[0]);
            est.f_manager.feature.push_back(d1);
            FeaturePerId d2 = f; d2.feature_id = 200000 + l; d2.start_frame = WINDOW_SIZE - 2; d2.feature_per_frame.resize(2);
            est.f_manager.feature.push_back(d2);
        }
    }
    if (p->prior_n > 0) {
        // a prior as the previous optimization() would have left it: reference-typed members + block addresses
        MarginalizationInfo* mi = new MarginalizationInfo();
        mi->n = p->prior_n; mi->m = 0;
        int off = 0, x0o = 0;
        for (int b = 0; b < p->prior_nblocks; ++b) {
            const int kind = p->prior_block_kind[b], idx = p->prior_block_index[b];
            const int gs = kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 7), ls = kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 6);
            mi->keep_block_size.push_back(gs);
            mi->keep_block_idx.push_back(off);
            double* d = new double[gs];
            memcpy(d, p->prior_x0 + x0o, sizeof(double) * gs);
            mi->keep_block_data.push_back(d);
            est.last_marginalization_parameter_blocks.push_back(kind == VG_BLK_POSE ? est.para_Pose[idx] : kind == VG_BLK_SPEEDBIAS ? est.para_SpeedBias[idx]
                                                                : kind == VG_BLK_EXPOSE ? est.para_Ex_Pose[0] : est.para_Td[0]);
            off += ls; x0o += gs;
        }
        mi->linearized_jacobians.resize(p->prior_n, p->prior_n);
        mi->linearized_residuals.resize(p->prior_n);
        for (int r = 0; r < p->prior_n; ++r) {
            mi->linearized_residuals(r) = p->prior_r0[r];
            for (int c = 0; c < p->prior_n; ++c) mi->linearized_jacobians(r, c) = p->prior_J0[(size_t)r * p->prior_n + c];
        }
        est.last_marginalization_info = mi;
    }
    est.marginalization_flag = margin_flag == VG_MARGIN_OLD ? Estimator::MARGIN_OLD : Estimator::MARGIN_SECOND_NEW;
    est.optimization();
    est.collectPrior();
    for (int i = 0; i <= WINDOW_SIZE; ++i) {
        Quaterniond q(est.Rs[i]);
        const double row[7] = {est.Ps[i].x(), est.Ps[i].y(), est.Ps[i].z(), q.x(), q.y(), q.z(), q.w()};
        memcpy(pose_out + 7 * i, row, sizeof(row));
        const double sb[9] = {est.Vs[i].x(), est.Vs[i].y(), est.Vs[i].z(), est.Bas[i].x(), est.Bas[i].y(), est.Bas[i].z(), est.Bgs[i].x(), est.Bgs[i].y(), est.Bgs[i].z()};
        memcpy(sb_out + 9 * i, sb, sizeof(sb));
    }
    int l = 0;
    for (auto& f : est.f_manager.feature) {
        if (f.feature_id >= 100000) continue;
        depth_out[l] = f.estimated_depth; solve_flag_out[l] = f.solve_flag; ++l;
    }
    *iters = est.last_summary.num_iterations;
    *prior_n = 0; *prior_nblocks = 0;
    if (est.last_marginalization_info) {
        MarginalizationInfo* mi = est.last_marginalization_info;
        *prior_n = mi->n; *prior_nblocks = (int)mi->keep_block_size.size();
        for (int b = 0; b < *prior_nblocks; ++b) {
            int kind = -1, index = -1;
            est.block_of(est.last_marginalization_parameter_blocks[b], kind, index);
            prior_kind[b] = kind; prior_index[b] = index;
        }
        for (int r = 0; r < mi->n; ++r) {
            prior_r0[r] = mi->linearized_residuals(r);
            for (int c = 0; c < mi->n; ++c) prior_J0[(size_t)r * mi->n + c] = mi->linearized_jacobians(r, c);
        }
    }
    return 0;
}


// The same round trip with relocalisation (estimator.cpp:769-801 + :596-616): p->relo_* become match_points / relo_Pose the
// way setReloFrame() would leave them; outputs = the by-products double2vector() computes + the gauge-fixed loop pose.
static int relo_roundtrip(const vg_ba_problem* p, int relo_frame_local_index, const double* prev_relo_t, const double* prev_relo_r, double* pose_out,
                          double* relo_fixed, double* relative_t, double* relative_q, double* relative_yaw, double* drift_r, double* drift_t, bool no_match);
extern "C" int vins_host_estimator_relo_roundtrip(const vg_ba_problem* p, int relo_frame_local_index, const double* prev_relo_t /*3*/,
                                                  const double* prev_relo_r /*9 row-major*/, double* pose_out /*K*7*/,
                                                  double* relo_fixed /*7: t, q(x y z w)*/, double* relative_t /*3*/, double* relative_q /*4 x y z w*/,
                                                  double* relative_yaw, double* drift_r /*9 row-major*/, double* drift_t /*3*/) {
    return relo_roundtrip(p, relo_frame_local_index, prev_relo_t, prev_relo_r, pose_out, relo_fixed, relative_t, relative_q, relative_yaw, drift_r, drift_t, false);
}
// the same with match_points that name no feature of the window: relo_Pose carries no factor (estimator.cpp:771-772 still adds the
// block), the by-products of :596-616 must come out of the gauge transform alone; relo_fixed returns relo_r / relo_t as used there
extern "C" int vins_host_estimator_relo_nomatch_roundtrip(const vg_ba_problem* p, int relo_frame_local_index, const double* prev_relo_t,
                                                          const double* prev_relo_r, double* pose_out, double* relo_fixed, double* relative_t,
                                                          double* relative_q, double* relative_yaw, double* drift_r, double* drift_t) {
    return relo_roundtrip(p, relo_frame_local_index, prev_relo_t, prev_relo_r, pose_out, relo_fixed, relative_t, relative_q, relative_yaw, drift_r, drift_t, true);
}
static int relo_roundtrip(const vg_ba_problem* p, int relo_frame_local_index, const double* prev_relo_t, const double* prev_relo_r, double* pose_out,
                          double* relo_fixed, double* relative_t, double* relative_q, double* relative_yaw, double* drift_r, double* drift_t, bool no_match) {
    if (p->K != WINDOW_SIZE + 1 || p->relo_n <= 0) return -1;
    Estimator est;
    ESTIMATE_EXTRINSIC = p->estimate_extrinsic; ESTIMATE_TD = p->estimate_td; NUM_ITERATIONS = p->max_iters;
    TR = p->tr; ROW_D = p->row; FOCAL_LENGTH_D = p->focal; G_NORM = p->g_norm;
    for (int i = 0; i <= WINDOW_SIZE; ++i) {
        est.Ps[i] = Vector3d(p->pose[7 * i], p->pose[7 * i + 1], p->pose[7 * i + 2]);
        est.Rs[i] = Quaterniond(p->pose[7 * i + 6], p->pose[7 * i + 3], p->pose[7 * i + 4], p->pose[7 * i + 5]).toRotationMatrix();
        est.Vs[i] = Vector3d(p->speedbias[9 * i], p->speedbias[9 * i + 1], p->speedbias[9 * i + 2]);
        est.Bas[i] = Vector3d(p->speedbias[9 * i + 3], p->speedbias[9 * i + 4], p->speedbias[9 * i + 5]);
        est.Bgs[i] = Vector3d(p->speedbias[9 * i + 6], p->speedbias[9 * i + 7], p->speedbias[9 * i + 8]);
    }
    est.tic[0] = Vector3d(p->ex_pose[0], p->ex_pose[1], p->ex_pose[2]);
    est.ric[0] = Quaterniond(p->ex_pose[6], p->ex_pose[3], p->ex_pose[4], p->ex_pose[5]).toRotationMatrix();
    est.td = p->td;
    IntegrationBase pre[WINDOW_SIZE + 1];
    for (int k = 0; k < WINDOW_SIZE; ++k) {
        const vg_imu_preint& m = p->imu[k];
        IntegrationBase& q = pre[k + 1];
        q.sum_dt = m.sum_dt;
        q.delta_p = Vector3d(m.delta_p[0], m.delta_p[1], m.delta_p[2]); q.delta_v = Vector3d(m.delta_v[0], m.delta_v[1], m.delta_v[2]);
        q.linearized_ba = Vector3d(m.linearized_ba[0], m.linearized_ba[1], m.linearized_ba[2]);
        q.linearized_bg = Vector3d(m.linearized_bg[0], m.linearized_bg[1], m.linearized_bg[2]);
        q.delta_q = Quaterniond(m.delta_q[3], m.delta_q[0], m.delta_q[1], m.delta_q[2]);
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c) { q.jacobian(r, c) = m.jacobian[r * 15 + c]; q.covariance(r, c) = m.covariance[r * 15 + c]; }
        est.pre_integrations[k + 1] = m.valid ? &q : nullptr;
    }
    for (int l = 0; l < p->L; ++l) {
        FeaturePerId f;
        f.feature_id = 10 * l + 3; f.start_frame = p->lm_start[l]; f.estimated_depth = 1.0 / p->inv_depth[l];
        for (int k = 0; k < p->lm_nobs[l]; ++k) {
            const double* o = p->obs + 7 * (p->lm_obs_off[l] + k);
            FeaturePerFrame fr;
            fr.point = Vector3d(o[0], o[1], 1.0); fr.uv.x() = o[2]; fr.uv.y() = o[3]; fr.velocity.x() = o[4]; fr.velocity.y() = o[5]; fr.cur_td = o[6];
            f.feature_per_frame.push_back(fr);
        }
        est.f_manager.feature.push_back(f);
    }
    // setReloFrame (estimator.cpp:1128-1148): matches as (x, y, feature id) ascending by id, the loop pose, its local index
    est.relocalization_info = true;
    est.relo_frame_local_index = relo_frame_local_index;
    for (int k = 0; k < p->relo_n; ++k) est.match_points.push_back(Vector3d(p->relo_xy[2 * k], p->relo_xy[2 * k + 1], 10.0 * p->relo_lm[k] + (no_match ? 5 : 3)));
    est.match_points.push_back(Vector3d(0, 0, 10.0 * p->L + 7));        // a match of a feature that is not in the window any more
    for (int k = 0; k < 7; ++k) est.relo_Pose[k] = p->relo_pose[k];
    est.prev_relo_t = Vector3d(prev_relo_t[0], prev_relo_t[1], prev_relo_t[2]);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) est.prev_relo_r(r, c) = prev_relo_r[3 * r + c];
    est.marginalization_flag = Estimator::MARGIN_OLD;
    est.optimization();
    est.collectPrior();
    if (est.relocalization_info) return -2;                               // double2vector() must have consumed it
    for (int i = 0; i <= WINDOW_SIZE; ++i) {
        Quaterniond q(est.Rs[i]);
        const double row[7] = {est.Ps[i].x(), est.Ps[i].y(), est.Ps[i].z(), q.x(), q.y(), q.z(), q.w()};
        memcpy(pose_out + 7 * i, row, sizeof(row));
    }
    for (int k = 0; k < 7; ++k) relo_fixed[k] = est.relo_Pose[k];
    if (no_match) {
        Quaterniond q(est.relo_r_fixed);
        const double f[7] = {est.relo_t_fixed.x(), est.relo_t_fixed.y(), est.relo_t_fixed.z(), q.x(), q.y(), q.z(), q.w()};
        memcpy(relo_fixed, f, sizeof(f));
    }
    relative_t[0] = est.relo_relative_t.x(); relative_t[1] = est.relo_relative_t.y(); relative_t[2] = est.relo_relative_t.z();
    relative_q[0] = est.relo_relative_q.x(); relative_q[1] = est.relo_relative_q.y(); relative_q[2] = est.relo_relative_q.z(); relative_q[3] = est.relo_relative_q.w();
    *relative_yaw = est.relo_relative_yaw;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) drift_r[3 * r + c] = est.drift_correct_r(r, c);
    drift_t[0] = est.drift_correct_t.x(); drift_t[1] = est.drift_correct_t.y(); drift_t[2] = est.drift_correct_t.z();
    return 0;
}


// Estimator::slideWindow() with MARGIN_SECOND_NEW (estimator.cpp:1069-1099): the interval WINDOW_SIZE-1 -> WINDOW_SIZE is folded
// into the interval that ends at frame WINDOW_SIZE-1.  Inputs: the two intervals as raw samples ([dt acc gyr] rows, first
// measurement [acc gyr], linearisation biases [ba bg]) and the states of the two newest frames (pose7 + sb9 each).
// Outputs: the merged pre-integration, the state now in slot WINDOW_SIZE-1, whether slot WINDOW_SIZE was released.
extern "C" int vins_host_slide_second_new(int na, const double* smp_a, const double* first_a, int nb, const double* smp_b, const double* first_b,
                                          const double* bias, const double* noise4 /*acc_n gyr_n acc_w gyr_w*/, const double* state_prev /*16*/,
                                          const double* state_new /*16*/, vg_imu_preint* merged, double* state_out /*16*/, int* slot_ws_released,
                                          int* nsamples_out) {
    try {
        Estimator est;
        ACC_N = noise4[0]; GYR_N = noise4[1]; ACC_W = noise4[2]; GYR_W = noise4[3];
        auto fill = [&](int n, const double* smp, const double* first) {
            IntegrationBase* p = new IntegrationBase();
            p->linearized_acc = Vector3d(first[0], first[1], first[2]);
            p->linearized_gyr = Vector3d(first[3], first[4], first[5]);
            p->linearized_ba = Vector3d(bias[0], bias[1], bias[2]);
            p->linearized_bg = Vector3d(bias[3], bias[4], bias[5]);
            for (int i = 0; i < n; ++i) {
                p->dt_buf.push_back(smp[7 * i]);
                p->acc_buf.push_back(Vector3d(smp[7 * i + 1], smp[7 * i + 2], smp[7 * i + 3]));
                p->gyr_buf.push_back(Vector3d(smp[7 * i + 4], smp[7 * i + 5], smp[7 * i + 6]));
            }
            est.repropagate(p);
            return p;
        };
        auto set = [&](int i, const double* s) {
            est.Ps[i] = Vector3d(s[0], s[1], s[2]);
            est.Rs[i] = Quaterniond(s[6], s[3], s[4], s[5]).toRotationMatrix();
            est.Vs[i] = Vector3d(s[7], s[8], s[9]); est.Bas[i] = Vector3d(s[10], s[11], s[12]); est.Bgs[i] = Vector3d(s[13], s[14], s[15]);
        };
        est.pre_integrations[WINDOW_SIZE - 1] = fill(na, smp_a, first_a);
        est.pre_integrations[WINDOW_SIZE] = fill(nb, smp_b, first_b);
        set(WINDOW_SIZE - 1, state_prev);
        set(WINDOW_SIZE, state_new);
        FeaturePerId f;                                   // one track that starts at the dropped frame and one that spans it
        f.feature_id = 1; f.start_frame = WINDOW_SIZE; f.feature_per_frame.resize(1);
        est.f_manager.feature.push_back(f);
        f.feature_id = 2; f.start_frame = WINDOW_SIZE - 3; f.feature_per_frame.resize(4);
        est.f_manager.feature.push_back(f);
        est.marginalization_flag = Estimator::MARGIN_SECOND_NEW;
        est.slideWindow();
        const IntegrationBase* p = est.pre_integrations[WINDOW_SIZE - 1];
        memset(merged, 0, sizeof(*merged));
        merged->valid = 1; merged->sum_dt = p->sum_dt;
        for (int k = 0; k < 3; ++k) { merged->delta_p[k] = p->delta_p(k); merged->delta_v[k] = p->delta_v(k); merged->linearized_ba[k] = p->linearized_ba(k); merged->linearized_bg[k] = p->linearized_bg(k); }
        merged->delta_q[0] = p->delta_q.x(); merged->delta_q[1] = p->delta_q.y(); merged->delta_q[2] = p->delta_q.z(); merged->delta_q[3] = p->delta_q.w();
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c) { merged->jacobian[r * 15 + c] = p->jacobian(r, c); merged->covariance[r * 15 + c] = p->covariance(r, c); }
        *nsamples_out = (int)p->dt_buf.size();
        const int i = WINDOW_SIZE - 1;
        Quaterniond q(est.Rs[i]);
        const double so[16] = {est.Ps[i].x(), est.Ps[i].y(), est.Ps[i].z(), q.x(), q.y(), q.z(), q.w(), est.Vs[i].x(), est.Vs[i].y(), est.Vs[i].z(),
                               est.Bas[i].x(), est.Bas[i].y(), est.Bas[i].z(), est.Bgs[i].x(), est.Bgs[i].y(), est.Bgs[i].z()};
        memcpy(state_out, so, sizeof(so));
        *slot_ws_released = est.pre_integrations[WINDOW_SIZE] == nullptr;
        // removeFront (feature_manager.cpp:333-351): track 1 moved to start WINDOW_SIZE-1, track 2 lost its observation at WINDOW_SIZE-1
        auto it = est.f_manager.feature.begin();
        if (it->start_frame != WINDOW_SIZE - 1) return -3;
        ++it;
        if (it->feature_per_frame.size() != 3) return -4;
        delete est.pre_integrations[WINDOW_SIZE - 1];
        est.pre_integrations[WINDOW_SIZE - 1] = nullptr;
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "vins_host_slide_second_new: %s\n", e.what());
        return -2;
    }
}


// Configuration readers (host/yaml_config.h): out[0..31] = the estimator globals in a fixed order, out[32..] the tracker's;
// intr[8] = fx fy cx cy k1 k2 p1 p2 after FeatureTracker::readIntrinsicParameter.
extern "C" int vins_host_read_parameters(const char* config_file, double* out /*64*/, double* intr /*8*/) {
    try {
        readEstimatorParameters(config_file);
        readFeatureTrackerParameters(config_file, "/vins/");
        FeatureTracker tr;
        tr.readIntrinsicParameter(config_file);
        int k = 0;
        out[k++] = SOLVER_TIME; out[k++] = NUM_ITERATIONS; out[k++] = MIN_PARALLAX; out[k++] = ACC_N; out[k++] = ACC_W; out[k++] = GYR_N; out[k++] = GYR_W;
        out[k++] = G_NORM; out[k++] = ROW_D; out[k++] = COL_D; out[k++] = ESTIMATE_EXTRINSIC; out[k++] = INIT_DEPTH; out[k++] = BIAS_ACC_THRESHOLD;
        out[k++] = BIAS_GYR_THRESHOLD; out[k++] = TD; out[k++] = ESTIMATE_TD; out[k++] = ROLLING_SHUTTER; out[k++] = TR;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out[k++] = RIC[0](r, c);
        for (int r = 0; r < 3; ++r) out[k++] = TIC[0](r);
        k = 32;
        out[k++] = MAX_CNT; out[k++] = MIN_DIST; out[k++] = ROW; out[k++] = COL; out[k++] = FREQ; out[k++] = F_THRESHOLD; out[k++] = SHOW_TRACK;
        out[k++] = EQUALIZE; out[k++] = FISHEYE; out[k++] = FE_WINDOW_SIZE; out[k++] = FOCAL_LENGTH;
        const double v[8] = {tr.m_camera.fx, tr.m_camera.fy, tr.m_camera.cx, tr.m_camera.cy, tr.m_camera.k1, tr.m_camera.k2, tr.m_camera.p1, tr.m_camera.p2};
        memcpy(intr, v, sizeof(v));
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "vins_host_read_parameters: %s\n", e.what());
        return -1;
    }
}
extern "C" const char* vins_host_result_path() { return VINS_RESULT_PATH.c_str(); }
extern "C" const char* vins_host_imu_topic() { return IMU_TOPIC.c_str(); }
