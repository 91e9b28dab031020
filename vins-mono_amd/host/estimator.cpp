// estimator.cpp — Estimator::optimization() on the MI355X path.  vector2double() / double2vector() follow
// vins_estimator/src/estimator.cpp:486-619; optimization() follows :670-1003 with the ceres::Problem, ceres::Solve
// and the MarginalizationInfo machinery replaced by ONE call of vg_ba_optimize (solve + gauge fix + marginalization).
#include "estimator.h"
#include <cstring>
#include <stdexcept>
#include <string>

int ESTIMATE_EXTRINSIC = 0, ESTIMATE_TD = 0, NUM_ITERATIONS = 8;
double TD = 0, TR = 0, ROW_D = 480, FOCAL_LENGTH_D = 460.0, G_NORM = 9.81007;

int FeatureManager::getFeatureCount() {                         // feature_manager.cpp:28-42
    int cnt = 0;
    for (auto& it : feature) {
        it.used_num = it.feature_per_frame.size();
        if (it.used_num >= 2 && it.start_frame < WINDOW_SIZE - 2) cnt++;
    }
    return cnt;
}

Estimator::Estimator() { for (auto& p : pre_integrations) p = nullptr; memset(&last_summary, 0, sizeof(last_summary)); }
Estimator::~Estimator() { if (vg_) vg_destroy(vg_); delete last_marginalization_info; }

void Estimator::vector2double() {                               // estimator.cpp:486-528
    for (int i = 0; i <= WINDOW_SIZE; i++) {
        para_Pose[i][0] = Ps[i].x(); para_Pose[i][1] = Ps[i].y(); para_Pose[i][2] = Ps[i].z();
        Quaterniond q{Rs[i]};
        para_Pose[i][3] = q.x(); para_Pose[i][4] = q.y(); para_Pose[i][5] = q.z(); para_Pose[i][6] = q.w();
        para_SpeedBias[i][0] = Vs[i].x(); para_SpeedBias[i][1] = Vs[i].y(); para_SpeedBias[i][2] = Vs[i].z();
        para_SpeedBias[i][3] = Bas[i].x(); para_SpeedBias[i][4] = Bas[i].y(); para_SpeedBias[i][5] = Bas[i].z();
        para_SpeedBias[i][6] = Bgs[i].x(); para_SpeedBias[i][7] = Bgs[i].y(); para_SpeedBias[i][8] = Bgs[i].z();
    }
    for (int i = 0; i < NUM_OF_CAM; i++) {
        para_Ex_Pose[i][0] = tic[i].x(); para_Ex_Pose[i][1] = tic[i].y(); para_Ex_Pose[i][2] = tic[i].z();
        Quaterniond q{ric[i]};
        para_Ex_Pose[i][3] = q.x(); para_Ex_Pose[i][4] = q.y(); para_Ex_Pose[i][5] = q.z(); para_Ex_Pose[i][6] = q.w();
    }
    int feature_index = -1;                                     // FeatureManager::getDepthVector (:184-200)
    for (auto& it : f_manager.feature) {
        it.used_num = it.feature_per_frame.size();
        if (!(it.used_num >= 2 && it.start_frame < WINDOW_SIZE - 2)) continue;
        para_Feature[++feature_index][0] = 1. / it.estimated_depth;
    }
    if (ESTIMATE_TD) para_Td[0][0] = td;
}

void Estimator::double2vector() {
    // estimator.cpp:530-619.  The yaw / position gauge fix (:541-577) is applied on the device (vg_ba_state is
    // post-fix), so this only converts the para_* arrays back to the Eigen members and the depths (setDepth, :141-159).
    for (int i = 0; i <= WINDOW_SIZE; i++) {
        Rs[i] = Quaterniond(para_Pose[i][6], para_Pose[i][3], para_Pose[i][4], para_Pose[i][5]).toRotationMatrix();
        Ps[i] = Vector3d(para_Pose[i][0], para_Pose[i][1], para_Pose[i][2]);
        Vs[i] = Vector3d(para_SpeedBias[i][0], para_SpeedBias[i][1], para_SpeedBias[i][2]);
        Bas[i] = Vector3d(para_SpeedBias[i][3], para_SpeedBias[i][4], para_SpeedBias[i][5]);
        Bgs[i] = Vector3d(para_SpeedBias[i][6], para_SpeedBias[i][7], para_SpeedBias[i][8]);
    }
    for (int i = 0; i < NUM_OF_CAM; i++) {
        tic[i] = Vector3d(para_Ex_Pose[i][0], para_Ex_Pose[i][1], para_Ex_Pose[i][2]);
        ric[i] = Quaterniond(para_Ex_Pose[i][6], para_Ex_Pose[i][3], para_Ex_Pose[i][4], para_Ex_Pose[i][5]).toRotationMatrix();
    }
    int feature_index = -1;
    for (auto& it : f_manager.feature) {
        it.used_num = it.feature_per_frame.size();
        if (!(it.used_num >= 2 && it.start_frame < WINDOW_SIZE - 2)) continue;
        it.estimated_depth = 1.0 / para_Feature[++feature_index][0];
        it.solve_flag = it.estimated_depth < 0 ? 2 : 1;
    }
    if (ESTIMATE_TD) td = para_Td[0][0];
}

void Estimator::optimization() {
    if (!vg_ && vg_create(&vg_) != VG_OK) throw std::runtime_error("vg_create failed: no MI355X / libvinsgpu (no CPU fallback)");
    vector2double();                                            // estimator.cpp:701
    const int K = WINDOW_SIZE + 1;
    // ---- factor tables instead of problem.AddResidualBlock (:711-764)
    vector<int> lm_start, lm_nobs, lm_off;
    vector<double> obs;
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
    vector<vg_imu_preint> imu(K - 1);
    for (int i = 0; i < WINDOW_SIZE; i++) {                      // IMUFactor(pre_integrations[j]), j = i + 1 (:711-718)
        const IntegrationBase* p = pre_integrations[i + 1];
        vg_imu_preint& m = imu[i];
        memset(&m, 0, sizeof(m));
        if (!p) continue;
        m.valid = 1; m.sum_dt = p->sum_dt;
        for (int k = 0; k < 3; ++k) { m.delta_p[k] = p->delta_p(k); m.delta_v[k] = p->delta_v(k); m.linearized_ba[k] = p->linearized_ba(k); m.linearized_bg[k] = p->linearized_bg(k); }
        m.delta_q[0] = p->delta_q.x(); m.delta_q[1] = p->delta_q.y(); m.delta_q[2] = p->delta_q.z(); m.delta_q[3] = p->delta_q.w();
        memcpy(m.jacobian, p->jacobian, sizeof(m.jacobian)); memcpy(m.covariance, p->covariance, sizeof(m.covariance));
    }
    vg_ba_problem pb;
    memset(&pb, 0, sizeof(pb));
    pb.K = K; pb.L = L; pb.n_obs = (int)obs.size() / 7;
    pb.pose = &para_Pose[0][0]; pb.speedbias = &para_SpeedBias[0][0]; pb.ex_pose = &para_Ex_Pose[0][0]; pb.td = para_Td[0][0];
    pb.inv_depth = &para_Feature[0][0];
    pb.lm_start = lm_start.data(); pb.lm_nobs = lm_nobs.data(); pb.lm_obs_off = lm_off.data(); pb.obs = obs.data(); pb.imu = imu.data();
    if (last_marginalization_info) {                             // MarginalizationFactor (:703-709)
        MarginalizationInfo* mi = last_marginalization_info;
        pb.prior_n = mi->n; pb.prior_nblocks = (int)mi->keep_block_kind.size();
        pb.prior_block_kind = mi->keep_block_kind.data(); pb.prior_block_index = mi->keep_block_index.data();
        pb.prior_J0 = mi->linearized_jacobians.data(); pb.prior_r0 = mi->linearized_residuals.data(); pb.prior_x0 = mi->keep_block_data.data();
    }
    pb.estimate_extrinsic = ESTIMATE_EXTRINSIC ? 1 : 0; pb.estimate_td = ESTIMATE_TD ? 1 : 0; pb.max_iters = NUM_ITERATIONS;
    pb.focal = FOCAL_LENGTH_D; pb.tr = TR; pb.row = ROW_D; pb.g_norm = G_NORM;
    // ---- outputs
    vector<double> lam(L > 0 ? L : 1);
    vg_ba_state st;
    st.pose = &para_Pose[0][0]; st.speedbias = &para_SpeedBias[0][0]; st.ex_pose = &para_Ex_Pose[0][0]; st.td = &para_Td[0][0];
    st.inv_depth = lam.data(); st.relo_pose = nullptr;
    const int cap = 6 * K + 32, capb = K + 8;
    MarginalizationInfo* mi_new = new MarginalizationInfo();
    mi_new->keep_block_kind.resize(capb); mi_new->keep_block_index.resize(capb); mi_new->keep_block_data.resize(9 * capb);
    mi_new->linearized_jacobians.resize((size_t)cap * cap); mi_new->linearized_residuals.resize(cap);
    vg_ba_prior pr;
    pr.cap = cap; pr.cap_blocks = capb; pr.block_kind = mi_new->keep_block_kind.data(); pr.block_index = mi_new->keep_block_index.data();
    pr.J0 = mi_new->linearized_jacobians.data(); pr.r0 = mi_new->linearized_residuals.data(); pr.x0 = mi_new->keep_block_data.data();
    const int flag = marginalization_flag == MARGIN_OLD ? VG_MARGIN_OLD : VG_MARGIN_SECOND_NEW;
    const int rc = vg_ba_optimize(vg_, &pb, flag, &st, &last_summary, &pr);       // ceres::Solve (:818) + marginalization (:825-1000)
    if (rc != VG_OK && rc != VG_ERR_NUMERIC) { delete mi_new; throw std::runtime_error(std::string("vg_ba_optimize: ") + vg_last_error(vg_)); }
    for (int l = 0; l < L; ++l) para_Feature[l][0] = lam[l];
    double2vector();                                            // :823
    if (pr.valid) {                                             // last_marginalization_info = marginalization_info (:926-929 / :992-996)
        mi_new->n = pr.n; mi_new->m = pr.m;
        mi_new->keep_block_kind.resize(pr.nblocks); mi_new->keep_block_index.resize(pr.nblocks);
        mi_new->linearized_jacobians.resize((size_t)pr.n * pr.n); mi_new->linearized_residuals.resize(pr.n);
        delete last_marginalization_info;
        last_marginalization_info = mi_new;
    } else {
        delete mi_new;      // MARGIN_SECOND_NEW without Pose[WINDOW_SIZE-1] in the prior: the old prior stays (:935-936)
    }
}
