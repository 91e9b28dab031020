// estimator.cpp — Estimator::optimization() on the MI355X path.  vector2double() / double2vector() follow
// vins_estimator/src/estimator.cpp:486-619; optimization() follows :670-1003 with the ceres::Problem, ceres::Solve
// and the MarginalizationInfo machinery replaced by ONE call of vg_ba_optimize (solve + gauge fix + marginalization).
#include "estimator.h"
#include "yaml_config.h"
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>

int ESTIMATE_EXTRINSIC = 0, ESTIMATE_TD = 0, NUM_ITERATIONS = 8;
double TD = 0, TR = 0, ROW_D = 480, FOCAL_LENGTH_D = 460.0, G_NORM = 9.81007, INIT_DEPTH = 5.0, SOLVER_TIME = 0.0;
double ACC_N = 0.08, ACC_W = 0.00004, GYR_N = 0.004, GYR_W = 2.0e-6;      // config/euroc/euroc_config.yaml:58-62
double MIN_PARALLAX = 10.0 / 460.0, BIAS_ACC_THRESHOLD = 0.1, BIAS_GYR_THRESHOLD = 0.1, COL_D = 752;
int ROLLING_SHUTTER = 0;
std::string IMU_TOPIC, VINS_RESULT_PATH, EX_CALIB_RESULT_PATH;
std::vector<Matrix3d> RIC;
std::vector<Vector3d> TIC;

void readEstimatorParameters(const std::string& config_file) {               // vins_estimator/src/parameters.cpp:42-137
    VinsYaml fs;
    if (!fs.load(config_file)) throw std::runtime_error("ERROR: Wrong path to settings: " + config_file);
    IMU_TOPIC = fs.str("imu_topic");
    SOLVER_TIME = fs.number("max_solver_time");
    NUM_ITERATIONS = (int)fs.number("max_num_iterations");
    MIN_PARALLAX = fs.number("keyframe_parallax");
    MIN_PARALLAX = MIN_PARALLAX / FOCAL_LENGTH_D;
    const std::string OUTPUT_PATH = fs.str("output_path");
    VINS_RESULT_PATH = OUTPUT_PATH + "/vins_result_no_loop.csv";
    ACC_N = fs.number("acc_n"); ACC_W = fs.number("acc_w");
    GYR_N = fs.number("gyr_n"); GYR_W = fs.number("gyr_w");
    G_NORM = fs.number("g_norm");
    ROW_D = fs.number("image_height");
    COL_D = fs.number("image_width");
    ESTIMATE_EXTRINSIC = (int)fs.number("estimate_extrinsic");
    RIC.clear(); TIC.clear();
    if (ESTIMATE_EXTRINSIC == 2) {
        RIC.push_back(Matrix3d());           // identity
        TIC.push_back(Vector3d());
        EX_CALIB_RESULT_PATH = OUTPUT_PATH + "/extrinsic_parameter.csv";
    } else {
        if (ESTIMATE_EXTRINSIC == 1) EX_CALIB_RESULT_PATH = OUTPUT_PATH + "/extrinsic_parameter.csv";
        const VinsYaml::Matrix* R = fs.matrix("extrinsicRotation");
        const VinsYaml::Matrix* T = fs.matrix("extrinsicTranslation");
        if (!R || !T || R->rows != 3 || R->cols != 3 || T->data.size() != 3) throw std::runtime_error("extrinsicRotation / extrinsicTranslation missing or not 3x3 / 3x1");
        Matrix3d eigen_R;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) eigen_R(r, c) = R->data[3 * r + c];
        // Eigen::Quaterniond Q(eigen_R); eigen_R = Q.normalized();   (:105-106: the matrix is re-orthonormalised through a quaternion)
        const Quaterniond Q(eigen_R);
        const double qn = std::sqrt(Q.w() * Q.w() + Q.x() * Q.x() + Q.y() * Q.y() + Q.z() * Q.z());
        RIC.push_back(Quaterniond(Q.w() / qn, Q.x() / qn, Q.y() / qn, Q.z() / qn).toRotationMatrix());
        TIC.push_back(Vector3d(T->data[0], T->data[1], T->data[2]));
    }
    INIT_DEPTH = 5.0;
    BIAS_ACC_THRESHOLD = 0.1;
    BIAS_GYR_THRESHOLD = 0.1;
    TD = fs.number("td");
    ESTIMATE_TD = (int)fs.number("estimate_td");
    ROLLING_SHUTTER = (int)fs.number("rolling_shutter");
    TR = ROLLING_SHUTTER ? fs.number("rolling_shutter_tr") : 0;
}

int FeatureManager::getFeatureCount() {                         // feature_manager.cpp:28-42
    int cnt = 0;
    for (auto& it : feature) {
        it.used_num = it.feature_per_frame.size();
        if (it.used_num >= 2 && it.start_frame < WINDOW_SIZE - 2) cnt++;
    }
    return cnt;
}

void FeatureManager::removeBackShiftDepth(Matrix3d marg_R, Vector3d marg_P, Matrix3d new_R, Vector3d new_P) {   // feature_manager.cpp:275-313
    for (auto it = feature.begin(), it_next = feature.begin(); it != feature.end(); it = it_next) {
        it_next++;
        if (it->start_frame != 0) it->start_frame--;
        else {
            Vector3d uv_i = it->feature_per_frame[0].point;
            it->feature_per_frame.erase(it->feature_per_frame.begin());
            if (it->feature_per_frame.size() < 2) { feature.erase(it); continue; }
            Vector3d pts_i = uv_i * it->estimated_depth;
            Vector3d w_pts_i = marg_R * pts_i + marg_P;
            Vector3d pts_j = new_R.transpose() * (w_pts_i - new_P);
            const double dep_j = pts_j(2);
            it->estimated_depth = dep_j > 0 ? dep_j : INIT_DEPTH;
        }
    }
}

void FeatureManager::removeBack() {                              // feature_manager.cpp:315-331
    for (auto it = feature.begin(), it_next = feature.begin(); it != feature.end(); it = it_next) {
        it_next++;
        if (it->start_frame != 0) it->start_frame--;
        else {
            it->feature_per_frame.erase(it->feature_per_frame.begin());
            if (it->feature_per_frame.size() == 0) feature.erase(it);
        }
    }
}

void FeatureManager::removeFront(int frame_count) {              // feature_manager.cpp:333-351
    for (auto it = feature.begin(), it_next = feature.begin(); it != feature.end(); it = it_next) {
        it_next++;
        if (it->start_frame == frame_count) it->start_frame--;
        else {
            const int j = WINDOW_SIZE - 1 - it->start_frame;
            if ((int)it->feature_per_frame.size() - 1 < j) continue;       // endFrame() < frame_count - 1
            it->feature_per_frame.erase(it->feature_per_frame.begin() + j);
            if (it->feature_per_frame.size() == 0) feature.erase(it);
        }
    }
}

void Estimator::repropagate(IntegrationBase* p) {
    if (!vg_ && vg_create(&vg_) != VG_OK) throw std::runtime_error("vg_create failed: no MI355X / libvinsgpu (no CPU fallback)");
    const int n = (int)p->dt_buf.size();
    std::vector<double> smp((size_t)(n > 0 ? n : 1) * 7, 0.0);
    for (int i = 0; i < n; ++i) {
        double* r = &smp[(size_t)7 * i];
        r[0] = p->dt_buf[i];
        for (int k = 0; k < 3; ++k) { r[1 + k] = p->acc_buf[i](k); r[4 + k] = p->gyr_buf[i](k); }
    }
    const double first[6] = {p->linearized_acc(0), p->linearized_acc(1), p->linearized_acc(2), p->linearized_gyr(0), p->linearized_gyr(1), p->linearized_gyr(2)};
    const double bias[6] = {p->linearized_ba(0), p->linearized_ba(1), p->linearized_ba(2), p->linearized_bg(0), p->linearized_bg(1), p->linearized_bg(2)};
    const double noise[4] = {ACC_N, GYR_N, ACC_W, GYR_W};
    const int off[2] = {0, n};
    vg_imu_preint out;
    if (vg_imu_preintegrate(vg_, 1, off, smp.data(), first, bias, noise, &out) != VG_OK) throw std::runtime_error(std::string("vg_imu_preintegrate: ") + vg_last_error(vg_));
    p->sum_dt = out.sum_dt;
    p->delta_p = Vector3d(out.delta_p[0], out.delta_p[1], out.delta_p[2]);
    p->delta_v = Vector3d(out.delta_v[0], out.delta_v[1], out.delta_v[2]);
    p->delta_q = Quaterniond(out.delta_q[3], out.delta_q[0], out.delta_q[1], out.delta_q[2]);
    for (int r = 0; r < 15; ++r)
        for (int c = 0; c < 15; ++c) { p->jacobian(r, c) = out.jacobian[r * 15 + c]; p->covariance(r, c) = out.covariance[r * 15 + c]; }
}

void Estimator::slideWindow() {
    // estimator.cpp:1005-1126 for frame_count == WINDOW_SIZE and solver_flag == NON_LINEAR; all_image_frame belongs to the
    // initialiser and is not mirrored here.  Ownership as in the reference: the IntegrationBase that ends up in slot
    // WINDOW_SIZE is deleted here (:1036 / :1093); the caller installs the next interval's object (the reference re-creates
    // an empty one and processIMU fills it).
    if (marginalization_flag == MARGIN_OLD) {
        back_R0 = Rs[0];
        back_P0 = Ps[0];
        for (int i = 0; i < WINDOW_SIZE; i++) {
            Rs[i].swap(Rs[i + 1]);
            std::swap(pre_integrations[i], pre_integrations[i + 1]);
            Ps[i].swap(Ps[i + 1]);
            Vs[i].swap(Vs[i + 1]);
            Bas[i].swap(Bas[i + 1]);
            Bgs[i].swap(Bgs[i + 1]);
        }
        Ps[WINDOW_SIZE] = Ps[WINDOW_SIZE - 1]; Vs[WINDOW_SIZE] = Vs[WINDOW_SIZE - 1]; Rs[WINDOW_SIZE] = Rs[WINDOW_SIZE - 1];
        Bas[WINDOW_SIZE] = Bas[WINDOW_SIZE - 1]; Bgs[WINDOW_SIZE] = Bgs[WINDOW_SIZE - 1];
        delete pre_integrations[WINDOW_SIZE];                    // (the object that was in slot 0)
        pre_integrations[WINDOW_SIZE] = nullptr;                 // the caller installs the next interval's pre-integration
        // slideWindowOld(), shift_depth = true (:1115-1126)
        Matrix3d R0 = back_R0 * ric[0], R1 = Rs[0] * ric[0];
        Vector3d P0 = back_P0 + back_R0 * tic[0], P1 = Ps[0] + Rs[0] * tic[0];
        f_manager.removeBackShiftDepth(R0, P0, R1, P1);
    } else {
        // :1069-1085: the IMU samples between frames WINDOW_SIZE-1 and WINDOW_SIZE are appended to the interval that ends at
        // frame WINDOW_SIZE-1, which then spans WINDOW_SIZE-2 .. WINDOW_SIZE.  The reference continues the running
        // integration sample by sample (push_back); integrating the concatenated list with the same linearisation biases is
        // the same sequence of operations (vg_imu_preintegrate).
        IntegrationBase* a = pre_integrations[WINDOW_SIZE - 1];
        IntegrationBase* b = pre_integrations[WINDOW_SIZE];
        if (b) {
            if (!a) throw std::runtime_error("slideWindow(MARGIN_SECOND_NEW): pre_integrations[WINDOW_SIZE - 1] is missing");
            if (b->dt_buf.empty() || (a->dt_buf.empty() && a->sum_dt != 0))
                throw std::runtime_error("slideWindow(MARGIN_SECOND_NEW): the pre-integrations carry no raw IMU samples (dt_buf / acc_buf / gyr_buf) to merge");
            a->dt_buf.insert(a->dt_buf.end(), b->dt_buf.begin(), b->dt_buf.end());
            a->acc_buf.insert(a->acc_buf.end(), b->acc_buf.begin(), b->acc_buf.end());
            a->gyr_buf.insert(a->gyr_buf.end(), b->gyr_buf.begin(), b->gyr_buf.end());
            repropagate(a);
        }
        Ps[WINDOW_SIZE - 1] = Ps[WINDOW_SIZE]; Vs[WINDOW_SIZE - 1] = Vs[WINDOW_SIZE]; Rs[WINDOW_SIZE - 1] = Rs[WINDOW_SIZE];
        Bas[WINDOW_SIZE - 1] = Bas[WINDOW_SIZE]; Bgs[WINDOW_SIZE - 1] = Bgs[WINDOW_SIZE];
        delete pre_integrations[WINDOW_SIZE];                    // :1093
        pre_integrations[WINDOW_SIZE] = nullptr;
        f_manager.removeFront(WINDOW_SIZE);                      // slideWindowNew() (:1107-1111)
    }
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

// Utility::R2ypr / ypr2R / normalizeAngle (vins_estimator/src/utility/utility.h:70-143), degrees
static Vector3d R2ypr_deg(const Matrix3d& R) {
    const double n0 = R(0, 0), n1 = R(1, 0), n2 = R(2, 0), o0 = R(0, 1), o1 = R(1, 1), a0 = R(0, 2), a1 = R(1, 2);
    const double y = atan2(n1, n0);
    const double p = atan2(-n2, n0 * cos(y) + n1 * sin(y));
    const double r = atan2(a0 * sin(y) - a1 * cos(y), -o0 * sin(y) + o1 * cos(y));
    return Vector3d(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);
}
static Matrix3d yaw2R_deg(double yaw_deg) {
    const double y = yaw_deg / 180.0 * M_PI;
    Matrix3d R;
    R(0, 0) = cos(y); R(0, 1) = -sin(y); R(0, 2) = 0;
    R(1, 0) = sin(y); R(1, 1) = cos(y);  R(1, 2) = 0;
    R(2, 0) = 0;      R(2, 1) = 0;       R(2, 2) = 1;
    return R;
}
static double normalize_angle_deg(double a) {
    return a > 0 ? a - 360.0 * std::floor((a + 180.0) / 360.0) : a + 360.0 * std::floor((-a + 180.0) / 360.0);
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
    if (relocalization_info) {
        // relative pose between the two loop frames (:596-616).  With matched landmarks relo_Pose was a parameter block of the
        // problem and comes back from the device already in the fixed gauge (rot_diff * (p - P0) + origin_P0, rot_diff * R): it
        // IS the reference's relo_r / relo_t.  Without a match the reference still adds the block (:771-772; no residual touches
        // it, the solve leaves it alone) and applies the gauge transform to it (:598-603): done here with the transform the
        // device reports (vg_ba_summary::gauge_rot / gauge_p0); origin_P0 = the gauge-fixed Ps[0].
        Matrix3d relo_r = Quaterniond(relo_Pose[6], relo_Pose[3], relo_Pose[4], relo_Pose[5]).toRotationMatrix();
        Vector3d relo_t(relo_Pose[0], relo_Pose[1], relo_Pose[2]);
        if (!relo_in_problem) {
            Matrix3d rot;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) rot(r, c) = last_summary.gauge_rot[3 * r + c];
            const double qn = std::sqrt(relo_Pose[3] * relo_Pose[3] + relo_Pose[4] * relo_Pose[4] + relo_Pose[5] * relo_Pose[5] + relo_Pose[6] * relo_Pose[6]);
            relo_r = rot * Quaterniond(relo_Pose[6] / qn, relo_Pose[3] / qn, relo_Pose[4] / qn, relo_Pose[5] / qn).toRotationMatrix();
            relo_t = rot * Vector3d(relo_Pose[0] - last_summary.gauge_p0[0], relo_Pose[1] - last_summary.gauge_p0[1], relo_Pose[2] - last_summary.gauge_p0[2]) + Ps[0];
        }
        relo_r_fixed = relo_r; relo_t_fixed = relo_t;
        const double drift_correct_yaw = R2ypr_deg(prev_relo_r).x() - R2ypr_deg(relo_r).x();
        drift_correct_r = yaw2R_deg(drift_correct_yaw);
        drift_correct_t = prev_relo_t - drift_correct_r * relo_t;
        relo_relative_t = relo_r.transpose() * (Ps[relo_frame_local_index] - relo_t);
        relo_relative_q = Quaterniond(relo_r.transpose() * Rs[relo_frame_local_index]);
        relo_relative_yaw = normalize_angle_deg(R2ypr_deg(Rs[relo_frame_local_index]).x() - R2ypr_deg(relo_r).x());
        relocalization_info = false;
    }
}

void Estimator::optimization() {
    if (vg_abi_version() != VG_ABI_VERSION) throw std::runtime_error("libvinsgpu.so was built from another include/vinsgpu.h (ABI version mismatch)");
    if (!vg_ && vg_create(&vg_) != VG_OK) throw std::runtime_error("vg_create failed: no MI355X / libvinsgpu (no CPU fallback)");
    collectPrior();                                             // the previous frame's marginalization result, if still on the device
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
    // relocalisation factors (:769-801): ProjectionFactor(first observation, matched point) on (para_Pose[start], relo_Pose,
    // ex, depth) for every optimised landmark that started at or before the loop frame and has a match (match_points are
    // sorted by feature id, like f_manager.feature)
    vector<int> relo_lm;
    vector<double> relo_xy;
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
    vector<vg_imu_preint> imu(K - 1);
    for (int i = 0; i < WINDOW_SIZE; i++) {                      // IMUFactor(pre_integrations[j]), j = i + 1 (:711-718)
        const IntegrationBase* p = pre_integrations[i + 1];
        vg_imu_preint& m = imu[i];
        memset(&m, 0, sizeof(m));
        if (!p) continue;
        m.valid = 1; m.sum_dt = p->sum_dt;
        for (int k = 0; k < 3; ++k) { m.delta_p[k] = p->delta_p(k); m.delta_v[k] = p->delta_v(k); m.linearized_ba[k] = p->linearized_ba(k); m.linearized_bg[k] = p->linearized_bg(k); }
        m.delta_q[0] = p->delta_q.x(); m.delta_q[1] = p->delta_q.y(); m.delta_q[2] = p->delta_q.z(); m.delta_q[3] = p->delta_q.w();
        for (int r = 0; r < 15; ++r)                            // Eigen is column-major, the C-ABI row-major: convert explicitly
            for (int c = 0; c < 15; ++c) { m.jacobian[r * 15 + c] = p->jacobian(r, c); m.covariance[r * 15 + c] = p->covariance(r, c); }
    }
    vg_ba_problem pb;
    memset(&pb, 0, sizeof(pb));
    pb.K = K; pb.L = L; pb.n_obs = (int)obs.size() / 7;
    pb.pose = &para_Pose[0][0]; pb.speedbias = &para_SpeedBias[0][0]; pb.ex_pose = &para_Ex_Pose[0][0]; pb.td = para_Td[0][0];
    pb.inv_depth = &para_Feature[0][0];
    pb.lm_start = lm_start.data(); pb.lm_nobs = lm_nobs.data(); pb.lm_obs_off = lm_off.data(); pb.obs = obs.data(); pb.imu = imu.data();
    // MarginalizationFactor(last_marginalization_info) on last_marginalization_parameter_blocks (:703-709): the blocks are
    // identified by their para_* ADDRESSES, as in the reference; the C-ABI wants (kind, frame index)
    vector<int> pkind, pindex;
    vector<double> pJ0, px0;
    if (last_marginalization_info) {
        MarginalizationInfo* mi = last_marginalization_info;
        const int n = mi->n, nb = (int)mi->keep_block_size.size();
        for (int b = 0; b < nb; ++b) {
            int kind, index;
            if (!block_of(last_marginalization_parameter_blocks[b], kind, index)) throw std::runtime_error("prior block address outside the para_* arrays");
            pkind.push_back(kind); pindex.push_back(index);
            px0.insert(px0.end(), mi->keep_block_data[b], mi->keep_block_data[b] + mi->keep_block_size[b]);
        }
        pJ0.resize((size_t)n * n);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) pJ0[(size_t)r * n + c] = mi->linearized_jacobians(r, c);
        pb.prior_n = n; pb.prior_nblocks = nb;
        pb.prior_block_kind = pkind.data(); pb.prior_block_index = pindex.data();
        pb.prior_J0 = pJ0.data(); pb.prior_r0 = mi->linearized_residuals.data(); pb.prior_x0 = px0.data();
    }
    relo_in_problem = relocalization_info && !relo_lm.empty();
    if (relo_in_problem) {
        pb.relo_n = (int)relo_lm.size(); pb.relo_pose = relo_Pose; pb.relo_lm = relo_lm.data(); pb.relo_xy = relo_xy.data();
    }
    pb.estimate_extrinsic = ESTIMATE_EXTRINSIC ? 1 : 0; pb.estimate_td = ESTIMATE_TD ? 1 : 0; pb.max_iters = NUM_ITERATIONS;
    pb.focal = FOCAL_LENGTH_D; pb.tr = TR; pb.row = ROW_D; pb.g_norm = G_NORM;
    // options.max_solver_time_in_seconds (:812-815); SOLVER_TIME = 0 (the default of this library) = no cap: a wall-clock cap makes
    // the result depend on timing
    pb.max_solver_time_s = marginalization_flag == MARGIN_OLD ? SOLVER_TIME * 4.0 / 5.0 : SOLVER_TIME;
    // ---- outputs
    vector<double> lam(L > 0 ? L : 1);
    vg_ba_state st;
    st.pose = &para_Pose[0][0]; st.speedbias = &para_SpeedBias[0][0]; st.ex_pose = &para_Ex_Pose[0][0]; st.td = &para_Td[0][0];
    st.inv_depth = lam.data(); st.relo_pose = pb.relo_n ? relo_Pose : nullptr;
    const int flag = marginalization_flag == MARGIN_OLD ? VG_MARGIN_OLD : VG_MARGIN_SECOND_NEW;
    // ceres::Solve (:818) returns here as soon as the states are back on the host; the marginalization (:825-1000) keeps
    // running on the device and is collected by collectPrior() -- its result is first needed by the next optimization()
    const int rc = vg_ba_optimize_begin(vg_, &pb, flag, &st, &last_summary);
    if (rc != VG_OK && rc != VG_ERR_NUMERIC) throw std::runtime_error(std::string("vg_ba_optimize_begin: ") + vg_last_error(vg_));
    for (int l = 0; l < L; ++l) para_Feature[l][0] = lam[l];
    double2vector();                                            // :823
    solver_failed = rc == VG_ERR_NUMERIC;
    prior_pending = true;
    if (solver_failed) collectPrior();                          // (drops the priors, see there)
}

// The marginalization result of the last optimization(): last_marginalization_info / ..._parameter_blocks are replaced when
// the device produced a prior (:926-929 / :992-996).  Called by the next optimization(); call it explicitly before reading
// last_marginalization_info (tests, serialisation).
void Estimator::collectPrior() {
    if (!prior_pending) return;
    prior_pending = false;
    const int K = WINDOW_SIZE + 1;
    const int cap = 6 * K + 32, capb = K + 8;
    vector<int> nkind(capb), nindex(capb);
    vector<double> nJ0((size_t)cap * cap), nr0(cap), nx0(9 * capb);      // x0 holds 9 doubles per block at most (speed-bias)
    vg_ba_prior pr;
    memset(&pr, 0, sizeof(pr));
    pr.cap = cap; pr.cap_blocks = capb; pr.block_kind = nkind.data(); pr.block_index = nindex.data();
    pr.J0 = nJ0.data(); pr.r0 = nr0.data(); pr.x0 = nx0.data();
    const int rc = vg_ba_optimize_prior(vg_, &pr);
    if (rc != VG_OK) throw std::runtime_error(std::string("vg_ba_optimize_prior: ") + vg_last_error(vg_));
    if (solver_failed) {
        // Non-finite cost / state on the device: no new prior exists, and the OLD one must not survive the caller's
        // slideWindow() (after MARGIN_OLD it still names pose / speed-bias 0 and un-shifted frame indices).  The reference
        // would carry a NaN prior on; dropping it is the conservative equivalent until failureDetection() restarts.
        delete last_marginalization_info;
        last_marginalization_info = nullptr;
        last_marginalization_parameter_blocks.clear();
        return;
    }
    if (pr.valid) {                                             // last_marginalization_info = marginalization_info (:926-929 / :992-996)
        MarginalizationInfo* mi = new MarginalizationInfo();
        mi->n = pr.n; mi->m = pr.m;
        mi->linearized_jacobians.resize(pr.n, pr.n);
        mi->linearized_residuals.resize(pr.n);
        for (int r = 0; r < pr.n; ++r) {
            mi->linearized_residuals(r) = nr0[r];
            for (int c = 0; c < pr.n; ++c) mi->linearized_jacobians(r, c) = nJ0[(size_t)r * pr.n + c];
        }
        vector<double*> blocks;
        int off = 0, x0o = 0;
        for (int b = 0; b < pr.nblocks; ++b) {
            const int kind = nkind[b], idx = nindex[b];
            const int gs = kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 7), ls = kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 6);
            mi->keep_block_size.push_back(gs);
            mi->keep_block_idx.push_back(pr.m + off);
            double* d = new double[gs];
            memcpy(d, nx0.data() + x0o, sizeof(double) * gs);
            mi->keep_block_data.push_back(d);
            // addr_shift (:913-924 / :969-990): the block's address in the SLID window (the device already re-labelled it)
            blocks.push_back(kind == VG_BLK_POSE ? para_Pose[idx] : kind == VG_BLK_SPEEDBIAS ? para_SpeedBias[idx] : kind == VG_BLK_EXPOSE ? para_Ex_Pose[0] : para_Td[0]);
            off += ls; x0o += gs;
        }
        delete last_marginalization_info;
        last_marginalization_info = mi;
        last_marginalization_parameter_blocks = blocks;
    }
    // else: MARGIN_SECOND_NEW without Pose[WINDOW_SIZE-1] in the prior: the old prior and its blocks stay (:935-936)
}

bool Estimator::block_of(const double* addr, int& kind, int& index) const {
    for (int i = 0; i <= WINDOW_SIZE; ++i) {
        if (addr == para_Pose[i]) { kind = VG_BLK_POSE; index = i; return true; }
        if (addr == para_SpeedBias[i]) { kind = VG_BLK_SPEEDBIAS; index = i; return true; }
    }
    if (addr == para_Ex_Pose[0]) { kind = VG_BLK_EXPOSE; index = 0; return true; }
    if (addr == para_Td[0]) { kind = VG_BLK_TD; index = 0; return true; }
    return false;
}
