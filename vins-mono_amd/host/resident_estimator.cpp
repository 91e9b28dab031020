// resident_estimator.cpp — see resident_estimator.h.  Host code around vg_ba_seq_*: the IMU side of Estimator::processIMU
// (estimator.cpp:83-117), the hand-over of an Estimator's window, and the per-frame packing of what processImage receives.
#include "resident_estimator.h"
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>

struct ResidentEstimators::Window {
    std::vector<double> pose, sb, ex, obs, depth, pJ0, pr0, px0;
    std::vector<vg_imu_preint> imu;
    std::vector<int> id, start, nobs, flag, pkind, pindex;
    vg_ba_problem prob;
    vg_ba_tracks tracks;
};

static void fill_preint(vg_imu_preint& m, const IntegrationBase* p) {
    memset(&m, 0, sizeof(m));
    if (!p) return;
    m.valid = 1; m.sum_dt = p->sum_dt;
    for (int k = 0; k < 3; ++k) { m.delta_p[k] = p->delta_p(k); m.delta_v[k] = p->delta_v(k); m.linearized_ba[k] = p->linearized_ba(k); m.linearized_bg[k] = p->linearized_bg(k); }
    m.delta_q[0] = p->delta_q.x(); m.delta_q[1] = p->delta_q.y(); m.delta_q[2] = p->delta_q.z(); m.delta_q[3] = p->delta_q.w();
    for (int r = 0; r < 15; ++r)
        for (int c = 0; c < 15; ++c) { m.jacobian[r * 15 + c] = p->jacobian(r, c); m.covariance[r * 15 + c] = p->covariance(r, c); }
}

ResidentEstimators::ResidentEstimators(int n, int max_features, int max_new_obs)
    : est_(n), win_(n, nullptr), max_features_(max_features), max_new_obs_(max_new_obs) {
    if (vg_abi_version() != VG_ABI_VERSION) throw std::runtime_error("libvinsgpu.so was built from another include/vinsgpu.h (ABI version mismatch)");
    if (vg_create(&vg_) != VG_OK) throw std::runtime_error("vg_create failed: no MI355X / libvinsgpu (no CPU fallback)");
    for (auto& e : est_) memset(&e.last_summary, 0, sizeof(e.last_summary));
}

ResidentEstimators::~ResidentEstimators() {
    for (Window* w : win_) delete w;
    if (vg_) { if (begun_) vg_ba_seq_end(vg_); vg_destroy(vg_); }
}

void ResidentEstimators::handOver(int i, Estimator& e, const Vector3d& acc_0, const Vector3d& gyr_0) {
    if (begun_) throw std::runtime_error("handOver after begin(): use reseed()");
    delete win_[i];
    win_[i] = pack(i, e, acc_0, gyr_0);
}

void ResidentEstimators::reseed(int i, Estimator& e, const Vector3d& acc_0, const Vector3d& gyr_0) {
    if (!begun_) throw std::runtime_error("reseed() before begin(): use handOver()");
    Window* w = pack(i, e, acc_0, gyr_0);
    const int rc = vg_ba_seq_import(vg_, i, &w->prob, &w->tracks);
    delete w;
    if (rc != VG_OK) throw std::runtime_error(std::string("vg_ba_seq_import: ") + vg_last_error(vg_));
}

ResidentEstimators::Window* ResidentEstimators::pack(int i, Estimator& e, const Vector3d& acc_0, const Vector3d& gyr_0) {
    const int K = WINDOW_SIZE + 1;
    One& o = est_[i];
    Window* w = new Window();
    e.collectPrior();                                             // a marginalization result still on the estimator's own handle
    e.vector2double();
    w->pose.assign(&e.para_Pose[0][0], &e.para_Pose[0][0] + 7 * K);
    w->sb.assign(&e.para_SpeedBias[0][0], &e.para_SpeedBias[0][0] + 9 * K);
    w->ex.assign(&e.para_Ex_Pose[0][0], &e.para_Ex_Pose[0][0] + 7);
    w->imu.resize(K - 1);
    for (int k = 0; k < WINDOW_SIZE; ++k) fill_preint(w->imu[k], k + 1 < WINDOW_SIZE ? e.pre_integrations[k + 1] : nullptr);
    for (auto& it : e.f_manager.feature) {
        w->id.push_back(it.feature_id); w->start.push_back(it.start_frame); w->nobs.push_back((int)it.feature_per_frame.size());
        w->flag.push_back(it.solve_flag); w->depth.push_back(it.estimated_depth);
        for (auto& f : it.feature_per_frame) {
            const double row[8] = {f.point.x(), f.point.y(), f.point.z(), f.uv.x(), f.uv.y(), f.velocity.x(), f.velocity.y(), f.cur_td};
            w->obs.insert(w->obs.end(), row, row + 8);
        }
    }
    vg_ba_problem& pb = w->prob;
    memset(&pb, 0, sizeof(pb));
    pb.K = K; pb.pose = w->pose.data(); pb.speedbias = w->sb.data(); pb.ex_pose = w->ex.data(); pb.td = e.td; pb.imu = w->imu.data();
    if (e.last_marginalization_info) {
        MarginalizationInfo* mi = e.last_marginalization_info;
        const int n = mi->n, nb = (int)mi->keep_block_size.size();
        for (int b = 0; b < nb; ++b) {
            int kind, index;
            if (!e.block_of(e.last_marginalization_parameter_blocks[b], kind, index)) throw std::runtime_error("prior block address outside the para_* arrays");
            w->pkind.push_back(kind); w->pindex.push_back(index);
            w->px0.insert(w->px0.end(), mi->keep_block_data[b], mi->keep_block_data[b] + mi->keep_block_size[b]);
        }
        w->pJ0.resize((size_t)n * n);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) w->pJ0[(size_t)r * n + c] = mi->linearized_jacobians(r, c);
        w->pr0.assign(mi->linearized_residuals.data(), mi->linearized_residuals.data() + n);
        pb.prior_n = n; pb.prior_nblocks = nb; pb.prior_block_kind = w->pkind.data(); pb.prior_block_index = w->pindex.data();
        pb.prior_J0 = w->pJ0.data(); pb.prior_r0 = w->pr0.data(); pb.prior_x0 = w->px0.data();
    }
    pb.estimate_extrinsic = ESTIMATE_EXTRINSIC ? 1 : 0; pb.estimate_td = ESTIMATE_TD ? 1 : 0; pb.max_iters = NUM_ITERATIONS;
    pb.focal = FOCAL_LENGTH_D; pb.tr = TR; pb.row = ROW_D; pb.g_norm = G_NORM; pb.max_solver_time_s = SOLVER_TIME;
    vg_ba_tracks& t = w->tracks;
    t.n_features = (int)w->id.size();
    t.feature_id = w->id.data(); t.start_frame = w->start.data(); t.n_obs = w->nobs.data(); t.solve_flag = w->flag.data();
    t.depth = w->depth.data(); t.obs = w->obs.data();
    // host mirror
    for (int k = 0; k < K; ++k) { o.Ps[k] = e.Ps[k]; o.Vs[k] = e.Vs[k]; o.Bas[k] = e.Bas[k]; o.Bgs[k] = e.Bgs[k]; o.Rs[k] = e.Rs[k]; }
    o.ric = e.ric[0]; o.tic = e.tic[0]; o.td = e.td;
    o.acc_0 = acc_0; o.gyr_0 = gyr_0; o.first_imu = true; o.g = Vector3d(0, 0, G_NORM);
    // pre_integrations[WINDOW_SIZE - 1] keeps its samples for a later merge; pre_integrations[WINDOW_SIZE] starts empty (:1037 / :1094)
    const IntegrationBase* p = e.pre_integrations[WINDOW_SIZE - 1];
    o.prev = Interval();
    if (p) {
        o.prev.linearized_acc = p->linearized_acc; o.prev.linearized_gyr = p->linearized_gyr;
        o.prev.linearized_ba = p->linearized_ba; o.prev.linearized_bg = p->linearized_bg;
        for (size_t s = 0; s < p->dt_buf.size(); ++s) {
            const double r[7] = {p->dt_buf[s], p->acc_buf[s](0), p->acc_buf[s](1), p->acc_buf[s](2), p->gyr_buf[s](0), p->gyr_buf[s](1), p->gyr_buf[s](2)};
            o.prev.samples.insert(o.prev.samples.end(), r, r + 7);
        }
    }
    o.cur = Interval();
    o.cur.linearized_acc = acc_0; o.cur.linearized_gyr = gyr_0; o.cur.linearized_ba = o.Bas[WINDOW_SIZE]; o.cur.linearized_bg = o.Bgs[WINDOW_SIZE];
    o.merge_pending = false; o.have_frame = false;
    o.last_R = o.Rs[WINDOW_SIZE]; o.last_P = o.Ps[WINDOW_SIZE]; o.last_R0 = o.Rs[0]; o.last_P0 = o.Ps[0];      // estimator.cpp:205-208
    o.failure_occur = false;
    return w;
}

void ResidentEstimators::handBack(int i, Estimator& e) {
    if (!begun_) throw std::runtime_error("handBack() before begin()");
    const int K = WINDOW_SIZE + 1;
    One& o = est_[i];
    std::vector<double> pose(7 * K), sb(9 * K), ex(7);
    double td = 0;
    std::vector<vg_imu_preint> imu(K - 1);
    const int cap = 6 * K + 32, capb = K + 8;
    std::vector<int> pkind(capb), pindex(capb);
    std::vector<double> pJ0((size_t)cap * cap), pr0(cap), px0(9 * capb);
    vg_ba_prior pr;
    memset(&pr, 0, sizeof(pr));
    pr.cap = cap; pr.cap_blocks = capb; pr.block_kind = pkind.data(); pr.block_index = pindex.data(); pr.J0 = pJ0.data(); pr.r0 = pr0.data(); pr.x0 = px0.data();
    if (vg_ba_seq_export(vg_, i, pose.data(), sb.data(), ex.data(), &td, imu.data(), &pr) != VG_OK) throw std::runtime_error(std::string("vg_ba_seq_export: ") + vg_last_error(vg_));
    for (int k = 0; k < K; ++k) {
        const double* x = &pose[7 * k];
        const double* s = &sb[9 * k];
        const double qn = std::sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
        e.Ps[k] = Vector3d(x[0], x[1], x[2]);
        e.Rs[k] = Quaterniond(x[6] / qn, x[3] / qn, x[4] / qn, x[5] / qn).toRotationMatrix();
        e.Vs[k] = Vector3d(s[0], s[1], s[2]); e.Bas[k] = Vector3d(s[3], s[4], s[5]); e.Bgs[k] = Vector3d(s[6], s[7], s[8]);
    }
    {
        const double qn = std::sqrt(ex[3] * ex[3] + ex[4] * ex[4] + ex[5] * ex[5] + ex[6] * ex[6]);
        e.tic[0] = Vector3d(ex[0], ex[1], ex[2]);
        e.ric[0] = Quaterniond(ex[6] / qn, ex[3] / qn, ex[4] / qn, ex[5] / qn).toRotationMatrix();
        e.td = td;
    }
    for (int j = 0; j <= WINDOW_SIZE; ++j) { delete e.pre_integrations[j]; e.pre_integrations[j] = nullptr; }
    if (o.merge_pending) {
        // the last frame was dropped as a non-keyframe: pre_integrations[WINDOW_SIZE - 1] has taken its samples (estimator.cpp:1069-1085)
        // but the device still holds the record of the shorter interval (the merged one travels with the NEXT frame): integrate it now
        const int ns = (int)o.prev.samples.size() / 7;
        const int off[2] = {0, ns};
        const double first[6] = {o.prev.linearized_acc.x(), o.prev.linearized_acc.y(), o.prev.linearized_acc.z(), o.prev.linearized_gyr.x(), o.prev.linearized_gyr.y(), o.prev.linearized_gyr.z()};
        const double bias[6] = {o.prev.linearized_ba.x(), o.prev.linearized_ba.y(), o.prev.linearized_ba.z(), o.prev.linearized_bg.x(), o.prev.linearized_bg.y(), o.prev.linearized_bg.z()};
        const double noise[4] = {ACC_N, GYR_N, ACC_W, GYR_W};
        if (vg_imu_preintegrate(vg_, 1, off, o.prev.samples.data(), first, bias, noise, &imu[K - 3]) != VG_OK)
            throw std::runtime_error(std::string("vg_imu_preintegrate: ") + vg_last_error(vg_));
    }
    for (int k = 0; k + 2 < K; ++k) {                               // imu[k] links frame k -> k + 1 = pre_integrations[k + 1]
        const vg_imu_preint& m = imu[k];
        // valid = 0 with a duration: the interval exceeded 10 s (the device marks it so that its factor is skipped, estimator.cpp:714).
        // The reference still HOLDS such a pre-integration -- optimization() tests sum_dt itself, slideWindow() merges into it -- so the
        // object is rebuilt with its sum_dt (and, for slot WINDOW_SIZE - 1, its samples); only a record that was never filled is absent.
        if (!m.valid && !(m.sum_dt > 0.0)) continue;
        IntegrationBase* p = new IntegrationBase();
        p->sum_dt = m.sum_dt;
        p->delta_p = Vector3d(m.delta_p[0], m.delta_p[1], m.delta_p[2]); p->delta_v = Vector3d(m.delta_v[0], m.delta_v[1], m.delta_v[2]);
        p->linearized_ba = Vector3d(m.linearized_ba[0], m.linearized_ba[1], m.linearized_ba[2]);
        p->linearized_bg = Vector3d(m.linearized_bg[0], m.linearized_bg[1], m.linearized_bg[2]);
        p->delta_q = Quaterniond(m.delta_q[3], m.delta_q[0], m.delta_q[1], m.delta_q[2]);
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c) { p->jacobian(r, c) = m.jacobian[r * 15 + c]; p->covariance(r, c) = m.covariance[r * 15 + c]; }
        if (k + 3 == K) {                                           // pre_integrations[WINDOW_SIZE - 1]: with the samples a later merge needs
            p->linearized_acc = o.prev.linearized_acc; p->linearized_gyr = o.prev.linearized_gyr;
            for (size_t q = 0; q + 6 < o.prev.samples.size(); q += 7) {
                p->dt_buf.push_back(o.prev.samples[q]);
                p->acc_buf.push_back(Vector3d(o.prev.samples[q + 1], o.prev.samples[q + 2], o.prev.samples[q + 3]));
                p->gyr_buf.push_back(Vector3d(o.prev.samples[q + 4], o.prev.samples[q + 5], o.prev.samples[q + 6]));
            }
        }
        e.pre_integrations[k + 1] = p;
    }
    // f_manager.feature
    int n = 0;
    std::vector<int> id(max_features_), start(max_features_), nobs(max_features_), flag(max_features_);
    std::vector<double> depth(max_features_), rows((size_t)max_features_ * K * 8);
    if (vg_ba_seq_get_tracks(vg_, i, max_features_, &n, id.data(), start.data(), nobs.data(), flag.data(), depth.data(), rows.data()) != VG_OK)
        throw std::runtime_error(std::string("vg_ba_seq_get_tracks: ") + vg_last_error(vg_));
    e.f_manager.feature.clear();
    for (int f = 0; f < n; ++f) {
        FeaturePerId it;
        it.feature_id = id[f]; it.start_frame = start[f]; it.estimated_depth = depth[f]; it.solve_flag = flag[f]; it.used_num = nobs[f];
        for (int j = 0; j < nobs[f]; ++j) {
            const double* r = &rows[((size_t)f * K + j) * 8];        // [x y u v vx vy cur_td z]
            FeaturePerFrame fr;
            fr.point = Vector3d(r[0], r[1], r[7]); fr.uv.x() = r[2]; fr.uv.y() = r[3]; fr.velocity.x() = r[4]; fr.velocity.y() = r[5]; fr.cur_td = r[6];
            it.feature_per_frame.push_back(fr);
        }
        e.f_manager.feature.push_back(it);
    }
    // last_marginalization_info + parameter blocks (what collectPrior() builds from a vg_ba_prior)
    delete e.last_marginalization_info;
    e.last_marginalization_info = nullptr;
    e.last_marginalization_parameter_blocks.clear();
    e.prior_pending = false;
    if (pr.valid && pr.n > 0) {
        MarginalizationInfo* mi = new MarginalizationInfo();
        mi->n = pr.n; mi->m = 0;
        mi->linearized_jacobians.resize(pr.n, pr.n);
        mi->linearized_residuals.resize(pr.n);
        for (int r = 0; r < pr.n; ++r) {
            mi->linearized_residuals(r) = pr0[r];
            for (int c = 0; c < pr.n; ++c) mi->linearized_jacobians(r, c) = pJ0[(size_t)r * pr.n + c];
        }
        int off = 0, x0o = 0;
        for (int b = 0; b < pr.nblocks; ++b) {
            const int kind = pkind[b], idx = pindex[b];
            const int gs = kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 7), ls = kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 6);
            mi->keep_block_size.push_back(gs);
            mi->keep_block_idx.push_back(off);
            double* d = new double[gs];
            memcpy(d, px0.data() + x0o, sizeof(double) * gs);
            mi->keep_block_data.push_back(d);
            e.last_marginalization_parameter_blocks.push_back(kind == VG_BLK_POSE ? e.para_Pose[idx] : kind == VG_BLK_SPEEDBIAS ? e.para_SpeedBias[idx]
                                                              : kind == VG_BLK_EXPOSE ? e.para_Ex_Pose[0] : e.para_Td[0]);
            off += ls; x0o += gs;
        }
        e.last_marginalization_info = mi;
    }
}

void ResidentEstimators::begin() {
    std::vector<const vg_ba_problem*> pb;
    std::vector<const vg_ba_tracks*> tr;
    for (Window* w : win_) {
        if (!w) throw std::runtime_error("begin(): an estimator has not been handed over");
        pb.push_back(&w->prob); tr.push_back(&w->tracks);
    }
    vg_ba_seq_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.max_features = max_features_; cfg.max_new_obs = max_new_obs_; cfg.init_depth = INIT_DEPTH; cfg.min_parallax = MIN_PARALLAX;
    if (vg_ba_seq_begin(vg_, size(), &cfg, pb.data(), tr.data()) != VG_OK) throw std::runtime_error(std::string("vg_ba_seq_begin: ") + vg_last_error(vg_));
    for (Window*& w : win_) { delete w; w = nullptr; }
    begun_ = true;
}

// Utility::deltaQ(theta).toRotationMatrix() (utility.h:16-28; Eigen's toRotationMatrix does not normalise)
static Matrix3d delta_R(const Vector3d& th) { return Quaterniond(1.0, th.x() / 2, th.y() / 2, th.z() / 2).toRotationMatrix(); }

void ResidentEstimators::processIMU(int i, double dt, const Vector3d& acc, const Vector3d& gyr) {
    One& o = est_[i];
    if (!o.first_imu) { o.first_imu = true; o.acc_0 = acc; o.gyr_0 = gyr; }
    const double r[7] = {dt, acc.x(), acc.y(), acc.z(), gyr.x(), gyr.y(), gyr.z()};
    o.cur.samples.insert(o.cur.samples.end(), r, r + 7);            // pre_integrations[frame_count]->push_back (:95): integrated in solve()
    const int j = WINDOW_SIZE;
    const Vector3d un_acc_0 = o.Rs[j] * (o.acc_0 - o.Bas[j]) - o.g;
    const Vector3d un_gyr = (o.gyr_0 + gyr) * 0.5 - o.Bgs[j];
    o.Rs[j] = o.Rs[j] * delta_R(un_gyr * dt);
    const Vector3d un_acc_1 = o.Rs[j] * (acc - o.Bas[j]) - o.g;
    const Vector3d un_acc = (un_acc_0 + un_acc_1) * 0.5;
    o.Ps[j] = o.Ps[j] + o.Vs[j] * dt + un_acc * (0.5 * dt * dt);
    o.Vs[j] = o.Vs[j] + un_acc * dt;
    o.acc_0 = acc; o.gyr_0 = gyr;
}

void ResidentEstimators::processImage(int i, const Image& image) {
    One& o = est_[i];
    o.ids.clear(); o.rows.clear();
    for (const auto& id_pts : image) {                              // ascending feature id, first camera (feature_manager.cpp:52-54)
        o.ids.push_back(id_pts.first);
        const auto& p = id_pts.second[0].second;
        for (int k = 0; k < 7; ++k) o.rows.push_back(p(k, 0));
    }
    o.have_frame = true;
}

void ResidentEstimators::solve() {
    if (!begun_) throw std::runtime_error("solve() before begin()");
    const int n = size(), K = WINDOW_SIZE + 1;
    // ---- the running intervals (and the merged ones) integrated in one batched call
    std::vector<int> off(1, 0), which;                              // interval q belongs to estimator which[q] / 2, odd = merged
    std::vector<double> smp, first, bias;
    auto add = [&](const Interval& it, int tag) {
        smp.insert(smp.end(), it.samples.begin(), it.samples.end());
        off.push_back((int)smp.size() / 7);
        const double f[6] = {it.linearized_acc.x(), it.linearized_acc.y(), it.linearized_acc.z(), it.linearized_gyr.x(), it.linearized_gyr.y(), it.linearized_gyr.z()};
        const double b[6] = {it.linearized_ba.x(), it.linearized_ba.y(), it.linearized_ba.z(), it.linearized_bg.x(), it.linearized_bg.y(), it.linearized_bg.z()};
        first.insert(first.end(), f, f + 6); bias.insert(bias.end(), b, b + 6);
        which.push_back(tag);
    };
    for (int i = 0; i < n; ++i) {
        if (!est_[i].have_frame) throw std::runtime_error("solve(): estimator " + std::to_string(i) + " has no frame");
        add(est_[i].cur, 2 * i);
        if (est_[i].merge_pending) add(est_[i].prev, 2 * i + 1);
    }
    if (smp.empty()) smp.resize(7, 0.0);
    std::vector<vg_imu_preint> rec(which.size());
    const double noise[4] = {ACC_N, GYR_N, ACC_W, GYR_W};
    if (vg_imu_preintegrate(vg_, (int)which.size(), off.data(), smp.data(), first.data(), bias.data(), noise, rec.data()) != VG_OK)
        throw std::runtime_error(std::string("vg_imu_preintegrate: ") + vg_last_error(vg_));
    std::vector<vg_ba_frame> fr(n);
    std::vector<const vg_ba_frame*> frp(n);
    for (int i = 0; i < n; ++i) { memset(&fr[i], 0, sizeof(vg_ba_frame)); frp[i] = &fr[i]; }
    for (size_t q = 0; q < which.size(); ++q) {
        vg_ba_frame& f = fr[which[q] / 2];
        if (which[q] & 1) f.imu_merged = &rec[q]; else f.imu_new = &rec[q];
    }
    for (int i = 0; i < n; ++i) {
        One& o = est_[i];
        vg_ba_frame& f = fr[i];
        const Quaterniond q(o.Rs[WINDOW_SIZE]);
        const double qn = std::sqrt(q.w() * q.w() + q.x() * q.x() + q.y() * q.y() + q.z() * q.z());
        const double pose[7] = {o.Ps[WINDOW_SIZE].x(), o.Ps[WINDOW_SIZE].y(), o.Ps[WINDOW_SIZE].z(), q.x() / qn, q.y() / qn, q.z() / qn, q.w() / qn};
        memcpy(f.pose, pose, sizeof(pose));
        const double sb[9] = {o.Vs[WINDOW_SIZE].x(), o.Vs[WINDOW_SIZE].y(), o.Vs[WINDOW_SIZE].z(), o.Bas[WINDOW_SIZE].x(), o.Bas[WINDOW_SIZE].y(),
                              o.Bas[WINDOW_SIZE].z(), o.Bgs[WINDOW_SIZE].x(), o.Bgs[WINDOW_SIZE].y(), o.Bgs[WINDOW_SIZE].z()};
        memcpy(f.speedbias, sb, sizeof(sb));
        f.n_obs = (int)o.ids.size(); f.feature_id = o.ids.data(); f.obs = o.rows.data();
    }
    if (vg_ba_seq_step_async(vg_, n, frp.data()) != VG_OK) throw std::runtime_error(std::string("vg_ba_seq_step_async: ") + vg_last_error(vg_));
    // ---- states of the solved windows, the key-frame decisions
    std::vector<double> pose((size_t)n * 7 * K), sb((size_t)n * 9 * K), ex((size_t)n * 7), td(n);
    std::vector<vg_ba_state> st(n);
    std::vector<vg_ba_state*> stp(n);
    std::vector<vg_ba_summary> sum(n);
    for (int i = 0; i < n; ++i) {
        memset(&st[i], 0, sizeof(vg_ba_state));
        st[i].pose = &pose[(size_t)i * 7 * K]; st[i].speedbias = &sb[(size_t)i * 9 * K]; st[i].ex_pose = &ex[(size_t)i * 7]; st[i].td = &td[i];
        stp[i] = &st[i];
    }
    const int rc = vg_ba_batch_download_state(vg_, n, stp.data(), sum.data());
    if (rc != VG_OK && rc != VG_ERR_NUMERIC) throw std::runtime_error(std::string("vg_ba_batch_download_state: ") + vg_last_error(vg_));
    std::vector<int> info((size_t)n * VG_SEQ_INFO_INTS);
    if (vg_ba_seq_info(vg_, n, info.data()) != VG_OK) throw std::runtime_error(std::string("vg_ba_seq_info: ") + vg_last_error(vg_));
    for (int i = 0; i < n; ++i) {
        One& o = est_[i];
        const int* nf = &info[(size_t)i * VG_SEQ_INFO_INTS];
        o.last_summary = sum[i];
        o.status = nf[VG_SEQ_STATUS]; o.n_features = nf[VG_SEQ_N_AFTER];
        o.marginalization_flag = nf[VG_SEQ_FLAG] == VG_MARGIN_OLD ? Estimator::MARGIN_OLD : Estimator::MARGIN_SECOND_NEW;
        // double2vector() + slideWindow() on the mirror (estimator.cpp:530-565, :1010-1050 / :1086-1099)
        Vector3d P[WINDOW_SIZE + 1], V[WINDOW_SIZE + 1], Ba[WINDOW_SIZE + 1], Bg[WINDOW_SIZE + 1];
        Matrix3d R[WINDOW_SIZE + 1];
        for (int k = 0; k < K; ++k) {
            const double* x = st[i].pose + 7 * k;
            const double* s = st[i].speedbias + 9 * k;
            const double qn = std::sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
            P[k] = Vector3d(x[0], x[1], x[2]);
            R[k] = Quaterniond(x[6] / qn, x[3] / qn, x[4] / qn, x[5] / qn).toRotationMatrix();
            V[k] = Vector3d(s[0], s[1], s[2]); Ba[k] = Vector3d(s[3], s[4], s[5]); Bg[k] = Vector3d(s[6], s[7], s[8]);
        }
        o.last_track_num = nf[VG_SEQ_N_TRACKED];
        {
            // failureDetection() on the solved window (estimator.cpp:621-667; the commented-out returns stay commented out)
            auto norm = [](const Vector3d& v) { return std::sqrt(v.x() * v.x() + v.y() * v.y() + v.z() * v.z()); };
            const Vector3d dP = P[WINDOW_SIZE] - o.last_P;
            o.failure_occur = sum[i].status != VG_OK || norm(Ba[WINDOW_SIZE]) > 2.5 || norm(Bg[WINDOW_SIZE]) > 1.0 || norm(dP) > 5 || std::fabs(dP.z()) > 1;
        }
        for (int k = 0; k < K; ++k) {
            int from;
            if (o.marginalization_flag == Estimator::MARGIN_OLD) from = k < K - 1 ? k + 1 : K - 1;
            else from = k <= K - 3 ? k : K - 1;
            o.Ps[k] = P[from]; o.Rs[k] = R[from]; o.Vs[k] = V[from]; o.Bas[k] = Ba[from]; o.Bgs[k] = Bg[from];
        }
        o.last_R = o.Rs[WINDOW_SIZE]; o.last_P = o.Ps[WINDOW_SIZE]; o.last_R0 = o.Rs[0]; o.last_P0 = o.Ps[0];      // :205-208
        {
            const double* e = st[i].ex_pose;
            const double qn = std::sqrt(e[3] * e[3] + e[4] * e[4] + e[5] * e[5] + e[6] * e[6]);
            o.tic = Vector3d(e[0], e[1], e[2]);
            o.ric = Quaterniond(e[6] / qn, e[3] / qn, e[4] / qn, e[5] / qn).toRotationMatrix();
            o.td = td[i];
        }
        // pre-integrations: MARGIN_OLD: the running interval becomes pre_integrations[WINDOW_SIZE - 1]; MARGIN_SECOND_NEW: its samples
        // are appended to pre_integrations[WINDOW_SIZE - 1] (:1069-1085), which is re-integrated for the next step
        if (o.marginalization_flag == Estimator::MARGIN_OLD) { o.prev = o.cur; o.merge_pending = false; }
        else { o.prev.samples.insert(o.prev.samples.end(), o.cur.samples.begin(), o.cur.samples.end()); o.merge_pending = true; }
        o.cur = Interval();                                         // new IntegrationBase{acc_0, gyr_0, Bas[WINDOW_SIZE], Bgs[WINDOW_SIZE]} (:1037 / :1094)
        o.cur.linearized_acc = o.acc_0; o.cur.linearized_gyr = o.gyr_0; o.cur.linearized_ba = o.Bas[WINDOW_SIZE]; o.cur.linearized_bg = o.Bgs[WINDOW_SIZE];
        o.have_frame = false;
    }
}
