// replay_main.cpp — ROS-free replay harness for the drop-in classes (SURVEY.md 8(d) configs[0] substitute).
//
//   vins_replay fe <frames.bin> <out.txt>   frames.bin = int32 n, w, h, pub_every ; n * w*h bytes
//     feeds the frames through FeatureTracker::readImage exactly as img_callback does (feature_tracker_node.cpp:86-111:
//     readImage, then updateID for every feature) and dumps, per frame, ids / cur_pts / track_cnt / cur_un_pts / velocity.
//
//   vins_replay ba <sequence.bin> <out.csv>
//     N consecutive sliding windows through the drop-in Estimator the way process() drives it once the system is
//     initialised (estimator.cpp:120-170): IMU pre-integration of the new frame interval (here: vg_imu_preintegrate on
//     the device instead of processIMU's sample-by-sample push_back), Estimator::optimization() (solve + MARGIN_OLD
//     marginalization, prior carried over through last_marginalization_info / ..._parameter_blocks) and
//     Estimator::slideWindow() (state shift + removeBackShiftDepth).  Writes one line per window in the format of
//     pubOdometry's result file (utility/visualization.cpp:157-172): stamp[ns], P, Q(w x y z), V of frame WINDOW_SIZE.
//     sequence.bin is written by tests/replay_util.py.
//
//   vins_replay seq <frames.bin> <out.csv>
//     N estimators whose windows stay ON THE DEVICE (ResidentEstimators, resident_estimator.h): the file holds, per estimator, the
//     window as it stands between two frames (states, the IMU samples of its intervals, every track) and then W frames at the level
//     of the node's callbacks: the IMU samples since the last frame (processIMU) and the `image` map (processImage).  One line per
//     frame and estimator: index, stamp[ns], P, Q(w x y z), V of frame WINDOW_SIZE as solved, the key-frame decision, tracks left.
//
//   vins_replay vio <window.bin> <frames.bin> <out.csv>
//     Both drop-ins in one process, wired the way the two nodes are (feature_tracker_node.cpp:86-160 publishes id / undistorted
//     point / pixel / velocity of every feature with track_cnt > 1; estimator_node.cpp:286-306 turns that message into the `image`
//     map of processImage): `seq` for one estimator whose tracks are NOT in the file (window.bin holds states and IMU samples only,
//     L = 0 and n = 0) but come from FeatureTracker::readImage over frames.bin -- the first WINDOW_SIZE frames fill the window that
//     is handed over, every further frame is one processImage + solve.  tests/e2e_vio.py renders the frames.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <vector>
#include "estimator.h"
#include "feature_tracker.h"
#include "resident_estimator.h"

static int replay_fe(const char* in, const char* out) {
    FILE* f = fopen(in, "rb");
    if (!f) { perror("frames"); return 2; }
    int hdr[4];
    if (fread(hdr, sizeof(int), 4, f) != 4) return 2;
    const int n = hdr[0], w = hdr[1], h = hdr[2], pub_every = hdr[3];
    COL = w; ROW = h;
    std::vector<unsigned char> buf((size_t)w * h);
    FeatureTracker tracker;
    FILE* o = fopen(out, "w");
    for (int k = 0; k < n; ++k) {
        if (fread(buf.data(), 1, buf.size(), f) != buf.size()) return 2;
        PUB_THIS_FRAME = (k % pub_every) == 0;
        cv::Mat img(h, w, cv::CV_8UC1, buf.data(), (size_t)w);
        tracker.readImage(img, 0.05 * k);
        for (unsigned int i = 0;; i++) if (!tracker.updateID(i)) break;
        fprintf(o, "frame %d n %zu\n", k, tracker.cur_pts.size());
        for (size_t i = 0; i < tracker.cur_pts.size(); ++i)
            fprintf(o, "%d %d %.9g %.9g %.9g %.9g %.9g %.9g\n", tracker.ids[i], tracker.track_cnt[i], tracker.cur_pts[i].x, tracker.cur_pts[i].y,
                    tracker.cur_un_pts[i].x, tracker.cur_un_pts[i].y, tracker.pts_velocity[i].x, tracker.pts_velocity[i].y);
    }
    fclose(o);
    fclose(f);
    return 0;
}

namespace {
struct Reader {
    FILE* f;
    void d(double* p, int n) { if (fread(p, sizeof(double), n, f) != (size_t)n) throw std::runtime_error("sequence file truncated"); }
    void i(int* p, int n) { if (fread(p, sizeof(int), n, f) != (size_t)n) throw std::runtime_error("sequence file truncated"); }
};
struct FrameInit { double t, pose[7], sb[9]; };

void set_frame(Estimator& e, int i, const FrameInit& fr) {
    e.Ps[i] = Vector3d(fr.pose[0], fr.pose[1], fr.pose[2]);
    e.Rs[i] = Quaterniond(fr.pose[6], fr.pose[3], fr.pose[4], fr.pose[5]).toRotationMatrix();
    e.Vs[i] = Vector3d(fr.sb[0], fr.sb[1], fr.sb[2]);
    e.Bas[i] = Vector3d(fr.sb[3], fr.sb[4], fr.sb[5]);
    e.Bgs[i] = Vector3d(fr.sb[6], fr.sb[7], fr.sb[8]);
}

// one frame interval: (acc_0, gyr_0) + S samples (dt, acc, gyr) -> IntegrationBase on the device
IntegrationBase* preintegrate(vg_handle* h, Reader& rd, int S, const double* bias, const double* noise) {
    double first[6];
    rd.d(first, 6);
    std::vector<double> smp((size_t)S * 7);
    rd.d(smp.data(), S * 7);
    const int off[2] = {0, S};
    vg_imu_preint out;
    if (vg_imu_preintegrate(h, 1, off, smp.data(), first, bias, noise, &out) != VG_OK) throw std::runtime_error(vg_last_error(h));
    IntegrationBase* p = new IntegrationBase();
    p->linearized_acc = Vector3d(first[0], first[1], first[2]);          // raw samples: slideWindow(MARGIN_SECOND_NEW) merges intervals
    p->linearized_gyr = Vector3d(first[3], first[4], first[5]);
    for (int i = 0; i < S; ++i) {
        p->dt_buf.push_back(smp[7 * i]);
        p->acc_buf.push_back(Vector3d(smp[7 * i + 1], smp[7 * i + 2], smp[7 * i + 3]));
        p->gyr_buf.push_back(Vector3d(smp[7 * i + 4], smp[7 * i + 5], smp[7 * i + 6]));
    }
    p->sum_dt = out.sum_dt;
    p->delta_p = Vector3d(out.delta_p[0], out.delta_p[1], out.delta_p[2]);
    p->delta_v = Vector3d(out.delta_v[0], out.delta_v[1], out.delta_v[2]);
    p->linearized_ba = Vector3d(out.linearized_ba[0], out.linearized_ba[1], out.linearized_ba[2]);
    p->linearized_bg = Vector3d(out.linearized_bg[0], out.linearized_bg[1], out.linearized_bg[2]);
    p->delta_q = Quaterniond(out.delta_q[3], out.delta_q[0], out.delta_q[1], out.delta_q[2]);
    for (int r = 0; r < 15; ++r)
        for (int c = 0; c < 15; ++c) { p->jacobian(r, c) = out.jacobian[r * 15 + c]; p->covariance(r, c) = out.covariance[r * 15 + c]; }
    return p;
}
}  // namespace

static int replay_ba(const char* in, const char* out) {
    Reader rd{fopen(in, "rb")};
    if (!rd.f) { perror("sequence"); return 2; }
    int hdr[4];
    rd.i(hdr, 4);
    if (hdr[0] != 0x31414256) { fprintf(stderr, "not a VBA1 sequence file\n"); return 2; }
    const int W = hdr[1], K = hdr[2], S = hdr[3];
    if (K != WINDOW_SIZE + 1) { fprintf(stderr, "sequence has K = %d, the Estimator is built for %d\n", K, WINDOW_SIZE + 1); return 2; }
    double ex[7], noise[4], par[2], bias[6];
    rd.d(ex, 7); rd.d(noise, 4); rd.d(par, 2); rd.d(bias, 6);
    G_NORM = par[0]; FOCAL_LENGTH_D = par[1]; ESTIMATE_EXTRINSIC = 0; ESTIMATE_TD = 0; NUM_ITERATIONS = 8;
    ACC_N = noise[0]; GYR_N = noise[1]; ACC_W = noise[2]; GYR_W = noise[3];
    vg_handle* h = nullptr;
    if (vg_create(&h) != VG_OK) { fprintf(stderr, "vg_create failed (no CPU fallback)\n"); return 3; }
    Estimator est;
    est.tic[0] = Vector3d(ex[0], ex[1], ex[2]);
    est.ric[0] = Quaterniond(ex[6], ex[3], ex[4], ex[5]).toRotationMatrix();
    std::vector<double> stamp(K);
    for (int i = 0; i < K; ++i) {
        FrameInit fr;
        rd.d(&fr.t, 1); rd.d(fr.pose, 7); rd.d(fr.sb, 9);
        set_frame(est, i, fr);
        stamp[i] = fr.t;
    }
    for (int i = 0; i + 1 < K; ++i) est.pre_integrations[i + 1] = preintegrate(h, rd, S, bias, noise);
    FILE* o = fopen(out, "w");
    for (int w = 0; w < W; ++w) {
        if (w > 0) {
            est.marginalization_flag = Estimator::MARGIN_OLD;
            est.slideWindow();                                           // (deletes the IntegrationBase that rotated out, like the reference)
            for (int i = 0; i + 1 < K; ++i) stamp[i] = stamp[i + 1];
            FrameInit fr;
            rd.d(&fr.t, 1); rd.d(fr.pose, 7); rd.d(fr.sb, 9);
            set_frame(est, WINDOW_SIZE, fr);
            stamp[K - 1] = fr.t;
            est.pre_integrations[WINDOW_SIZE] = preintegrate(h, rd, S, bias, noise);
        }
        // features of this window: tracks from the file, depths carried over by removeBackShiftDepth where the track existed
        std::map<int, double> carried;
        for (auto& f : est.f_manager.feature) carried[f.feature_id] = f.estimated_depth;
        est.f_manager.feature.clear();
        int L;
        rd.i(&L, 1);
        for (int l = 0; l < L; ++l) {
            int meta[3];
            double init_inv_depth;
            rd.i(meta, 3);
            rd.d(&init_inv_depth, 1);
            FeaturePerId f;
            f.feature_id = meta[0]; f.start_frame = meta[1];
            for (int k = 0; k < meta[2]; ++k) {
                double r[7];
                rd.d(r, 7);
                FeaturePerFrame fr;
                fr.point = Vector3d(r[0], r[1], 1.0); fr.uv.x() = r[2]; fr.uv.y() = r[3]; fr.velocity.x() = r[4]; fr.velocity.y() = r[5]; fr.cur_td = r[6];
                f.feature_per_frame.push_back(fr);
            }
            auto it = carried.find(f.feature_id);
            f.estimated_depth = it != carried.end() ? it->second : 1.0 / init_inv_depth;
            est.f_manager.feature.push_back(f);
        }
        est.marginalization_flag = Estimator::MARGIN_OLD;
        est.optimization();
        const Quaterniond q(est.Rs[WINDOW_SIZE]);
        fprintf(o, "%.0f,%.5f,%.5f,%.5f,%.5f,%.5f,%.5f,%.5f,%.5f,%.5f,%.5f,\n", stamp[K - 1] * 1e9, est.Ps[WINDOW_SIZE].x(), est.Ps[WINDOW_SIZE].y(),
                est.Ps[WINDOW_SIZE].z(), q.w(), q.x(), q.y(), q.z(), est.Vs[WINDOW_SIZE].x(), est.Vs[WINDOW_SIZE].y(), est.Vs[WINDOW_SIZE].z());
    }
    for (int k = 0; k <= WINDOW_SIZE; ++k) { delete est.pre_integrations[k]; est.pre_integrations[k] = nullptr; }   // (the harness owns them, like the reference's clearState)
    fclose(o);
    fclose(rd.f);
    vg_destroy(h);
    return 0;
}

namespace {
// the front end of `vio`: one frame through readImage + updateID, returned as the `image` map (features seen at least twice)
struct FrontEnd {
    FILE* f = nullptr;
    int n = 0, w = 0, h = 0, k = 0;
    std::vector<unsigned char> buf;
    std::unique_ptr<FeatureTracker> tracker;
    bool open(const char* path) {
        f = fopen(path, "rb");
        int hdr[4];
        if (!f || fread(hdr, sizeof(int), 4, f) != 4) return false;
        n = hdr[0]; w = hdr[1]; h = hdr[2];
        COL = w; ROW = h;
        buf.resize((size_t)w * h);
        tracker.reset(new FeatureTracker());
        return true;
    }
    ResidentEstimators::Image next() {
        if (k >= n || fread(buf.data(), 1, buf.size(), f) != buf.size()) throw std::runtime_error("frames file exhausted");
        PUB_THIS_FRAME = true;
        cv::Mat img(h, w, cv::CV_8UC1, buf.data(), (size_t)w);
        tracker->readImage(img, 0.05 * k++);
        for (unsigned int i = 0;; i++) if (!tracker->updateID(i)) break;
        ResidentEstimators::Image image;
        for (size_t i = 0; i < tracker->cur_pts.size(); ++i) {
            if (tracker->track_cnt[i] <= 1) continue;
            Eigen::Matrix<double, 7, 1> p;
            p(0, 0) = tracker->cur_un_pts[i].x; p(1, 0) = tracker->cur_un_pts[i].y; p(2, 0) = 1.0;
            p(3, 0) = tracker->cur_pts[i].x; p(4, 0) = tracker->cur_pts[i].y;
            p(5, 0) = tracker->pts_velocity[i].x; p(6, 0) = tracker->pts_velocity[i].y;
            image[tracker->ids[i]].emplace_back(0, p);
        }
        return image;
    }
    ~FrontEnd() { if (f) fclose(f); }
};
}  // namespace

static int replay_seq(const char* in, const char* out, const char* frames = nullptr) {
    Reader rd{fopen(in, "rb")};
    if (!rd.f) { perror("frames"); return 2; }
    FrontEnd fe;
    if (frames && !fe.open(frames)) { perror("images"); return 2; }
    int hdr[5];
    rd.i(hdr, 5);
    if (hdr[0] != 0x31515356) { fprintf(stderr, "not a VSQ1 file\n"); return 2; }
    const int N = hdr[1], W = hdr[2], K = hdr[3], S = hdr[4];
    if (K != WINDOW_SIZE + 1) { fprintf(stderr, "file has K = %d, the Estimator is built for %d\n", K, WINDOW_SIZE + 1); return 2; }
    double noise[4], par[4];
    rd.d(noise, 4); rd.d(par, 4);
    G_NORM = par[0]; FOCAL_LENGTH_D = par[1]; MIN_PARALLAX = par[2]; INIT_DEPTH = par[3];
    ESTIMATE_EXTRINSIC = 0; ESTIMATE_TD = 0; NUM_ITERATIONS = 8;
    ACC_N = noise[0]; GYR_N = noise[1]; ACC_W = noise[2]; GYR_W = noise[3];
    vg_handle* h = nullptr;
    if (vg_create(&h) != VG_OK) { fprintf(stderr, "vg_create failed (no CPU fallback)\n"); return 3; }
    std::unique_ptr<ResidentEstimators> resp(new ResidentEstimators(N, 512, 512));
    // VINS_REPLAY_HANDBACK=<frame>: after that frame every window is handed back into a host Estimator (handBack), a NEW batch
    // takes them over (handOver + begin) and estimator 0 is re-seeded once more in place (reseed): the rest of the run must not notice
    const char* hb_env = getenv("VINS_REPLAY_HANDBACK");
    const int handback_frame = hb_env ? atoi(hb_env) : -1;
    for (int i = 0; i < N; ++i) {
        // the estimator as the reference's code would hold it between two frames, then handed over
        Estimator est;
        double ex[7], bias[6], last[6];
        rd.d(ex, 7); rd.d(bias, 6);
        est.tic[0] = Vector3d(ex[0], ex[1], ex[2]);
        est.ric[0] = Quaterniond(ex[6], ex[3], ex[4], ex[5]).toRotationMatrix();
        for (int k = 0; k < K; ++k) {
            FrameInit fr;
            rd.d(&fr.t, 1); rd.d(fr.pose, 7); rd.d(fr.sb, 9);
            set_frame(est, k, fr);
        }
        for (int k = 0; k + 2 < K; ++k) est.pre_integrations[k + 1] = preintegrate(h, rd, S, bias, noise);
        rd.d(last, 6);
        int L;
        rd.i(&L, 1);
        for (int l = 0; l < L; ++l) {
            int meta[3];
            double depth;
            rd.i(meta, 3); rd.d(&depth, 1);
            FeaturePerId f;
            f.feature_id = meta[0]; f.start_frame = meta[1]; f.estimated_depth = depth;
            for (int k = 0; k < meta[2]; ++k) {
                double r[8];
                rd.d(r, 8);
                FeaturePerFrame fr;
                fr.point = Vector3d(r[0], r[1], r[2]); fr.uv.x() = r[3]; fr.uv.y() = r[4]; fr.velocity.x() = r[5]; fr.velocity.y() = r[6]; fr.cur_td = r[7];
                f.feature_per_frame.push_back(fr);
            }
            est.f_manager.feature.push_back(f);
        }
        if (frames) {
            // the window's tracks from the front end: frame k of the window = image k (what addFeatureCheckParallax would have
            // collected, feature_manager.cpp:45-73, before the first solve of this replay)
            if (N != 1 || L != 0) { fprintf(stderr, "vio: window.bin must hold one estimator without tracks\n"); return 2; }
            std::map<int, FeaturePerId*> at;            // list nodes stay where they are
            for (int k = 0; k < WINDOW_SIZE; ++k) {
                const ResidentEstimators::Image image = fe.next();
                for (const auto& id_pts : image) {
                    auto it = at.find(id_pts.first);
                    if (it == at.end()) {
                        FeaturePerId f;
                        f.feature_id = id_pts.first; f.start_frame = k; f.estimated_depth = -1.0;
                        est.f_manager.feature.push_back(f);
                        it = at.emplace(id_pts.first, &est.f_manager.feature.back()).first;
                    }
                    const Eigen::Matrix<double, 7, 1>& r = id_pts.second[0].second;
                    FeaturePerFrame fr;
                    fr.point = Vector3d(r(0, 0), r(1, 0), r(2, 0)); fr.uv.x() = r(3, 0); fr.uv.y() = r(4, 0);
                    fr.velocity.x() = r(5, 0); fr.velocity.y() = r(6, 0); fr.cur_td = 0.0;
                    it->second->feature_per_frame.push_back(fr);
                }
            }
        }
        resp->handOver(i, est, Vector3d(last[0], last[1], last[2]), Vector3d(last[3], last[4], last[5]));
        for (int k = 0; k <= WINDOW_SIZE; ++k) { delete est.pre_integrations[k]; est.pre_integrations[k] = nullptr; }
    }
    resp->begin();
    FILE* o = fopen(out, "w");
    std::vector<double> smp((size_t)S * 7);
    double fe_ms = 0;
    for (int w = 0; w < W; ++w) {
        std::vector<double> stamp(N);
        for (int i = 0; i < N; ++i) {
            rd.d(&stamp[i], 1);
            rd.d(smp.data(), S * 7);
            for (int s = 0; s < S; ++s)
                resp->processIMU(i, smp[7 * s], Vector3d(smp[7 * s + 1], smp[7 * s + 2], smp[7 * s + 3]), Vector3d(smp[7 * s + 4], smp[7 * s + 5], smp[7 * s + 6]));
            int n;
            rd.i(&n, 1);
            ResidentEstimators::Image image;
            for (int k = 0; k < n; ++k) {
                int id;
                double r[7];
                rd.i(&id, 1); rd.d(r, 7);
                Eigen::Matrix<double, 7, 1> p;
                for (int c = 0; c < 7; ++c) p(c, 0) = r[c];
                image[id].emplace_back(0, p);
            }
            if (frames) {
                const auto t0 = std::chrono::steady_clock::now();
                image = fe.next();
                fe_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            resp->processImage(i, image);
        }
        const auto t_solve = std::chrono::steady_clock::now();
        resp->solve();
        const double solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_solve).count();
        for (int i = 0; i < N; ++i) {
            // (the mirror is post-slide: the frame just solved sits in slot WINDOW_SIZE either way)
            const ResidentEstimators::One& e = (*resp)[i];
            const Quaterniond q(e.Rs[WINDOW_SIZE]);
            fprintf(o, "%d,%.0f,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,%d,%d,%d,%d\n", i, stamp[i] * 1e9, e.Ps[WINDOW_SIZE].x(), e.Ps[WINDOW_SIZE].y(),
                    e.Ps[WINDOW_SIZE].z(), q.w(), q.x(), q.y(), q.z(), e.Vs[WINDOW_SIZE].x(), e.Vs[WINDOW_SIZE].y(), e.Vs[WINDOW_SIZE].z(),
                    (int)e.marginalization_flag, e.n_features, e.status, e.failure_occur ? 1 : 0);
            if (frames) fprintf(o, "t,%d,%.3f,%.3f\n", i, fe_ms, solve_ms);      // vio: wall time of readImage / of processImage's solve, ms
        }
        if (w == handback_frame) {
            std::unique_ptr<ResidentEstimators> next(new ResidentEstimators(N, 512, 512));
            std::vector<std::unique_ptr<Estimator>> back;
            for (int i = 0; i < N; ++i) {
                back.emplace_back(new Estimator());
                resp->handBack(i, *back[i]);
                next->handOver(i, *back[i], (*resp)[i].acc_0, (*resp)[i].gyr_0);
            }
            next->begin();
            next->reseed(0, *back[0], (*resp)[0].acc_0, (*resp)[0].gyr_0);
            for (auto& e : back)
                for (int k = 0; k <= WINDOW_SIZE; ++k) { delete e->pre_integrations[k]; e->pre_integrations[k] = nullptr; }
            resp = std::move(next);
        }
    }
    fclose(o);
    fclose(rd.f);
    vg_destroy(h);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "fe")) return replay_fe(argv[2], argv[3]);
    if (argc >= 4 && !strcmp(argv[1], "ba")) {
        try { return replay_ba(argv[2], argv[3]); }
        catch (const std::exception& e) { fprintf(stderr, "vins_replay ba: %s\n", e.what()); return 1; }
    }
    if (argc >= 4 && !strcmp(argv[1], "seq")) {
        try { return replay_seq(argv[2], argv[3]); }
        catch (const std::exception& e) { fprintf(stderr, "vins_replay seq: %s\n", e.what()); return 1; }
    }
    if (argc >= 5 && !strcmp(argv[1], "vio")) {
        try { return replay_seq(argv[2], argv[4], argv[3]); }
        catch (const std::exception& e) { fprintf(stderr, "vins_replay vio: %s\n", e.what()); return 1; }
    }
    fprintf(stderr, "usage: vins_replay fe <frames.bin> <out.txt> | vins_replay ba <sequence.bin> <out.csv> | vins_replay seq <frames.bin> <out.csv> | vins_replay vio <window.bin> <frames.bin> <out.csv>\n");
    return 2;
}
