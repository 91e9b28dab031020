// feature_tracker_readimage.cpp — THE drop-in for the front end: four member functions of the REFERENCE'S `class FeatureTracker`
// (feature_tracker/src/feature_tracker.h:28-65), on the MI355X path:
//     void FeatureTracker::readImage(const cv::Mat&, double)    feature_tracker.cpp:81-167
//     void FeatureTracker::setMask()                            feature_tracker.cpp:36-69
//     void FeatureTracker::rejectWithF()                        feature_tracker.cpp:169-202
//     void FeatureTracker::undistortedPoints()                  feature_tracker.cpp:258-306
//
// How it is used.  This file includes the reference's own "feature_tracker.h" — `camodocal::CameraPtr m_camera`, the cv::Mat
// members, `showUndistortion`, the globals of parameters.h all stay what they are — and defines exactly those four members.  In a
// catkin workspace: add this file and include/vinsgpu.h to feature_tracker, link libvinsgpu.so, and remove the four definitions
// from feature_tracker.cpp (or weaken them:  objcopy --weaken-symbol=_ZN14FeatureTracker9readImageERKN2cv3MatEd
// --weaken-symbol=_ZN14FeatureTracker7setMaskEv --weaken-symbol=_ZN14FeatureTracker11rejectWithFEv
// --weaken-symbol=_ZN14FeatureTracker17undistortedPointsEv).  feature_tracker_node.cpp, parameters.cpp, addPoints / updateID /
// readIntrinsicParameter / showUndistortion / inBorder / reduceVector and all of camera_model stay the reference's.  Here (no
// OpenCV / ROS in the image) oracle/Makefile target `ref_fe` does precisely that against the header stand-ins of
// oracle/ref_stubs_fe and produces oracle/_ref/libvins_ref_fe_gpu.so; tests/test_fe_dropin_*.py drive the reference's
// img_callback() through it and through the all-reference build side by side.
//
// readImage() itself is ONE library call per frame, vg_fe_read_image (include/vinsgpu.h): frame and cur_pts go up in one block, the
// device runs :87-125 (CLAHE, pyramid, LK, border test, reduceVector) and, on a published frame, rejectWithF, setMask's walk, the
// detection, addPoints and the lifting of undistortedPoints without the host in between; the results come back in one block.  The one
// question a published frame still puts to the host is the ORDER of setMask's walk: the reference's std::sort by track_cnt (:48) is not
// stable, so the order among equal counts is whatever this translation unit's std::sort makes of the sequence -- the callback below runs
// that very sort call and hands the permutation down.  A camera that is not a camodocal::PinholeCamera takes the step-by-step members
// (its lifting runs on the host).
//
// What the step-by-step bodies (setMask / rejectWithF / undistortedPoints as members, and readImage for other camera models) do instead
// of the reference's:
//   :87-93   cv::createCLAHE(3.0, Size(8,8))->apply          -> vg_fe_push_frames(.., EQUALIZE): upload + CLAHE + pyramid on the device
//   :113     cv::calcOpticalFlowPyrLK(cur_img, forw_img, ..)  -> vg_fe_track (the pyramid of cur_img is still on the device)
//   :149     cv::goodFeaturesToTrack(forw_img, .., mask)      -> vg_fe_detect_masked (the mask setMask() built stays on the device)
//   :191     cv::findFundamentalMat(.., FM_RANSAC, ..)        -> vg_fe_reject_with_f
//   :48-68   std::sort by track_cnt + mask walk + cv::circle  -> the SAME std::sort call on the host (the order among equal counts is
//                                                                whatever the platform's std::sort gives: the reference's too), then
//                                                                vg_fe_set_mask walks that order on the device
//   :262-267 m_camera->liftProjective per point               -> vg_fe_undistort for a camodocal::PinholeCamera (the EuRoC / default
//                                                                model); any other camera model lifts through m_camera on the host
// Differences a caller can see: `forw_img` / `cur_img` / `prev_img` refer to the image as it came in (the equalized image lives on
// the device: vg_fe_get_level(.., 0, ..) returns it) and `mask` is left empty (vg_fe_get_mask).  The node reads neither.
// State the reference class has no member for (the device handle) lives in a side table keyed by `this`.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "feature_tracker.h"
#include "vinsgpu.h"

namespace {

struct FeSide {
    vg_handle* vg = nullptr;
    int capacity = 0, width = 0, height = 0;
    bool pinhole = false;          // m_camera is a camodocal::PinholeCamera: lifting runs on the device
    double intr[8];                // fx fy cx cy k1 k2 p1 p2
    const void* camera = nullptr;  // the camera object `intr` was read from
    std::vector<uint8_t> base_copy;   // the fisheye mask without row padding, when the cv::Mat has some
    int stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // vins_fe_gpu_stats
};
std::mutex g_mu;
std::unordered_map<const FeatureTracker*, FeSide> g_side;
FeSide& side_of(const FeatureTracker* t) {
    std::lock_guard<std::mutex> lock(g_mu);
    return g_side[t];
}
void chk(int rc, vg_handle* h, const char* what) {
    if (rc != VG_OK) throw std::runtime_error(std::string(what) + ": " + (h ? vg_last_error(h) : "no handle") + " (no CPU fallback)");
}
// the device side of one tracker: created at the first call, re-configured when the image size or MAX_CNT changes
FeSide& ensure(FeatureTracker* t, int w, int h) {
    FeSide& s = side_of(t);
    if (vg_abi_version() != VG_ABI_VERSION) throw std::runtime_error("libvinsgpu.so was built from another include/vinsgpu.h (ABI version mismatch)");
    if (!s.vg) chk(vg_create(&s.vg), s.vg, "vg_create");
    // vg_fe_read_image takes streams of up to 2048 points; a list never exceeds MAX_CNT (:144-156), the factor is slack.  MAX_CNT
    // beyond 2048 takes the step-by-step branch of readImage.
    const int cap = MAX_CNT <= 2048 ? std::min(std::max(MAX_CNT, 1) * 4, 2048) : MAX_CNT;
    if (s.capacity < cap || s.width != w || s.height != h) {
        chk(vg_fe_configure(s.vg, w, h, 1, cap), s.vg, "vg_fe_configure");
        s.capacity = cap; s.width = w; s.height = h;
    }
    if (s.camera != t->m_camera.get()) {
        s.camera = t->m_camera.get();
        camodocal::PinholeCameraPtr pin = boost::dynamic_pointer_cast<camodocal::PinholeCamera>(t->m_camera);
        s.pinhole = static_cast<bool>(pin);
        if (pin) {
            const camodocal::PinholeCamera::Parameters& p = pin->getParameters();
            const double v[8] = {p.fx(), p.fy(), p.cx(), p.cy(), p.k1(), p.k2(), p.p1(), p.p2()};
            std::memcpy(s.intr, v, sizeof(v));
        }
    }
    return s;
}

}  // namespace

void FeatureTracker::setMask() {                                      // feature_tracker.cpp:36-69
    FeSide& s = ensure(this, COL, ROW);
    // prefer to keep features that are tracked for long time (:43-51): the reference's own sort, verbatim semantics
    vector<pair<int, pair<cv::Point2f, int>>> cnt_pts_id;
    for (unsigned int i = 0; i < forw_pts.size(); i++) cnt_pts_id.push_back(make_pair(track_cnt[i], make_pair(forw_pts[i], ids[i])));
    sort(cnt_pts_id.begin(), cnt_pts_id.end(),
         [](const pair<int, pair<cv::Point2f, int>>& a, const pair<int, pair<cv::Point2f, int>>& b) { return a.first > b.first; });
    const int n = (int)cnt_pts_id.size();
    if (n > s.capacity) throw std::runtime_error("FeatureTracker::setMask: more points than the configured capacity");
    std::vector<float> xy((size_t)s.capacity * 2, 0.f);
    std::vector<int> cnt((size_t)s.capacity, 0), kept((size_t)s.capacity, 0);
    for (int i = 0; i < n; ++i) { xy[2 * i] = cnt_pts_id[i].second.first.x; xy[2 * i + 1] = cnt_pts_id[i].second.first.y; cnt[i] = cnt_pts_id[i].first; }
    // :38-41: the fisheye mask or all-255 is the canvas; the walk in sorted order (the device's own sort by count is stable, i.e. the
    // identity on an already sorted list) keeps a point iff the canvas is still 255 under it and blanks a disc of MIN_DIST around it
    std::vector<uint8_t> base;
    const uint8_t* basep[1] = {nullptr};
    if (FISHEYE) {
        if (fisheye_mask.rows != ROW || fisheye_mask.cols != COL) throw std::runtime_error("FeatureTracker::setMask: fisheye_mask is not ROW x COL");
        base.resize((size_t)ROW * COL);
        for (int y = 0; y < ROW; ++y) std::memcpy(base.data() + (size_t)y * COL, fisheye_mask.data + (size_t)y * fisheye_mask.step, (size_t)COL);
        basep[0] = base.data();
    }
    int nk = 0;
    chk(vg_fe_set_mask(s.vg, xy.data(), cnt.data(), &n, basep, MIN_DIST, kept.data(), &nk), s.vg, "vg_fe_set_mask");
    forw_pts.clear();
    ids.clear();
    track_cnt.clear();
    for (int k = 0; k < nk; ++k) {
        const auto& it = cnt_pts_id[kept[k]];
        forw_pts.push_back(it.second.first);
        ids.push_back(it.second.second);
        track_cnt.push_back(it.first);
    }
}

namespace {
void undistorted_points(FeatureTracker* t, FeSide& s, const float* lifted);

// What readImage does with the statuses of a frame (:115-128, :193-198): forw_pts as the tracking left them, reduceVector by
// `status && inBorder`, track_cnt++, reduceVector by findFundamentalMat's mask -- the reference's own reduceVector calls on the class's
// vectors, fed with what the device computed.
void apply_statuses(FeatureTracker* t, const vg_fe_frame_out* a, int n_in) {
    t->forw_pts.resize(n_in);
    for (int i = 0; i < n_in; i++) t->forw_pts[i] = cv::Point2f(a->forw_xy[2 * i], a->forw_xy[2 * i + 1]);
    if (n_in > 0) {
        vector<uchar> status(a->status_lk, a->status_lk + n_in);
        reduceVector(t->prev_pts, status);
        reduceVector(t->cur_pts, status);
        reduceVector(t->forw_pts, status);
        reduceVector(t->ids, status);
        reduceVector(t->cur_un_pts, status);
        reduceVector(t->track_cnt, status);
    }
    for (auto& n : t->track_cnt) n++;
    if (a->ransac_ran) {
        vector<uchar> status(a->status_f, a->status_f + a->n1);
        reduceVector(t->prev_pts, status);
        reduceVector(t->cur_pts, status);
        reduceVector(t->forw_pts, status);
        reduceVector(t->cur_un_pts, status);
        reduceVector(t->ids, status);
        reduceVector(t->track_cnt, status);
    }
}

struct OrderCtx {
    FeatureTracker* t;
    int n_in;
    vector<pair<int, pair<cv::Point2f, int>>> sorted;      // setMask's cnt_pts_id (:43), the third field holding the list index instead of the id
};
// the order of setMask's walk = the reference's sort call (:47-51): same element type, same comparator, same std::sort -- the permutation
// of a sort depends on the outcomes of its comparisons only, and those look at the counts
int order_callback(void* user, const vg_fe_frame_out* after, int* order) {
    OrderCtx* c = static_cast<OrderCtx*>(user);
    const FeatureTracker* t = c->t;
    // The survivors with their counts (:115-128, :193-198: track_cnt++ after the tracking), formed ON THE SIDE: the class's vectors are
    // only touched once the library call has returned VG_OK, so a frame that fails later (detection overflow, a HIP error) leaves the
    // tracker as it was.
    c->sorted.clear();
    for (int i = 0, k = 0, idx = 0; i < c->n_in; i++) {
        if (!after->status_lk[i]) continue;
        const bool keep = !after->ransac_ran || after->status_f[k];
        k++;
        if (keep) c->sorted.push_back(make_pair(t->track_cnt[i] + 1, make_pair(cv::Point2f(after->forw_xy[2 * i], after->forw_xy[2 * i + 1]), idx++)));
    }
    if ((int)c->sorted.size() != after->n2) return 1;
    sort(c->sorted.begin(), c->sorted.end(),
         [](const pair<int, pair<cv::Point2f, int>>& a, const pair<int, pair<cv::Point2f, int>>& b) { return a.first > b.first; });
    for (int q = 0; q < after->n2; q++) order[q] = c->sorted[q].second.second;
    return 0;
}
}  // namespace

void FeatureTracker::readImage(const cv::Mat& _img, double _cur_time) {     // feature_tracker.cpp:81-167
    cur_time = _cur_time;
    // the reference assumes the configured size (COL / ROW: setMask, inBorder, the fisheye mask); a frame of another size would make
    // this member and the other three re-configure the device side against each other every frame -- tracking would degrade silently
    if (_img.cols != COL || _img.rows != ROW)
        throw std::runtime_error("FeatureTracker::readImage: frame is " + std::to_string(_img.cols) + "x" + std::to_string(_img.rows) +
                                 ", the configuration says " + std::to_string(COL) + "x" + std::to_string(ROW));
    FeSide& s = ensure(this, COL, ROW);
    const float* lifted = nullptr;
    if (s.pinhole && s.capacity <= 2048) {
        // ---- one call per frame
        vg_fe_frame_in in;
        std::memset(&in, 0, sizeof(in));
        in.struct_size = (int)sizeof(in);
        in.img = _img.data; in.stride = (int)_img.step; in.equalize = EQUALIZE; in.publish = PUB_THIS_FRAME ? 1 : 0;
        in.cur_xy = cur_pts.empty() ? nullptr : &cur_pts[0].x; in.n = (int)cur_pts.size();
        in.max_cnt = MAX_CNT; in.min_dist = MIN_DIST; in.quality = 0.01; in.f_threshold = F_THRESHOLD; in.focal_length = FOCAL_LENGTH;
        std::memcpy(in.intr, s.intr, sizeof(in.intr));
        if (FISHEYE && PUB_THIS_FRAME) {                                     // :38-41: the fisheye mask is setMask's canvas
            if (fisheye_mask.rows != ROW || fisheye_mask.cols != COL) throw std::runtime_error("FeatureTracker::readImage: fisheye_mask is not ROW x COL");
            if ((int)fisheye_mask.step == COL)
                in.base_mask = fisheye_mask.data;
            else {
                s.base_copy.resize((size_t)ROW * COL);
                for (int y = 0; y < ROW; ++y) std::memcpy(s.base_copy.data() + (size_t)y * COL, fisheye_mask.data + (size_t)y * fisheye_mask.step, (size_t)COL);
                in.base_mask = s.base_copy.data();
            }
        }
        OrderCtx ctx{this, in.n, {}};
        in.order = order_callback; in.user = &ctx;
        if (in.n > s.capacity) throw std::runtime_error("FeatureTracker::readImage: more points than the configured capacity");
        vg_fe_frame_out out;
        chk(vg_fe_read_image(s.vg, &in, &out), s.vg, "vg_fe_read_image");
        s.stats[0]++; s.stats[1] += in.publish; s.stats[2] += out.ransac_ran; s.stats[3] += (out.fallback & 1) ? 1 : 0;
        s.stats[4] += (out.fallback & 2) ? 1 : 0; s.stats[5] += out.ransac_niters;
        if (forw_img.empty())
            prev_img = cur_img = forw_img = _img;
        else
            forw_img = _img;
        apply_statuses(this, &out, in.n);                                    // only now: the call returned VG_OK
        if (PUB_THIS_FRAME) {
            // setMask's outcome (:53-68): the kept points in the order of the walk
            vector<cv::Point2f> kept_pts;
            vector<int> kept_ids, kept_cnt;
            for (int k = 0; k < out.n_kept; k++) {
                const auto& it = ctx.sorted[out.kept[k]];
                kept_pts.push_back(it.second.first);
                kept_ids.push_back(ids[it.second.second]);
                kept_cnt.push_back(it.first);
            }
            forw_pts = kept_pts; ids = kept_ids; track_cnt = kept_cnt;
            n_pts.clear();                                                   // :144-156
            for (int k = 0; k < out.n_new; k++) n_pts.push_back(cv::Point2f(out.new_xy[2 * k], out.new_xy[2 * k + 1]));
            addPoints();                                                     // the reference's (:71-79)
        }
        if ((int)forw_pts.size() != out.n_final) throw std::runtime_error("FeatureTracker::readImage: list length differs from the device's");
        lifted = out.un_xy;
    } else {
        // ---- step by step (camera models whose lifting runs through camodocal on the host)
        s.stats[6]++;
        // EQUALIZE (:85-95) and `forw_img = img` (:97-104): the frame goes to the device, where the (optional) CLAHE and the pyramid
        // that calcOpticalFlowPyrLK would build of it are formed; the previous frame's pyramid stays where it is
        const uint8_t* planes[1] = {_img.data};
        chk(vg_fe_push_frames(s.vg, planes, (int)_img.step, EQUALIZE), s.vg, "vg_fe_push_frames");
        if (forw_img.empty())
            prev_img = cur_img = forw_img = _img;
        else
            forw_img = _img;

        forw_pts.clear();

        if (cur_pts.size() > 0) {                                            // :108-125
            vector<uchar> status(cur_pts.size());
            vector<float> err(cur_pts.size());
            forw_pts.resize(cur_pts.size());
            chk(vg_fe_track(s.vg, 0, &cur_pts[0].x, (int)cur_pts.size(), &forw_pts[0].x, status.data(), err.data()), s.vg, "vg_fe_track");
            for (int i = 0; i < int(forw_pts.size()); i++)
                if (status[i] && !inBorder(forw_pts[i])) status[i] = 0;
            reduceVector(prev_pts, status);
            reduceVector(cur_pts, status);
            reduceVector(forw_pts, status);
            reduceVector(ids, status);
            reduceVector(cur_un_pts, status);
            reduceVector(track_cnt, status);
        }

        for (auto& n : track_cnt) n++;                                       // :127-128

        if (PUB_THIS_FRAME) {                                                // :130-158
            rejectWithF();
            setMask();
            int n_max_cnt = MAX_CNT - static_cast<int>(forw_pts.size());
            if (n_max_cnt > 0) {
                n_pts.resize(n_max_cnt);
                int n = 0;
                chk(vg_fe_detect_masked(s.vg, 0, n_max_cnt, 0.01, (double)MIN_DIST, &n_pts[0].x, &n), s.vg, "vg_fe_detect_masked");
                n_pts.resize(n);
            } else
                n_pts.clear();
            addPoints();                                                     // the reference's (:71-79)
        }
    }
    prev_img = cur_img;                                                      // :160-166
    prev_pts = cur_pts;
    prev_un_pts = cur_un_pts;
    cur_img = forw_img;
    cur_pts = forw_pts;
    undistorted_points(this, s, lifted);
    prev_time = cur_time;
}

void FeatureTracker::rejectWithF() {                                        // feature_tracker.cpp:169-202
    if (forw_pts.size() >= 8) {
        FeSide& s = ensure(this, COL, ROW);
        // :175-188 as the reference has it: 2 x <= 150 points through camodocal on the host, in double (the float pair handed to the
        // RANSAC is rounded from FOCAL_LENGTH * x / z + COL / 2)
        vector<cv::Point2f> un_cur_pts(cur_pts.size()), un_forw_pts(forw_pts.size());
        for (unsigned int i = 0; i < cur_pts.size(); i++) {
            Eigen::Vector3d tmp_p;
            m_camera->liftProjective(Eigen::Vector2d(cur_pts[i].x, cur_pts[i].y), tmp_p);
            tmp_p.x() = FOCAL_LENGTH * tmp_p.x() / tmp_p.z() + COL / 2.0;
            tmp_p.y() = FOCAL_LENGTH * tmp_p.y() / tmp_p.z() + ROW / 2.0;
            un_cur_pts[i] = cv::Point2f(tmp_p.x(), tmp_p.y());

            m_camera->liftProjective(Eigen::Vector2d(forw_pts[i].x, forw_pts[i].y), tmp_p);
            tmp_p.x() = FOCAL_LENGTH * tmp_p.x() / tmp_p.z() + COL / 2.0;
            tmp_p.y() = FOCAL_LENGTH * tmp_p.y() / tmp_p.z() + ROW / 2.0;
            un_forw_pts[i] = cv::Point2f(tmp_p.x(), tmp_p.y());
        }
        const int n = (int)forw_pts.size();
        vector<uchar> status(n);
        chk(vg_fe_reject_with_f(s.vg, &un_cur_pts[0].x, &un_forw_pts[0].x, n, F_THRESHOLD, status.data(), nullptr, nullptr), s.vg, "vg_fe_reject_with_f");
        reduceVector(prev_pts, status);
        reduceVector(cur_pts, status);
        reduceVector(forw_pts, status);
        reduceVector(cur_un_pts, status);
        reduceVector(ids, status);
        reduceVector(track_cnt, status);
    }
}

namespace {
// undistortedPoints() (feature_tracker.cpp:258-306); `lifted` = the (x / z, y / z) pairs of cur_pts when the device already formed them
void undistorted_points(FeatureTracker* t, FeSide& s, const float* lifted) {
    vector<cv::Point2f>& cur_pts = t->cur_pts;
    vector<cv::Point2f>& cur_un_pts = t->cur_un_pts;
    vector<cv::Point2f>& pts_velocity = t->pts_velocity;
    vector<int>& ids = t->ids;
    map<int, cv::Point2f>& cur_un_pts_map = t->cur_un_pts_map;
    map<int, cv::Point2f>& prev_un_pts_map = t->prev_un_pts_map;
    const double cur_time = t->cur_time, prev_time = t->prev_time;
    cur_un_pts.clear();
    cur_un_pts_map.clear();
    const int n = (int)cur_pts.size();
    std::vector<float> un_own;
    const float* un = lifted;
    if (!un) {
        un_own.resize((size_t)std::max(n, 1) * 2);
        if (n > 0 && s.pinhole)
            chk(vg_fe_undistort(s.vg, &cur_pts[0].x, n, s.intr, un_own.data()), s.vg, "vg_fe_undistort");
        else
            for (int i = 0; i < n; i++) {                                    // MEI / KANNALA_BRANDT / SCARAMUZZA: camodocal on the host (:262-266)
                Eigen::Vector2d a(cur_pts[i].x, cur_pts[i].y);
                Eigen::Vector3d b;
                t->m_camera->liftProjective(a, b);
                un_own[2 * i] = (float)(b.x() / b.z());
                un_own[2 * i + 1] = (float)(b.y() / b.z());
            }
        un = un_own.data();
    }
    for (int i = 0; i < n; i++) {
        cur_un_pts.push_back(cv::Point2f(un[2 * i], un[2 * i + 1]));
        cur_un_pts_map.insert(make_pair(ids[i], cv::Point2f(un[2 * i], un[2 * i + 1])));
    }
    // caculate points velocity (:270-304)
    if (!prev_un_pts_map.empty()) {
        double dt = cur_time - prev_time;
        pts_velocity.clear();
        for (unsigned int i = 0; i < cur_un_pts.size(); i++) {
            if (ids[i] != -1) {
                std::map<int, cv::Point2f>::iterator it = prev_un_pts_map.find(ids[i]);
                if (it != prev_un_pts_map.end()) {
                    double v_x = (cur_un_pts[i].x - it->second.x) / dt;
                    double v_y = (cur_un_pts[i].y - it->second.y) / dt;
                    pts_velocity.push_back(cv::Point2f(v_x, v_y));
                } else
                    pts_velocity.push_back(cv::Point2f(0, 0));
            } else {
                pts_velocity.push_back(cv::Point2f(0, 0));
            }
        }
    } else {
        for (unsigned int i = 0; i < cur_pts.size(); i++) pts_velocity.push_back(cv::Point2f(0, 0));
    }
    prev_un_pts_map = cur_un_pts_map;
}
}  // namespace

void FeatureTracker::undistortedPoints() { undistorted_points(this, ensure(this, COL, ROW), nullptr); }

// ---- C entry points for the surrounding code (the reference class has no member to hang these on)
extern "C" {
// release the device handle of a FeatureTracker that is about to be destroyed or re-used for another stream
void vins_fe_gpu_release(FeatureTracker* t) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_side.find(t);
    if (it == g_side.end()) return;
    if (it->second.vg) vg_destroy(it->second.vg);
    g_side.erase(it);
}
// how the frames of this tracker ran: [0] frames through vg_fe_read_image, [1] of them published, [2] with rejectWithF, [3] / [4] whose
// estimate went back to the host (a RANSAC sample OpenCV would have redrawn / the LMedS range), [5] sum of the RANSAC iterations that
// counted, [6] frames through the step-by-step members
void vins_fe_gpu_stats(FeatureTracker* t, int* out8) { std::memcpy(out8, side_of(t).stats, sizeof(int) * 8); }
// the device handle (e.g. for vg_fe_get_mask / vg_fe_get_level: the members `mask` and the equalized `cur_img` stay on the device)
vg_handle* vins_fe_gpu_handle(FeatureTracker* t) { return side_of(t).vg; }
}
