// feature_tracker.cpp — see feature_tracker.h.  Control flow follows feature_tracker/src/feature_tracker.cpp of the
// reference (cited per block); the arithmetic runs on the GPU through libvinsgpu.so.
#include "feature_tracker.h"
#include <cstring>
#include "yaml_config.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <stdexcept>

int ROW = 480, COL = 752, MAX_CNT = 150, MIN_DIST = 30, EQUALIZE = 1, FISHEYE = 0, FOCAL_LENGTH = 460;
bool PUB_THIS_FRAME = false;
double F_THRESHOLD = 1.0;
int FREQ = 10, SHOW_TRACK = 1, FE_WINDOW_SIZE = 20;
std::string IMAGE_TOPIC, FE_IMU_TOPIC, FISHEYE_MASK;

void readFeatureTrackerParameters(const std::string& config_file, const std::string& vins_folder) {   // feature_tracker/src/parameters.cpp:37-74
    VinsYaml fs;
    if (!fs.load(config_file)) throw std::runtime_error("ERROR: Wrong path to settings: " + config_file);
    IMAGE_TOPIC = fs.str("image_topic");
    FE_IMU_TOPIC = fs.str("imu_topic");
    MAX_CNT = (int)fs.number("max_cnt");
    MIN_DIST = (int)fs.number("min_dist");
    ROW = (int)fs.number("image_height");
    COL = (int)fs.number("image_width");
    FREQ = (int)fs.number("freq");
    F_THRESHOLD = fs.number("F_threshold");
    SHOW_TRACK = (int)fs.number("show_track");
    EQUALIZE = (int)fs.number("equalize");
    FISHEYE = (int)fs.number("fisheye");
    if (FISHEYE == 1) FISHEYE_MASK = vins_folder + "config/fisheye_mask.jpg";
    FE_WINDOW_SIZE = 20;
    FOCAL_LENGTH = 460;
    PUB_THIS_FRAME = false;
    if (FREQ == 0) FREQ = 100;
}
int FeatureTracker::n_id = 0;

static int cvRoundf(float v) { return (int)std::lrintf(v); }

bool inBorder(const cv::Point2f& pt) {                       // feature_tracker.cpp:5-11
    const int BORDER_SIZE = 1;
    int img_x = cvRoundf(pt.x);
    int img_y = cvRoundf(pt.y);
    return BORDER_SIZE <= img_x && img_x < COL - BORDER_SIZE && BORDER_SIZE <= img_y && img_y < ROW - BORDER_SIZE;
}

void reduceVector(vector<cv::Point2f>& v, vector<uchar> status) {   // :13-20
    int j = 0;
    for (int i = 0; i < int(v.size()); i++)
        if (status[i]) v[j++] = v[i];
    v.resize(j);
}
void reduceVector(vector<int>& v, vector<uchar> status) {           // :22-29
    int j = 0;
    for (int i = 0; i < int(v.size()); i++)
        if (status[i]) v[j++] = v[i];
    v.resize(j);
}

void PinholeModel::liftProjective(double u, double v, double& x, double& y) const {
    // PinholeCamera::liftProjective, recursive distortion model with n = 8 (PinholeCamera.cc:450-510, :646-661)
    const double mx_d = (1.0 / fx) * u + (-cx / fx), my_d = (1.0 / fy) * v + (-cy / fy);
    auto distortion = [&](double px, double py, double& dx, double& dy) {
        const double mx2 = px * px, my2 = py * py, mxy = px * py, rho2 = mx2 + my2, rad = k1 * rho2 + k2 * rho2 * rho2;
        dx = px * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2);
        dy = py * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2);
    };
    double dx, dy;
    distortion(mx_d, my_d, dx, dy);
    double mx_u = mx_d - dx, my_u = my_d - dy;
    for (int i = 1; i < 8; ++i) {
        distortion(mx_u, my_u, dx, dy);
        mx_u = mx_d - dx; my_u = my_d - dy;
    }
    x = mx_u; y = my_u;
}

FeatureTracker::FeatureTracker() : cur_time(0), prev_time(0) {}
FeatureTracker::~FeatureTracker() { if (vg_) vg_destroy(vg_); }

static void chk(int rc, vg_handle* h, const char* what) {
    if (rc != VG_OK) throw std::runtime_error(std::string(what) + ": " + (h ? vg_last_error(h) : "no handle") + " (no CPU fallback)");
}

void FeatureTracker::setMask() {                             // feature_tracker.cpp:36-69, on the device (vg_fe_set_mask)
    // The reference orders the points with std::sort by track_cnt (:48-51): the order among equal counts is whatever the platform's
    // std::sort gives.  The same call is made here, and the device walks the list in that order (its own sort by count is stable,
    // i.e. the identity on a sorted list) -- the result is the reference's on the platform this is built on (tests/test_fe_dropin.py).
    // `mask` itself stays on the device: goodFeaturesToTrack below consumes it there (vg_fe_detect_masked).
    const int n = (int)forw_pts.size();
    if (n > fe_capacity_) throw std::runtime_error("FeatureTracker::setMask: more points than the configured capacity");
    vector<pair<int, pair<cv::Point2f, int>>> cnt_pts_id;
    for (int i = 0; i < n; i++) cnt_pts_id.push_back(make_pair(track_cnt[i], make_pair(forw_pts[i], ids[i])));
    sort(cnt_pts_id.begin(), cnt_pts_id.end(),
         [](const pair<int, pair<cv::Point2f, int>>& a, const pair<int, pair<cv::Point2f, int>>& b) { return a.first > b.first; });
    std::vector<float> xy((size_t)fe_capacity_ * 2, 0.f);
    std::vector<int> cnt((size_t)fe_capacity_, 0), kept((size_t)fe_capacity_, 0);
    for (int i = 0; i < n; ++i) { xy[2 * i] = cnt_pts_id[i].second.first.x; xy[2 * i + 1] = cnt_pts_id[i].second.first.y; cnt[i] = cnt_pts_id[i].first; }
    const uint8_t* base[1] = {FISHEYE ? fisheye_mask.data : nullptr};
    int nk = 0;
    chk(vg_fe_set_mask(vg_, xy.data(), cnt.data(), &n, base, MIN_DIST, kept.data(), &nk), vg_, "vg_fe_set_mask");
    forw_pts.clear(); ids.clear(); track_cnt.clear();
    for (int k = 0; k < nk; ++k) {
        const auto& it = cnt_pts_id[kept[k]];
        forw_pts.push_back(it.second.first); ids.push_back(it.second.second); track_cnt.push_back(it.first);
    }
}

void FeatureTracker::addPoints() {                            // :71-79
    for (auto& p : n_pts) {
        forw_pts.push_back(p);
        ids.push_back(-1);
        track_cnt.push_back(1);
    }
}

// What readImage does with the statuses of a frame (:115-128, :193-198): forw_pts as the tracking left them, reduceVector by
// `status && inBorder`, track_cnt++, reduceVector by findFundamentalMat's mask
void FeatureTracker::applyStatuses(const vg_fe_frame_out& a, int n_in) {
    forw_pts.resize(n_in);
    for (int i = 0; i < n_in; i++) forw_pts[i] = cv::Point2f(a.forw_xy[2 * i], a.forw_xy[2 * i + 1]);
    if (n_in > 0) {
        vector<uchar> status(a.status_lk, a.status_lk + n_in);
        reduceVector(prev_pts, status);
        reduceVector(cur_pts, status);
        reduceVector(forw_pts, status);
        reduceVector(ids, status);
        reduceVector(cur_un_pts, status);
        reduceVector(track_cnt, status);
    }
    for (auto& n : track_cnt) n++;
    if (a.ransac_ran) {
        vector<uchar> status(a.status_f, a.status_f + a.n1);
        reduceVector(prev_pts, status);
        reduceVector(cur_pts, status);
        reduceVector(forw_pts, status);
        reduceVector(cur_un_pts, status);
        reduceVector(ids, status);
        reduceVector(track_cnt, status);
    }
}

namespace {
struct OrderCtx {
    FeatureTracker* t;
    int n_in;
    vector<pair<int, pair<cv::Point2f, int>>> sorted;      // setMask's cnt_pts_id (:43) with the list index in the place of the id
};
// The order of setMask's walk = the reference's sort call (:47-51): std::sort by track_cnt is not stable, the order among equal counts is
// what the platform's std::sort makes of the sequence; its permutation depends on the comparisons only, and those look at the counts.
int order_callback(void* user, const vg_fe_frame_out* after, int* order) {
    OrderCtx* c = static_cast<OrderCtx*>(user);
    // the survivors with their counts (:115-128, :193-198), formed on the side: the tracker's own vectors are only touched once the library
    // call has returned VG_OK, so a frame that fails later leaves the tracker as it was
    c->sorted.clear();
    for (int i = 0, k = 0, idx = 0; i < c->n_in; i++) {
        if (!after->status_lk[i]) continue;
        const bool keep = !after->ransac_ran || after->status_f[k];
        k++;
        if (keep) c->sorted.push_back(make_pair(c->t->track_cnt[i] + 1, make_pair(cv::Point2f(after->forw_xy[2 * i], after->forw_xy[2 * i + 1]), idx++)));
    }
    if ((int)c->sorted.size() != after->n2) return 1;
    sort(c->sorted.begin(), c->sorted.end(),
         [](const pair<int, pair<cv::Point2f, int>>& a, const pair<int, pair<cv::Point2f, int>>& b) { return a.first > b.first; });
    for (int q = 0; q < after->n2; q++) order[q] = c->sorted[q].second.second;
    return 0;
}
}  // namespace

// ONE library call per frame (vg_fe_read_image): the frame and cur_pts go up in one block; CLAHE, pyramid, LK, the border test,
// reduceVector and -- on a published frame -- rejectWithF, setMask's walk, goodFeaturesToTrack, addPoints and the lifting of
// undistortedPoints run on the device without the host in between; the members setMask() / rejectWithF() / undistortedPoints() remain
// for callers that drive the steps themselves.
void FeatureTracker::readImage(const cv::Mat& _img, double _cur_time) {    // :81-167
    cur_time = _cur_time;
    if (!configured_) {
        chk(vg_create(&vg_), vg_, "vg_create");
        // vg_fe_read_image takes streams of up to 2048 points; a list never exceeds MAX_CNT (:144-156), the factor is slack
        if (MAX_CNT > 2048) throw std::runtime_error("FeatureTracker::readImage: max_cnt > 2048 is not offered (vg_fe_read_image)");
        fe_capacity_ = std::min(std::max(MAX_CNT, 1) * 4, 2048);
        chk(vg_fe_configure(vg_, COL, ROW, 1, fe_capacity_), vg_, "vg_fe_configure");
        configured_ = true;
    }
    if ((int)cur_pts.size() > fe_capacity_) throw std::runtime_error("FeatureTracker::readImage: more points than the configured capacity");
    vg_fe_frame_in in;
    std::memset(&in, 0, sizeof(in));
    in.struct_size = (int)sizeof(in);
    in.img = _img.data; in.stride = (int)_img.step; in.equalize = EQUALIZE; in.publish = PUB_THIS_FRAME ? 1 : 0;
    in.cur_xy = cur_pts.empty() ? nullptr : &cur_pts[0].x; in.n = (int)cur_pts.size();
    in.max_cnt = MAX_CNT; in.min_dist = MIN_DIST; in.quality = 0.01; in.f_threshold = F_THRESHOLD; in.focal_length = FOCAL_LENGTH;
    const double intr[8] = {m_camera.fx, m_camera.fy, m_camera.cx, m_camera.cy, m_camera.k1, m_camera.k2, m_camera.p1, m_camera.p2};
    std::memcpy(in.intr, intr, sizeof(intr));
    in.base_mask = (FISHEYE && PUB_THIS_FRAME) ? fisheye_mask.data : nullptr;           // :38-41 (contiguous ROW x COL, readFeatureTrackerParameters)
    OrderCtx ctx{this, in.n, {}};
    in.order = order_callback; in.user = &ctx;
    vg_fe_frame_out out;
    chk(vg_fe_read_image(vg_, &in, &out), vg_, "vg_fe_read_image");
    if (forw_img.empty()) prev_img = cur_img = forw_img = _img;
    else forw_img = _img;
    applyStatuses(out, in.n);                                                // only now: the call returned VG_OK
    if (PUB_THIS_FRAME) {
        vector<cv::Point2f> kept_pts;                                        // setMask's outcome (:53-68): the kept points in walk order
        vector<int> kept_ids, kept_cnt;
        for (int k = 0; k < out.n_kept; k++) {
            const auto& it = ctx.sorted[out.kept[k]];
            kept_pts.push_back(it.second.first);
            kept_ids.push_back(ids[it.second.second]);
            kept_cnt.push_back(it.first);
        }
        forw_pts = kept_pts; ids = kept_ids; track_cnt = kept_cnt;
        n_pts.clear();                                                       // :144-156
        for (int k = 0; k < out.n_new; k++) n_pts.push_back(cv::Point2f(out.new_xy[2 * k], out.new_xy[2 * k + 1]));
        addPoints();
    }
    if ((int)forw_pts.size() != out.n_final) throw std::runtime_error("FeatureTracker::readImage: list length differs from the device's");
    prev_img = cur_img;                                                      // :160-166
    prev_pts = cur_pts;
    prev_un_pts = cur_un_pts;
    cur_img = forw_img;
    cur_pts = forw_pts;
    liftedPoints(out.un_xy);
    prev_time = cur_time;
}

void FeatureTracker::rejectWithF() {                                       // feature_tracker.cpp:169-202
    if (forw_pts.size() >= 8) {
        const int n = (int)forw_pts.size();
        vector<cv::Point2f> un_cur_pts(cur_pts.size()), un_forw_pts(forw_pts.size());
        for (unsigned int i = 0; i < cur_pts.size(); i++) {
            double x, y;
            m_camera.liftProjective(cur_pts[i].x, cur_pts[i].y, x, y);
            x = FOCAL_LENGTH * x + COL / 2.0;
            y = FOCAL_LENGTH * y + ROW / 2.0;
            un_cur_pts[i] = cv::Point2f((float)x, (float)y);
            m_camera.liftProjective(forw_pts[i].x, forw_pts[i].y, x, y);
            x = FOCAL_LENGTH * x + COL / 2.0;
            y = FOCAL_LENGTH * y + ROW / 2.0;
            un_forw_pts[i] = cv::Point2f((float)x, (float)y);
        }
        vector<uchar> status(n);
        // cv::findFundamentalMat(un_cur_pts, un_forw_pts, cv::FM_RANSAC, F_THRESHOLD, 0.99, status) on the device
        chk(vg_fe_reject_with_f(vg_, &un_cur_pts[0].x, &un_forw_pts[0].x, n, F_THRESHOLD, status.data(), nullptr, nullptr), vg_, "vg_fe_reject_with_f");
        reduceVector(prev_pts, status);
        reduceVector(cur_pts, status);
        reduceVector(forw_pts, status);
        reduceVector(cur_un_pts, status);
        reduceVector(ids, status);
        reduceVector(track_cnt, status);
    }
}

bool FeatureTracker::updateID(unsigned int i) {                              // :204-214
    if (i < ids.size()) {
        if (ids[i] == -1) ids[i] = n_id++;
        return true;
    }
    return false;
}

void FeatureTracker::readIntrinsicParameter(const string& calib_file) {
    // feature_tracker.cpp:216-220 -> CameraFactory::generateCameraFromYamlFile -> PinholeCamera::Parameters::readFromYamlFile
    // (camera_model/src/camera_models/PinholeCamera.cc): model_type PINHOLE, distortion_parameters k1 k2 p1 p2, projection_parameters
    // fx fy cx cy.  Other camera models (MEI, KANNALA_BRANDT) are outside this front end (camodocal is out of scope).
    VinsYaml fs;
    if (!fs.load(calib_file)) throw std::runtime_error("readIntrinsicParameter: cannot read " + calib_file);
    const std::string model = fs.str("model_type", "PINHOLE");
    if (model != "PINHOLE") throw std::runtime_error("readIntrinsicParameter: camera model " + model + " is not supported (PINHOLE only)");
    m_camera.k1 = fs.number("distortion_parameters.k1"); m_camera.k2 = fs.number("distortion_parameters.k2");
    m_camera.p1 = fs.number("distortion_parameters.p1"); m_camera.p2 = fs.number("distortion_parameters.p2");
    m_camera.fx = fs.number("projection_parameters.fx"); m_camera.fy = fs.number("projection_parameters.fy");
    m_camera.cx = fs.number("projection_parameters.cx"); m_camera.cy = fs.number("projection_parameters.cy");
}

void FeatureTracker::undistortedPoints() {                                   // :258-306, lifting on the device
    const int n = (int)cur_pts.size();
    std::vector<float> un((size_t)std::max(n, 1) * 2);
    const double intr[8] = {m_camera.fx, m_camera.fy, m_camera.cx, m_camera.cy, m_camera.k1, m_camera.k2, m_camera.p1, m_camera.p2};
    if (n > 0) chk(vg_fe_undistort(vg_, &cur_pts[0].x, n, intr, un.data()), vg_, "vg_fe_undistort");
    liftedPoints(un.data());
}

// :262-304 given the lifted (x / z, y / z) pairs of cur_pts
void FeatureTracker::liftedPoints(const float* un) {
    cur_un_pts.clear();
    cur_un_pts_map.clear();
    const int n = (int)cur_pts.size();
    for (int i = 0; i < n; i++) {
        cur_un_pts.push_back(cv::Point2f(un[2 * i], un[2 * i + 1]));
        cur_un_pts_map.insert(make_pair(ids[i], cv::Point2f(un[2 * i], un[2 * i + 1])));
    }
    if (!prev_un_pts_map.empty()) {
        double dt = cur_time - prev_time;
        pts_velocity.clear();
        for (unsigned int i = 0; i < cur_un_pts.size(); i++) {
            if (ids[i] != -1) {
                auto it = prev_un_pts_map.find(ids[i]);
                if (it != prev_un_pts_map.end()) {
                    double v_x = (cur_un_pts[i].x - it->second.x) / dt;
                    double v_y = (cur_un_pts[i].y - it->second.y) / dt;
                    pts_velocity.push_back(cv::Point2f((float)v_x, (float)v_y));
                } else
                    pts_velocity.push_back(cv::Point2f(0, 0));
            } else
                pts_velocity.push_back(cv::Point2f(0, 0));
        }
    } else {
        for (unsigned int i = 0; i < cur_pts.size(); i++) pts_velocity.push_back(cv::Point2f(0, 0));
    }
    prev_un_pts_map = cur_un_pts_map;
}
