// feature_tracker.cpp — see feature_tracker.h.  Control flow follows feature_tracker/src/feature_tracker.cpp of the
// reference (cited per block); the arithmetic runs on the GPU through libvinsgpu.so.
#include "feature_tracker.h"
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

void FeatureTracker::readImage(const cv::Mat& _img, double _cur_time) {    // :81-167
    cur_time = _cur_time;
    if (!configured_) {
        chk(vg_create(&vg_), vg_, "vg_create");
        fe_capacity_ = std::max(MAX_CNT, 1) * 4;
        chk(vg_fe_configure(vg_, COL, ROW, 1, fe_capacity_), vg_, "vg_fe_configure");
        configured_ = true;
    }
    // EQUALIZE (CLAHE, :87-93) happens on the device; `forw_img = img` (:97-104) = the device pyramid rotation
    const uint8_t* planes[1] = {_img.data};
    chk(vg_fe_push_frames(vg_, planes, (int)_img.step, EQUALIZE), vg_, "vg_fe_push_frames");
    if (forw_img.empty()) prev_img = cur_img = forw_img = _img;
    else forw_img = _img;

    forw_pts.clear();
    if (cur_pts.size() > 0) {                                                // :108-125
        vector<uchar> status(cur_pts.size());
        vector<float> err(cur_pts.size());
        forw_pts.resize(cur_pts.size());
        chk(vg_fe_track(vg_, 0, &cur_pts[0].x, (int)cur_pts.size(), &forw_pts[0].x, status.data(), err.data()), vg_, "vg_fe_track");
        for (int i = 0; i < int(forw_pts.size()); i++)
            if (status[i] && !inBorder(forw_pts[i])) status[i] = 0;
        reduceVector(prev_pts, status);
        reduceVector(cur_pts, status);
        reduceVector(forw_pts, status);
        reduceVector(ids, status);
        reduceVector(cur_un_pts, status);
        reduceVector(track_cnt, status);
    }
    for (auto& n : track_cnt) n++;                                           // :127-128

    if (PUB_THIS_FRAME) {                                                    // :130-158
        rejectWithF();
        setMask();
        int n_max_cnt = MAX_CNT - static_cast<int>(forw_pts.size());
        if (n_max_cnt > 0) {
            n_pts.resize(n_max_cnt);
            int n = 0;
            chk(vg_fe_detect_masked(vg_, 0, n_max_cnt, 0.01, (double)MIN_DIST, &n_pts[0].x, &n), vg_, "vg_fe_detect_masked");
            n_pts.resize(n);
        } else
            n_pts.clear();
        addPoints();
    }
    prev_img = cur_img;                                                      // :160-166
    prev_pts = cur_pts;
    prev_un_pts = cur_un_pts;
    cur_img = forw_img;
    cur_pts = forw_pts;
    undistortedPoints();
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
    cur_un_pts.clear();
    cur_un_pts_map.clear();
    const int n = (int)cur_pts.size();
    std::vector<float> un((size_t)std::max(n, 1) * 2);
    const double intr[8] = {m_camera.fx, m_camera.fy, m_camera.cx, m_camera.cy, m_camera.k1, m_camera.k2, m_camera.p1, m_camera.p2};
    if (n > 0) chk(vg_fe_undistort(vg_, &cur_pts[0].x, n, intr, un.data()), vg_, "vg_fe_undistort");
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
