// feature_tracker.h — drop-in FeatureTracker for the MI355X path: same class name, members and method signatures as
// feature_tracker/src/feature_tracker.h:28-65 of the reference.  readImage() is ONE library call per frame, vg_fe_read_image
// (include/vinsgpu.h); the members setMask / rejectWithF / undistortedPoints keep the step-by-step entry points
// of feature_tracker.cpp:81-167 for callers that drive the steps themselves:
//   cv::createCLAHE(...)->apply      -> vg_fe_push_frames(.., equalize)     (:87-93)
//   cv::calcOpticalFlowPyrLK         -> vg_fe_track                          (:113)
//   cv::goodFeaturesToTrack          -> vg_fe_detect                         (:149)
//   cv::findFundamentalMat(RANSAC)   -> vg_fe_reject_with_f                  (:191; deterministic RANSAC, ASSUMPTIONS F9)
// Differences (documented in INTEGRATION.md): the camera model is the EuRoC pinhole restated from
// camera_model/src/camera_models/PinholeCamera.cc:450-510,646-661 instead of camodocal::CameraPtr.
#pragma once
#include <map>
#include <string>
#include <vector>
#include "compat/cv_compat.h"
#include "../../include/vinsgpu.h"

using namespace std;

// globals of feature_tracker/src/parameters.h (same names)
extern int ROW, COL, MAX_CNT, MIN_DIST, EQUALIZE, FISHEYE, FOCAL_LENGTH;
extern bool PUB_THIS_FRAME;
extern double F_THRESHOLD;
extern int FREQ, SHOW_TRACK, FE_WINDOW_SIZE;
extern std::string IMAGE_TOPIC, FE_IMU_TOPIC, FISHEYE_MASK;
// feature_tracker/src/parameters.cpp:37-74 without the ROS node handle: the path of the configuration file is passed directly
void readFeatureTrackerParameters(const std::string& config_file, const std::string& vins_folder = "");

bool inBorder(const cv::Point2f& pt);
void reduceVector(vector<cv::Point2f>& v, vector<uchar> status);
void reduceVector(vector<int>& v, vector<uchar> status);

struct PinholeModel {            // camodocal::PinholeCamera parameters (config/euroc/euroc_config.yaml:13-22)
    double fx = 461.6, fy = 460.3, cx = 363.0, cy = 248.1, k1 = -2.917e-01, k2 = 8.228e-02, p1 = 5.333e-05, p2 = -1.578e-04;
    void liftProjective(double u, double v, double& x, double& y) const;
};

class FeatureTracker {
  public:
    FeatureTracker();
    ~FeatureTracker();

    void readImage(const cv::Mat& _img, double _cur_time);
    void setMask();
    void addPoints();
    bool updateID(unsigned int i);
    void readIntrinsicParameter(const string& calib_file);   // PINHOLE section of the configuration file (host/yaml_config.h)
    void rejectWithF();
    void undistortedPoints();
    // the two halves of readImage around the library call (public for the order callback; not part of the reference's interface)
    void applyStatuses(const vg_fe_frame_out& after, int n_in);
    void liftedPoints(const float* un_xy);

    cv::Mat mask;
    cv::Mat fisheye_mask;
    cv::Mat prev_img, cur_img, forw_img;
    vector<cv::Point2f> n_pts;
    vector<cv::Point2f> prev_pts, cur_pts, forw_pts;
    vector<cv::Point2f> prev_un_pts, cur_un_pts;
    vector<cv::Point2f> pts_velocity;
    vector<int> ids;
    vector<int> track_cnt;
    map<int, cv::Point2f> cur_un_pts_map;
    map<int, cv::Point2f> prev_un_pts_map;
    PinholeModel m_camera;
    double cur_time;
    double prev_time;

    static int n_id;

  private:
    vg_handle* vg_ = nullptr;      // owns the device-side pyramids of cur_img / forw_img
    int fe_capacity_ = 0;
    bool configured_ = false;
};
