// TEST INFRASTRUCTURE — C entry points around the reference's OWN front-end node (feature_tracker/src/feature_tracker_node.cpp,
// feature_tracker.cpp, parameters.cpp and camera_model/src/camera_models/*.cc compiled unchanged where they lie, oracle/Makefile
// target ref_fe).  Nothing of the reference is restated here: vfe_start() runs the node's main() (renamed on the command line;
// ros::spin() of the stand-in returns at once), vfe_image() hands a mono8 sensor_msgs::Image to the node's img_callback() —
// first-image handling, stream-discontinuity reset, the PUB_THIS_FRAME frequency gate (:29-62), readImage(), updateID() (:103-111)
// and the assembly of the `feature` point cloud (:113-163) all run as the reference wrote them.  What the stand-in
// ros::Publisher captured is read back through vfe_published_*.
//
// Two builds link this file:  oracle/_ref/libvins_ref_fe.so      the reference's FeatureTracker (its cv:: calls forward to
//                                                                 oracle/fe_cpu.cpp, see cv_standin.h)
//                             oracle/_ref/libvins_ref_fe_gpu.so   the same objects with readImage / setMask / rejectWithF /
//                                                                 undistortedPoints replaced by the product's drop-in
//                                                                 (vins-mono_amd/host/dropin/feature_tracker_readimage.cpp)
#include <cstring>
#include <string>
#include <vector>

#include <ros/ros.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/PointCloud.h>
#include <std_msgs/Bool.h>
#include "feature_tracker.h"

// globals of feature_tracker_node.cpp (:14-27)
extern FeatureTracker trackerData[NUM_OF_CAM];
extern double first_image_time;
extern int pub_count;
extern bool first_image_flag;
extern double last_image_time;
extern bool init_pub;
void img_callback(const sensor_msgs::ImageConstPtr& img_msg);
int vins_ref_fe_node_main(int argc, char** argv);
extern "C" void vins_fe_gpu_release(FeatureTracker*) __attribute__((weak));      // defined by the drop-in build only
extern "C" void vins_fe_gpu_stats(FeatureTracker*, int*) __attribute__((weak));

namespace {
std::vector<const sensor_msgs::PointCloud*> clouds() {
    std::vector<const sensor_msgs::PointCloud*> out;
    for (const auto& m : ros::captured())
        if (m.topic == "feature") out.push_back(static_cast<const sensor_msgs::PointCloud*>(m.msg.get()));
    return out;
}
}  // namespace

extern "C" {
int vfe_abi_version() { return 1; }
int vfe_has_gpu_readimage() { return vins_fe_gpu_release != nullptr ? 1 : 0; }
// how the drop-in ran the frames of tracker 0 so far (vins_fe_gpu_stats of feature_tracker_readimage.cpp); zeros in the all-reference build
void vfe_gpu_stats(int* out8) {
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (vins_fe_gpu_stats) vins_fe_gpu_stats(&trackerData[0], out8);
}

// the node from a fresh state: its globals as their initialisers leave them, then its main()
int vfe_start(const char* config_file, const char* vins_folder) {
    for (int i = 0; i < NUM_OF_CAM; ++i) {
        if (vins_fe_gpu_release) vins_fe_gpu_release(&trackerData[i]);
        trackerData[i] = FeatureTracker();
    }
    FeatureTracker::n_id = 0;
    first_image_time = 0; pub_count = 1; first_image_flag = true; last_image_time = 0; init_pub = 0;
    CAM_NAMES.clear();
    ros::captured().clear();
    ros::NodeHandle::params()["config_file"] = config_file;
    ros::NodeHandle::params()["vins_folder"] = vins_folder ? vins_folder : "";
    char arg0[] = "feature_tracker";
    char* argv[] = {arg0, nullptr};
    return vins_ref_fe_node_main(1, argv);
}
// overrides after readParameters() (tests vary them without writing configuration files)
void vfe_set_option(const char* name, double v) {
    const std::string n(name);
    if (n == "max_cnt") MAX_CNT = (int)v;
    else if (n == "min_dist") MIN_DIST = (int)v;
    else if (n == "freq") FREQ = (int)v;
    else if (n == "F_threshold") F_THRESHOLD = v;
    else if (n == "equalize") EQUALIZE = (int)v;
    else if (n == "show_track") SHOW_TRACK = (int)v;
    else if (n == "fisheye") FISHEYE = (int)v;
    else { std::fprintf(stderr, "vfe_set_option: unknown option %s\n", name); std::abort(); }
}
void vfe_set_fisheye_mask(const unsigned char* m, int w, int h) {
    for (int i = 0; i < NUM_OF_CAM; ++i) {
        trackerData[i].fisheye_mask = cv::Mat(h, w, CV_8UC1);
        std::memcpy(trackerData[i].fisheye_mask.data, m, (size_t)w * h);
    }
}
// one sensor_msgs/Image (mono8) on IMAGE_TOPIC
void vfe_image(double stamp, const unsigned char* data, int w, int h, int step) {
    sensor_msgs::ImagePtr msg(new sensor_msgs::Image);
    msg->header.stamp.fromSec(stamp);
    msg->height = h; msg->width = w; msg->step = step; msg->encoding = "mono8";
    msg->data.assign(data, data + (size_t)step * h);
    img_callback(msg);
}
// readImage() + the updateID() loop of img_callback (:86-111) with PUB_THIS_FRAME given by the caller instead of the frequency
// gate (the stand-alone replay harness `vins_replay fe` publishes every k-th frame)
void vfe_read_image_direct(double stamp, const unsigned char* data, int w, int h, int step, int pub_this_frame) {
    PUB_THIS_FRAME = pub_this_frame != 0;
    cv::Mat img(h, w, CV_8UC1, const_cast<unsigned char*>(data), (size_t)step);
    trackerData[0].readImage(img.clone(), stamp);
    for (unsigned int i = 0;; i++)
        if (!trackerData[0].updateID(i)) break;
}
int vfe_pub_this_frame() { return PUB_THIS_FRAME ? 1 : 0; }
int vfe_first_image_flag() { return first_image_flag ? 1 : 0; }
int vfe_n_id() { return FeatureTracker::n_id; }
// the members feature_tracker_node.cpp reads after readImage() (:129-150), camera 0
int vfe_track_count() { return (int)trackerData[0].ids.size(); }
void vfe_tracks(int* ids, int* track_cnt, float* cur_pts, float* cur_un_pts, float* pts_velocity) {
    const FeatureTracker& t = trackerData[0];
    const size_t n = t.ids.size();
    if (t.track_cnt.size() != n || t.cur_pts.size() != n || t.cur_un_pts.size() != n || t.pts_velocity.size() != n) {
        std::fprintf(stderr, "vfe_tracks: member vectors of FeatureTracker disagree in length (%zu %zu %zu %zu %zu)\n", n, t.track_cnt.size(), t.cur_pts.size(),
                     t.cur_un_pts.size(), t.pts_velocity.size());
        std::abort();
    }
    for (size_t i = 0; i < n; ++i) {
        ids[i] = t.ids[i]; track_cnt[i] = t.track_cnt[i];
        cur_pts[2 * i] = t.cur_pts[i].x; cur_pts[2 * i + 1] = t.cur_pts[i].y;
        cur_un_pts[2 * i] = t.cur_un_pts[i].x; cur_un_pts[2 * i + 1] = t.cur_un_pts[i].y;
        pts_velocity[2 * i] = t.pts_velocity[i].x; pts_velocity[2 * i + 1] = t.pts_velocity[i].y;
    }
}
// what went out on the `feature` topic so far; rows of 8 floats: x y z id u v vx vy (feature_tracker_node.cpp:135-156)
int vfe_published_count() { return (int)clouds().size(); }
int vfe_published_size(int k) { return (int)clouds()[k]->points.size(); }
double vfe_published_stamp(int k) { return clouds()[k]->header.stamp.toSec(); }
void vfe_published(int k, float* rows8) {
    const sensor_msgs::PointCloud& c = *clouds()[k];
    if (c.channels.size() != 5) { std::fprintf(stderr, "vfe_published: %zu channels\n", c.channels.size()); std::abort(); }
    for (size_t i = 0; i < c.points.size(); ++i) {
        float* r = rows8 + 8 * i;
        r[0] = c.points[i].x; r[1] = c.points[i].y; r[2] = c.points[i].z;
        for (int ch = 0; ch < 5; ++ch) r[3 + ch] = c.channels[ch].values[i];
    }
}
int vfe_restart_count() {
    int n = 0;
    for (const auto& m : ros::captured()) n += m.topic == "restart";
    return n;
}
// PinholeCamera::liftProjective of the camera the node loaded (camera_model/src/camera_models/PinholeCamera.cc:450-510)
void vfe_lift(const double* uv, int n, double* xyz) {
    for (int i = 0; i < n; ++i) {
        Eigen::Vector3d P;
        trackerData[0].m_camera->liftProjective(Eigen::Vector2d(uv[2 * i], uv[2 * i + 1]), P);
        xyz[3 * i] = P.x(); xyz[3 * i + 1] = P.y(); xyz[3 * i + 2] = P.z();
    }
}
// FeatureTracker::setMask alone on given tracks (feature_tracker.cpp:36-69): kept tracks back, in kept order
int vfe_set_mask(const float* pts, const int* ids, const int* cnt, int n, float* pts_out, int* ids_out, int* cnt_out) {
    FeatureTracker& t = trackerData[0];
    t.forw_pts.clear(); t.ids.clear(); t.track_cnt.clear();
    for (int i = 0; i < n; ++i) { t.forw_pts.push_back(cv::Point2f(pts[2 * i], pts[2 * i + 1])); t.ids.push_back(ids[i]); t.track_cnt.push_back(cnt[i]); }
    t.setMask();
    const int k = (int)t.forw_pts.size();
    for (int i = 0; i < k; ++i) { pts_out[2 * i] = t.forw_pts[i].x; pts_out[2 * i + 1] = t.forw_pts[i].y; ids_out[i] = t.ids[i]; cnt_out[i] = t.track_cnt[i]; }
    return k;
}
// the `mask` member as setMask() left it (all-reference build; the drop-in keeps its mask on the device)
int vfe_get_mask(unsigned char* out, int w, int h) {
    const cv::Mat& m = trackerData[0].mask;
    if (m.empty() || m.cols != w || m.rows != h) return 0;
    for (int y = 0; y < h; ++y) std::memcpy(out + (size_t)y * w, m.data + (size_t)y * m.step, (size_t)w);
    return 1;
}
}  // extern "C"
