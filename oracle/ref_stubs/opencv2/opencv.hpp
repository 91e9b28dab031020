// TEST INFRASTRUCTURE — stand-in for <opencv2/opencv.hpp>.  Only the NAMES the estimator translation units mention
// (members of MotionEstimator / InitialEXRotation, locals of Estimator::initialStructure) exist; every function aborts.
// Those code paths belong to the initialisation (vins_estimator/src/initial/*), which SURVEY.md section 8 puts out of
// scope; the _ref driver never reaches them (it starts the estimator in NON_LINEAR mode).
#ifndef VINS_REF_STUB_OPENCV_HPP
#define VINS_REF_STUB_OPENCV_HPP
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "yaml_config.h"      // vins-mono_amd/host (the product's configuration reader)
namespace cv {
[[noreturn]] inline void vins_ref_unreachable(const char *what) {
    std::fprintf(stderr, "oracle/_ref: cv::%s reached — the initialisation path is out of scope for this build\n", what);
    std::abort();
}
template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    template <typename A, typename B> Point_(A a, B b) : x(static_cast<T>(a)), y(static_cast<T>(b)) {}
};
template <typename T> struct Point3_ {
    T x, y, z;
    Point3_() : x(0), y(0), z(0) {}
    template <typename A, typename B, typename C> Point3_(A a, B b, C c) : x(static_cast<T>(a)), y(static_cast<T>(b)), z(static_cast<T>(c)) {}
};
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;
class Mat {
  public:
    int rows = 0, cols = 0;
    std::vector<double> data;      // row-major doubles (what the configuration matrices hold: dt: d)
    Mat() {}
};
template <typename T> class Mat_;
template <typename T> struct MatCommaInitializer_ {
    template <typename V> MatCommaInitializer_ &operator,(V) { return *this; }
    operator Mat() const { return Mat(); }
    operator Mat_<T>() const;
};
template <typename T> class Mat_ : public Mat {
  public:
    Mat_() {}
    Mat_(int, int) {}
    Mat_(const Mat &) {}
    template <typename V> MatCommaInitializer_<T> operator<<(V) { return MatCommaInitializer_<T>(); }
    T &operator()(int, int) { vins_ref_unreachable("Mat_::operator()"); }
};
template <typename T> MatCommaInitializer_<T>::operator Mat_<T>() const { return Mat_<T>(); }
inline void Rodrigues(const Mat &, Mat &) { vins_ref_unreachable("Rodrigues"); }

// cv::FileStorage (READ) as far as {vins_estimator,feature_tracker}/src/parameters.cpp use it, implemented by the PRODUCT's
// configuration reader (vins-mono_amd/host/yaml_config.h): the reference's readParameters() compiled unchanged against this
// stand-in is the check of that reader (tests/test_config_reader.py).
class FileNode {
  public:
    FileNode(const VinsYaml *y, const std::string &key) : y_(y), key_(key) {}
    operator int() const { return static_cast<int>(y_->number(key_)); }
    operator float() const { return static_cast<float>(y_->number(key_)); }
    operator double() const { return y_->number(key_); }
    operator std::string() const { return y_->str(key_); }
    bool empty() const { return !y_->has(key_); }
    const VinsYaml *y_;
    std::string key_;
};
inline void operator>>(const FileNode &n, std::string &v) { v = n.y_->str(n.key_); }
inline void operator>>(const FileNode &n, int &v) { v = static_cast<int>(n.y_->number(n.key_)); }
inline void operator>>(const FileNode &n, double &v) { v = n.y_->number(n.key_); }
inline void operator>>(const FileNode &n, Mat &m) {
    const VinsYaml::Matrix *src = n.y_->matrix(n.key_);
    m = Mat();
    if (src) { m.rows = src->rows; m.cols = src->cols; m.data = src->data; }
}
class FileStorage {
  public:
    enum Mode { READ = 0, WRITE = 1 };
    FileStorage() {}
    FileStorage(const std::string &path, int) { y_.load(path); }
    bool isOpened() const { return y_.opened(); }
    FileNode operator[](const std::string &key) const { return FileNode(&y_, key); }
    FileNode operator[](const char *key) const { return FileNode(&y_, key); }
    void release() {}
  private:
    VinsYaml y_;
};
template <typename P3, typename P2>
inline bool solvePnP(const std::vector<P3> &, const std::vector<P2> &, const Mat &, const Mat &, Mat &, Mat &, bool = false, int = 0) {
    vins_ref_unreachable("solvePnP");
}
}  // namespace cv
#endif
