// TEST INFRASTRUCTURE — stand-in for <opencv2/opencv.hpp>.  Only the NAMES the estimator translation units mention
// (members of MotionEstimator / InitialEXRotation, locals of Estimator::initialStructure) exist; every function aborts.
// Those code paths belong to the initialisation (vins_estimator/src/initial/*), which SURVEY.md section 8 puts out of
// scope; the _ref driver never reaches them (it starts the estimator in NON_LINEAR mode).
#ifndef VINS_REF_STUB_OPENCV_HPP
#define VINS_REF_STUB_OPENCV_HPP
#include <cstdio>
#include <cstdlib>
#include <vector>
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
template <typename P3, typename P2>
inline bool solvePnP(const std::vector<P3> &, const std::vector<P2> &, const Mat &, const Mat &, Mat &, Mat &, bool = false, int = 0) {
    vins_ref_unreachable("solvePnP");
}
}  // namespace cv
#endif
