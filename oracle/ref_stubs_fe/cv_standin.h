// TEST INFRASTRUCTURE — stand-in for the OpenCV headers the reference's FRONT END includes (feature_tracker/src/*.cpp,
// camera_model/src/camera_models/*.cc), so that those translation units compile UNCHANGED in an image without OpenCV
// (oracle/Makefile, target ref_fe -> oracle/_ref/libvins_ref_fe.so).
//
// What is functional and what it stands on:
//   cv::Mat / Point_ / Size_ / Scalar / Ptr / FileStorage(READ)   plain containers, written here
//   cv::createCLAHE()->apply, cv::calcOpticalFlowPyrLK, cv::goodFeaturesToTrack, cv::findFundamentalMat(FM_RANSAC)
//        FORWARD to oracle/fe_cpu.cpp (the restatement of OpenCV 3.3's algorithms, ASSUMPTIONS.md F1-F9).  PARITY UNPINNED for
//        these five calls: OpenCV itself is absent.  Everything AROUND them — FeatureTracker::readImage / setMask /
//        rejectWithF / undistortedPoints / updateID, the node's PUB_THIS_FRAME gate, PinholeCamera::liftProjective — is the
//        reference's own code, compiled from /root/reference where it lies.
//   cv::circle(mask, c, r, 0, -1)   the integer midpoint loop of drawing.cpp Circle() as recalled (ASSUMPTIONS.md F7 addendum)
// Everything the front end never reaches (calibration: findHomography, solve, solvePnP, convertMaps, imshow ...) aborts.
#ifndef VINS_REF_FE_CV_STANDIN_H
#define VINS_REF_FE_CV_STANDIN_H
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>
#include "yaml_config.h"      // vins-mono_amd/host: the configuration reader (cv::FileStorage stand-in reads through it)

typedef unsigned char uchar;

extern "C" {      // oracle/fe_cpu.cpp
void oracle_fe_lk(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_xy, int n, int max_level, float* next_xy, uint8_t* status, float* err);
int oracle_fe_gftt(const uint8_t* img, int w, int h, const uint8_t* mask, int max_corners, double quality, double min_dist, float* out_xy);
int oracle_fe_clahe(const uint8_t* src, int w, int h, double clip, uint8_t* dst);
int oracle_fe_reject_with_f(const float* p1, const float* p2, int n, double threshold, uint8_t* status, double* F_out);
}

#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32F 5
#define CV_32FC1 5
#define CV_64F 6
#define CV_64FC1 6
#define CV_GRAY2RGB 8
#define CV_GRAY2BGR 8

namespace cv {
[[noreturn]] inline void vins_ref_fe_unreachable(const char* what) {
    std::fprintf(stderr, "oracle/_ref (front end): cv::%s reached — outside the FeatureTracker::readImage path (calibration / display)\n", what);
    std::abort();
}
inline int cvRound(double v) { return (int)std::lrint(v); }        // round half to even (SSE2 cvtsd2si)
template <typename T> inline T saturate_cast(float v);
template <> inline int saturate_cast<int>(float v) { return cvRound(v); }
template <> inline float saturate_cast<float>(float v) { return v; }
template <> inline double saturate_cast<double>(float v) { return v; }

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T a, T b) : x(a), y(b) {}
    template <typename U> operator Point_<U>() const;
    bool operator==(const Point_& o) const { return x == o.x && y == o.y; }
    Point_ operator-(const Point_& o) const { return Point_(x - o.x, y - o.y); }
    Point_ operator+(const Point_& o) const { return Point_(x + o.x, y + o.y); }
};
template <typename A, typename B> struct PointCast { static B c(A v) { return static_cast<B>(v); } };
template <> struct PointCast<float, int> { static int c(float v) { return cvRound(v); } };     // saturate_cast<int>(float)
template <> struct PointCast<double, int> { static int c(double v) { return cvRound(v); } };
template <typename T> template <typename U> Point_<T>::operator Point_<U>() const { return Point_<U>(PointCast<T, U>::c(x), PointCast<T, U>::c(y)); }
template <typename T> struct Point3_ {
    T x, y, z;
    Point3_() : x(0), y(0), z(0) {}
    Point3_(T a, T b, T c) : x(a), y(b), z(c) {}
};
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;
template <typename T> struct Size_ {
    T width, height;
    Size_() : width(0), height(0) {}
    Size_(T w, T h) : width(w), height(h) {}
    bool operator==(const Size_& o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size_& o) const { return !(*this == o); }
};
typedef Size_<int> Size;
struct Scalar {
    double val[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
};
template <typename T> using Ptr = std::shared_ptr<T>;

// dense, row-major, reference-counted; depth = type & 7, channels = (type >> 3) + 1
class Mat {
  public:
    int rows = 0, cols = 0, flags = 0;
    size_t step = 0;
    uchar* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    Mat(int r, int c, int type, const Scalar& s) { create(r, c, type); fill(s); }
    Mat(int r, int c, int type, void* ext, size_t step_ = 0) : rows(r), cols(c), flags(type), data((uchar*)ext) { step = step_ ? step_ : (size_t)c * elemSize(); }
    void create(int r, int c, int type) {
        rows = r; cols = c; flags = type; step = (size_t)c * elemSize();
        buf_ = std::make_shared<std::vector<uchar>>((size_t)r * step, (uchar)0);
        data = buf_->data();
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat zeros(Size s, int type) { return Mat(s, type); }
    static Mat eye(int r, int c, int type) {
        Mat m(r, c, type);
        for (int i = 0; i < std::min(r, c); ++i) {
            if ((type & 7) == CV_32F) m.at<float>(i, i) = 1.f;
            else if ((type & 7) == CV_64F) m.at<double>(i, i) = 1.0;
            else m.at<uchar>(i, i) = 1;
        }
        return m;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return flags; }
    int depth() const { return flags & 7; }
    int channels() const { return (flags >> 3) + 1; }
    size_t elemSize() const { const int d = flags & 7; return (size_t)(d == CV_8U ? 1 : d == CV_32F ? 4 : d == CV_64F ? 8 : 1) * ((flags >> 3) + 1); }
    Size size() const { return Size(cols, rows); }
    Mat clone() const {
        Mat m(rows, cols, flags);
        for (int y = 0; y < rows; ++y) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * elemSize());
        return m;
    }
    Mat rowRange(int a, int b) const { Mat m = *this; m.data = data + (size_t)a * step; m.rows = b - a; return m; }
    void release() { *this = Mat(); }
    template <typename T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step + (size_t)x * sizeof(T)); }
    template <typename T> const T& at(int y, int x) const { return *(const T*)(data + (size_t)y * step + (size_t)x * sizeof(T)); }
    template <typename T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }          // vectors (1 x n or n x 1)
    template <typename T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <typename T> T& at(Point p) { return at<T>(p.y, p.x); }
    template <typename T> const T& at(Point p) const { return at<T>(p.y, p.x); }
    template <typename T> T* ptr(int y = 0) { return (T*)(data + (size_t)y * step); }
    template <typename T> const T* ptr(int y = 0) const { return (const T*)(data + (size_t)y * step); }
    // a contiguous 8-bit copy (the oracle functions take w * h bytes)
    std::vector<uchar> bytes() const {
        std::vector<uchar> v((size_t)rows * cols);
        for (int y = 0; y < rows; ++y) std::memcpy(v.data() + (size_t)y * cols, data + (size_t)y * step, (size_t)cols);
        return v;
    }
  private:
    void fill(const Scalar& s) {
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols * channels(); ++x) {
                const double v = s.val[x % channels()];
                if (depth() == CV_32F) ((float*)(data + (size_t)y * step))[x] = (float)v;
                else if (depth() == CV_64F) ((double*)(data + (size_t)y * step))[x] = v;
                else (data + (size_t)y * step)[x] = (uchar)v;
            }
    }
    std::shared_ptr<std::vector<uchar>> buf_;
};
typedef const Mat& InputArray;
typedef Mat& InputOutputArray;
struct OutputArray {           // only ever a sink here (cv::noArray(), perViewErrors of the calibration code)
    Mat* m = nullptr;
    OutputArray() {}
    OutputArray(Mat& mm) : m(&mm) {}
    bool needed() const { return m != nullptr; }
    void create(int, int, int) const { vins_ref_fe_unreachable("OutputArray::create"); }
    Mat getMat() const { vins_ref_fe_unreachable("OutputArray::getMat"); }
};
inline OutputArray noArray() { return OutputArray(); }

// ---- drawing.cpp: cv::circle -> Circle(img, center, radius, color, fill) for thickness < 0, LINE_8, shift 0 (8-bit, 1 channel).
// Outline circles (thickness > 0) only occur in the node's SHOW_TRACK visualisation: ignored.
inline void circle(Mat& img, Point center, int radius, const Scalar& color, int thickness = 1, int = 8, int = 0) {
    if (thickness >= 0) return;
    if (img.type() != CV_8UC1) vins_ref_fe_unreachable("circle on a non-8UC1 image");
    const uchar col = (uchar)color.val[0];
    const int W = img.cols, H = img.rows;
    auto hline = [&](int y, int x0, int x1) {
        if ((unsigned)y >= (unsigned)H) return;
        x0 = std::max(x0, 0); x1 = std::min(x1, W - 1);
        for (int x = x0; x <= x1; ++x) img.at<uchar>(y, x) = col;
    };
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        hline(center.y - dy, center.x - dx, center.x + dx);
        hline(center.y + dy, center.x - dx, center.x + dx);
        hline(center.y - dx, center.x - dy, center.x + dy);
        hline(center.y + dx, center.x - dy, center.x + dy);
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

// ---- imgproc: CLAHE (clahe.cpp) -> oracle_fe_clahe
class CLAHE {
  public:
    CLAHE(double clip, Size tiles) : clip_(clip), tiles_(tiles) {}
    void apply(const Mat& src, Mat& dst) {
        if (src.type() != CV_8UC1 || tiles_.width != 8 || tiles_.height != 8) vins_ref_fe_unreachable("CLAHE::apply (8x8 tiles on 8UC1 only)");
        std::vector<uchar> in = src.bytes();
        Mat out(src.rows, src.cols, CV_8UC1);
        if (oracle_fe_clahe(in.data(), src.cols, src.rows, clip_, out.data) != 0) vins_ref_fe_unreachable("CLAHE::apply (image size must divide by the tile grid)");
        dst = out;
    }
  private:
    double clip_;
    Size tiles_;
};
inline Ptr<CLAHE> createCLAHE(double clipLimit = 40.0, Size tileGridSize = Size(8, 8)) { return std::make_shared<CLAHE>(clipLimit, tileGridSize); }
inline void cvtColor(const Mat&, Mat&, int) {}                         // SHOW_TRACK visualisation only
inline void convertMaps(const Mat&, const Mat&, Mat&, Mat&, int, bool = false) { vins_ref_fe_unreachable("convertMaps"); }

// ---- video: calcOpticalFlowPyrLK (lkpyramid.cpp) -> oracle_fe_lk; default TermCriteria(COUNT + EPS, 30, 0.01), flags 0, minEigThreshold 1e-4
inline void calcOpticalFlowPyrLK(const Mat& prevImg, const Mat& nextImg, const std::vector<Point2f>& prevPts, std::vector<Point2f>& nextPts,
                                 std::vector<uchar>& status, std::vector<float>& err, Size winSize = Size(21, 21), int maxLevel = 3) {
    if (winSize.width != 21 || winSize.height != 21 || prevImg.size() != nextImg.size() || prevImg.type() != CV_8UC1) vins_ref_fe_unreachable("calcOpticalFlowPyrLK (21x21 window on 8UC1 only)");
    const int n = (int)prevPts.size();
    nextPts.resize(n); status.resize(n); err.resize(n);
    if (!n) return;
    std::vector<uchar> a = prevImg.bytes(), b = nextImg.bytes();
    oracle_fe_lk(a.data(), b.data(), prevImg.cols, prevImg.rows, &prevPts[0].x, n, maxLevel, &nextPts[0].x, status.data(), err.data());
}

// ---- imgproc: goodFeaturesToTrack (featureselect.cpp) -> oracle_fe_gftt (blockSize 3, Harris off)
inline void goodFeaturesToTrack(const Mat& image, std::vector<Point2f>& corners, int maxCorners, double qualityLevel, double minDistance, const Mat& mask = Mat()) {
    if (image.type() != CV_8UC1) vins_ref_fe_unreachable("goodFeaturesToTrack (8UC1 only)");
    std::vector<uchar> img = image.bytes(), m;
    if (!mask.empty()) m = mask.bytes();
    // maxCorners <= 0 means "no limit" in OpenCV; the reference only calls with a positive count (feature_tracker.cpp:141-149)
    const int cap = maxCorners > 0 ? maxCorners : image.rows * image.cols;
    std::vector<float> xy((size_t)cap * 2);
    const int n = oracle_fe_gftt(img.data(), image.cols, image.rows, m.empty() ? nullptr : m.data(), cap, qualityLevel, minDistance, xy.data());
    corners.resize(n);
    for (int i = 0; i < n; ++i) corners[i] = Point2f(xy[2 * i], xy[2 * i + 1]);
}

// ---- calib3d
enum { FM_7POINT = 1, FM_8POINT = 2, FM_LMEDS = 4, FM_RANSAC = 8 };
enum { DECOMP_LU = 0, DECOMP_SVD = 1, DECOMP_NORMAL = 16 };
inline Mat findFundamentalMat(const std::vector<Point2f>& p1, const std::vector<Point2f>& p2, int method, double threshold, double confidence, std::vector<uchar>& status) {
    if (method != FM_RANSAC || confidence != 0.99 || p1.size() != p2.size()) vins_ref_fe_unreachable("findFundamentalMat (FM_RANSAC, 0.99 only)");
    const int n = (int)p1.size();
    status.assign(n, 1);
    Mat F(3, 3, CV_64F);
    if (n) oracle_fe_reject_with_f(&p1[0].x, &p2[0].x, n, threshold, status.data(), F.ptr<double>());
    return F;
}
template <typename A, typename B> inline Mat findHomography(const A&, const B&) { vins_ref_fe_unreachable("findHomography"); }
inline bool solve(const Mat&, const Mat&, Mat&, int = 0) { vins_ref_fe_unreachable("solve"); }
template <typename P3, typename P2, typename D> inline bool solvePnP(const std::vector<P3>&, const std::vector<P2>&, const Mat&, const D&, Mat&, Mat&, bool = false, int = 0) { vins_ref_fe_unreachable("solvePnP"); }
inline void Rodrigues(const Mat&, Mat&) { vins_ref_fe_unreachable("Rodrigues"); }
template <typename A> inline double norm(const A&) { vins_ref_fe_unreachable("norm"); }
struct SVD { static void solveZ(const Mat&, Mat&) { vins_ref_fe_unreachable("SVD::solveZ"); } };

// ---- highgui
inline void imshow(const std::string&, const Mat&) { vins_ref_fe_unreachable("imshow"); }
inline int waitKey(int = 0) { vins_ref_fe_unreachable("waitKey"); }
// imread(path, 0): no image codecs here — a binary PGM ("P5 w h 255") is read whatever the file is called (the tests write the fisheye
// mask of feature_tracker_node.cpp:216 that way); anything else is an empty Mat, as an unreadable file is in OpenCV
inline Mat imread(const std::string& path, int = 1) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return Mat();
    int w = 0, h = 0, mx = 0;
    Mat m;
    if (std::fscanf(f, "P5 %d %d %d", &w, &h, &mx) == 3 && mx == 255 && w > 0 && h > 0 && std::fgetc(f) != EOF) {
        m = Mat(h, w, CV_8UC1);
        if (std::fread(m.data, 1, (size_t)w * h, f) != (size_t)w * h) m = Mat();
    }
    std::fclose(f);
    return m;
}

// ---- cv::FileStorage: READ through the product's configuration reader; WRITE is a sink (calibration output, never on this path)
class FileNode {
  public:
    FileNode() {}
    FileNode(const VinsYaml* y, const std::string& key) : y_(y), key_(key) {}
    FileNode operator[](const std::string& k) const { return FileNode(y_, key_ + "." + k); }
    FileNode operator[](const char* k) const { return FileNode(y_, key_ + "." + k); }
    bool isNone() const { return !y_ || !y_->has(key_); }
    bool empty() const { return isNone(); }
    operator int() const { return (int)y_->number(key_); }
    operator float() const { return (float)y_->number(key_); }
    operator double() const { return y_->number(key_); }
    operator std::string() const { return y_->str(key_); }
    const VinsYaml* y_ = nullptr;
    std::string key_;
};
inline void operator>>(const FileNode& n, std::string& v) { v = n.y_->str(n.key_); }
inline void operator>>(const FileNode& n, int& v) { v = (int)n.y_->number(n.key_); }
inline void operator>>(const FileNode& n, double& v) { v = n.y_->number(n.key_); }
inline void operator>>(const FileNode& n, Mat& m) {
    const VinsYaml::Matrix* src = n.y_->matrix(n.key_);
    m = Mat();
    if (!src) return;
    m = Mat(src->rows, src->cols, CV_64F);
    for (int r = 0; r < src->rows; ++r)
        for (int c = 0; c < src->cols; ++c) m.at<double>(r, c) = src->data[(size_t)r * src->cols + c];
}
class FileStorage {
  public:
    enum Mode { READ = 0, WRITE = 1 };
    FileStorage() {}
    FileStorage(const std::string& path, int mode) : write_(mode == WRITE) { if (!write_) y_.load(path); }
    bool isOpened() const { return write_ || y_.opened(); }
    FileNode operator[](const std::string& key) const { return FileNode(&y_, key); }
    FileNode operator[](const char* key) const { return FileNode(&y_, key); }
    void release() {}
    template <typename T> FileStorage& operator<<(const T&) { if (!write_) vins_ref_fe_unreachable("FileStorage << on a READ storage"); return *this; }
  private:
    VinsYaml y_;
    bool write_ = false;
};
}  // namespace cv
using cv::cvRound;
#endif
