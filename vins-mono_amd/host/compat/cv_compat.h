// compat/cv_compat.h — the tiny subset of OpenCV types that appears in the FeatureTracker class interface
// (feature_tracker/src/feature_tracker.h:28-65), so that the drop-in class compiles in this container where
// OpenCV is absent.  In a real catkin workspace build with -DVINS_REAL_OPENCV and include <opencv2/opencv.hpp>
// instead: every expression used in feature_tracker.cpp is source-compatible.  Nothing here crosses the C-ABI.
#pragma once
#ifndef VINS_REAL_OPENCV
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
namespace cv {
struct Point2f {
    float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
typedef unsigned char uchar;
enum { CV_8UC1 = 0 };
// ref-counted 8-bit single-channel image view
class Mat {
  public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uchar* data = nullptr;
    Mat() {}
    Mat(int r, int c, int /*type*/, uchar fill) : rows(r), cols(c), step((size_t)c), buf_(new std::vector<uchar>((size_t)r * c, fill)) { data = buf_->data(); }
    Mat(int r, int c, int /*type*/, void* ext, size_t step_) : rows(r), cols(c), step(step_), data((uchar*)ext) {}
    bool empty() const { return data == nullptr; }
    int type() const { return CV_8UC1; }
    Mat clone() const {
        Mat m(rows, cols, CV_8UC1, (uchar)0);
        for (int y = 0; y < rows; ++y) memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols);
        return m;
    }
    Mat rowRange(int a, int b) const { Mat m = *this; m.data = data + (size_t)a * step; m.rows = b - a; return m; }
    template <typename T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step + x * sizeof(T)); }
    template <typename T> const T& at(int y, int x) const { return *(const T*)(data + (size_t)y * step + x * sizeof(T)); }
  private:
    std::shared_ptr<std::vector<uchar>> buf_;
};
}  // namespace cv
typedef unsigned char uchar;      // OpenCV exports `uchar` at global scope
#else
#include <opencv2/opencv.hpp>
#endif
