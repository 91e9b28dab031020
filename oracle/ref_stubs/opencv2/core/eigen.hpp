// TEST INFRASTRUCTURE — stand-in for <opencv2/core/eigen.hpp> (see ../opencv.hpp: names only, every call aborts).
#ifndef VINS_REF_STUB_OPENCV_EIGEN_HPP
#define VINS_REF_STUB_OPENCV_EIGEN_HPP
#include "../opencv.hpp"
namespace cv {
template <typename E> inline void eigen2cv(const E &, Mat &) { vins_ref_unreachable("eigen2cv"); }
template <typename E> inline void cv2eigen(const Mat &m, E &e) {      // functional: readParameters() converts the extrinsics with it
    e.resize(m.rows, m.cols);
    for (int r = 0; r < m.rows; r++)
        for (int c = 0; c < m.cols; c++) e(r, c) = m.data[static_cast<size_t>(r) * m.cols + c];
}
}  // namespace cv
#endif
