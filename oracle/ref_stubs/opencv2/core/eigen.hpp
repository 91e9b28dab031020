// TEST INFRASTRUCTURE — stand-in for <opencv2/core/eigen.hpp> (see ../opencv.hpp: names only, every call aborts).
#ifndef VINS_REF_STUB_OPENCV_EIGEN_HPP
#define VINS_REF_STUB_OPENCV_EIGEN_HPP
#include "../opencv.hpp"
namespace cv {
template <typename E> inline void eigen2cv(const E &, Mat &) { vins_ref_unreachable("eigen2cv"); }
template <typename E> inline void cv2eigen(const Mat &, E &) { vins_ref_unreachable("cv2eigen"); }
}  // namespace cv
#endif
