// TEST INFRASTRUCTURE — stand-in for <opencv2/core/eigen.hpp> on the functional cv::Mat of ../../cv_standin.h
#ifndef VINS_REF_FE_CV_EIGEN_HPP
#define VINS_REF_FE_CV_EIGEN_HPP
#include "../../cv_standin.h"
namespace cv {
template <typename E> inline void eigen2cv(const E& e, Mat& m) {
    m = Mat((int)e.rows(), (int)e.cols(), CV_64F);
    for (int r = 0; r < (int)e.rows(); r++)
        for (int c = 0; c < (int)e.cols(); c++) m.at<double>(r, c) = e(r, c);
}
template <typename E> inline void cv2eigen(const Mat& m, E& e) {
    e.resize(m.rows, m.cols);
    for (int r = 0; r < m.rows; r++)
        for (int c = 0; c < m.cols; c++) e(r, c) = m.depth() == CV_64F ? m.at<double>(r, c) : (double)m.at<float>(r, c);
}
}  // namespace cv
#endif
