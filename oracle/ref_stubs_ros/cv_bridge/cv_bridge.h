// TEST INFRASTRUCTURE — stand-in for <cv_bridge/cv_bridge.h>: toCvCopy of a mono8 sensor_msgs::Image = a deep copy of its
// rows into a cv::Mat (cv_bridge.cpp: same encoding, no conversion); cvtColor / toImageMsg only serve the SHOW_TRACK picture.
#ifndef VINS_REF_FE_CV_BRIDGE_H
#define VINS_REF_FE_CV_BRIDGE_H
#include <cstring>
#include <string>
#include <boost/shared_ptr.hpp>
#include <opencv2/core/core.hpp>
#include <sensor_msgs/Image.h>
namespace cv_bridge {
class CvImage;
typedef boost::shared_ptr<CvImage> CvImagePtr;
typedef boost::shared_ptr<CvImage const> CvImageConstPtr;
class CvImage {
  public:
    std_msgs::Header header;
    std::string encoding;
    cv::Mat image;
    sensor_msgs::ImagePtr toImageMsg() const {
        sensor_msgs::ImagePtr m(new sensor_msgs::Image);
        m->header = header; m->encoding = encoding; m->height = image.rows; m->width = image.cols;
        return m;
    }
};
inline CvImagePtr toCvCopy(const sensor_msgs::Image& src, const std::string& encoding = std::string()) {
    if (src.encoding != "mono8" || (!encoding.empty() && encoding != "mono8")) {
        std::fprintf(stderr, "oracle/_ref (front end): cv_bridge::toCvCopy of encoding '%s' is outside this stand-in (mono8 only)\n", src.encoding.c_str());
        std::abort();
    }
    CvImagePtr out(new CvImage);
    out->header = src.header; out->encoding = "mono8";
    out->image = cv::Mat((int)src.height, (int)src.width, CV_8UC1);
    for (uint32_t y = 0; y < src.height; ++y) std::memcpy(out->image.data + (size_t)y * out->image.step, src.data.data() + (size_t)y * src.step, src.width);
    return out;
}
inline CvImagePtr toCvCopy(const sensor_msgs::ImageConstPtr& src, const std::string& encoding = std::string()) { return toCvCopy(*src, encoding); }
inline CvImagePtr cvtColor(const CvImageConstPtr& src, const std::string& encoding) {
    CvImagePtr out(new CvImage(*src));
    out->encoding = encoding;
    return out;
}
}  // namespace cv_bridge
#endif
