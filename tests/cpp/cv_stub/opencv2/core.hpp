// COMPILE-ONLY stand-in for the handful of <opencv2/core.hpp> declarations that include/aswMethods_mi355x.hpp touches in
// its ASW_WITH_OPENCV branch: cv::Mat (data / rows / cols / step[0] / channels() / depth() / empty() and the
// (rows, cols, type) constructor), cv::Point, CV_8U / CV_32F / CV_MAKETYPE.
//
// What this is for: OpenCV is absent from this build environment, so without it the cv::Mat half of the drop-in header --
// the half a maintainer of the reference actually includes -- would never be seen by a compiler (tests/test_cpp_shim.py
// builds tests/cpp/shim_demo.cpp against it with -DASW_WITH_OPENCV).  What it is NOT: it is not OpenCV, it is not used to
// build or run the reference, and nothing that compiles against it is parity evidence -- it only shows that view() /
// make() / AswPoint and every cv::Mat expression of the header are well-formed against the member names and types OpenCV
// documents (cv::Mat::step is a MatStep convertible to size_t and indexable, depth() is CV_MAT_DEPTH(flags), ...).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#define CV_CN_SHIFT 3
#define CV_DEPTH_MAX (1 << CV_CN_SHIFT)
#define CV_8U 0
#define CV_32F 5
#define CV_MAT_DEPTH_MASK (CV_DEPTH_MAX - 1)
#define CV_MAT_DEPTH(flags) ((flags) & CV_MAT_DEPTH_MASK)
#define CV_MAKETYPE(depth, cn) (CV_MAT_DEPTH(depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_MAT_CN(flags) ((((flags) >> CV_CN_SHIFT) & 511) + 1)

namespace cv {

struct MatStep {
    size_t p[2] = {0, 0};
    size_t& operator[](int i) { return p[i]; }
    const size_t& operator[](int i) const { return p[i]; }
    operator size_t() const { return p[0]; }
};

class Mat {
public:
    int flags = 0, rows = 0, cols = 0;
    unsigned char* data = nullptr;
    MatStep step;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type)
    {
        flags = type; rows = r; cols = c;
        const size_t esz = (size_t)channels() * (depth() == CV_32F ? 4 : 1);
        step[0] = (size_t)c * esz; step[1] = esz;
        buf_ = std::make_shared<std::vector<unsigned char>>(step[0] * (size_t)r);
        data = buf_->data();
    }
    int type() const { return flags & 4095; }
    int depth() const { return CV_MAT_DEPTH(flags); }
    int channels() const { return CV_MAT_CN(flags); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }

private:
    std::shared_ptr<std::vector<unsigned char>> buf_;  // reference-counted pixels, like cv::Mat's header copies
};

struct Point {
    int x = 0, y = 0;
    Point() {}
    Point(int x_, int y_) : x(x_), y(y_) {}
};

}  // namespace cv
