// aswMethods_mi355x.hpp -- C++ surface of the reference (aswStereoMatch/methods/aswMethods.h, "M.h") on
// top of the C-ABI of asw_mi355x.h.  Header-only.
//
//  * With OpenCV available (#include <opencv2/core.hpp> found, or ASW_WITH_OPENCV defined) the functions
//    take and return cv::Mat with EXACTLY the reference signatures, so a maintainer of the reference
//    replaces `#include "methods/aswMethods.h"` by this header and links libasw_mi355x.so
//    (INTEGRATION.md).
//  * Without OpenCV the same functions are available on asw::Mat, a minimal stand-in for the part of
//    cv::Mat the path uses (rows, cols, type, step, data, empty()).
//
// Error behaviour mirrors the reference: where it returns silently or an empty Mat (size mismatch, even
// window) so does the shim; statuses with no reference equivalent (HIP error, unsupported method) throw
// std::runtime_error -- the reference throws cv::Exception in comparable situations (M.cpp:103-116).
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "asw_mi355x.h"

#if !defined(ASW_WITH_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#define ASW_WITH_OPENCV 1
#endif
#endif
#ifdef ASW_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

// parametersStereo.h:4-24 -- same names and values (skip when the reference header is also included)
#ifndef ASW_REFERENCE_ENUMS_DEFINED
#define ASW_REFERENCE_ENUMS_DEFINED
enum DisparityType { DISPARITY_LEFT = 0, DISPARITY_RIGHT = 1 };
enum StereoMatchingAlgorithms {
    BM = 0, SGBM = 1, ADAPTIVE_WEIGHT = 2, ADAPTIVE_WEIGHT_8DIRECT = 3, ADAPTIVE_WEIGHT_GEODESIC = 4,
    ADAPTIVE_WEIGHT_BILATERAL_GRID = 5, ADAPTIVE_WEIGHT_BLO1 = 6, ADAPTIVE_WEIGHT_GUIDED_FILTER = 7,
    ADAPTIVE_WEIGHT_GUIDED_FILTER_2 = 8, ADAPTIVE_WEIGHT_GUIDED_FILTER_3 = 9, ADAPTIVE_WEIGHT_MEDIAN = 10, NCC = 11
};
#endif

namespace asw {

// ---- minimal Mat used when OpenCV is absent --------------------------------------------------------
struct Mat {
    int rows = 0, cols = 0, channels_ = 0, depth_ = ASW_8U;
    size_t step = 0;
    std::shared_ptr<std::vector<uint8_t>> buf;
    uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int depth, int ch) { create(r, c, depth, ch); }
    void create(int r, int c, int depth, int ch)
    {
        rows = r; cols = c; depth_ = depth; channels_ = ch;
        step = (size_t)c * ch * (depth == ASW_32F ? 4 : 1);
        buf = std::make_shared<std::vector<uint8_t>>(step * r);
        data = buf->data();
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int channels() const { return channels_; }
    int depth() const { return depth_; }
};

// cv::Point where OpenCV is absent (key of getGeodesicDist's map, M.h:141)
struct Point {
    int x = 0, y = 0;
    Point() {}
    Point(int x_, int y_) : x(x_), y(y_) {}
};

namespace detail {

inline asw_ctx* context()
{
    // one lazily created context per host thread (contexts are not shared between threads)
    struct Holder {
        asw_ctx* c = nullptr;
        ~Holder() { if (c) asw_destroy(c); }
    };
    static thread_local Holder h;
    if (!h.c) {
        int rc = asw_create(0, &h.c);
        if (rc != ASW_OK) throw std::runtime_error(std::string("asw_create: ") + asw_status_string(rc));
    }
    return h.c;
}

inline bool silent(int rc) { return rc == ASW_ERR_SIZE_MISMATCH || rc == ASW_ERR_EVEN_WINDOW; }
inline void raise_unless_ok(int rc, const char* what)
{
    if (rc != ASW_OK && !silent(rc)) throw std::runtime_error(std::string(what) + ": " + asw_status_string(rc));
}

#ifdef ASW_WITH_OPENCV
typedef cv::Mat MatT;
inline asw_image view(const cv::Mat& m)
{
    asw_image v{m.data, m.rows, m.cols, m.channels(), m.depth() == CV_32F ? ASW_32F : (m.depth() == CV_8U ? ASW_8U : -1), m.step[0]};
    return v;
}
inline cv::Mat make(int r, int c, int depth, int ch) { return cv::Mat(r, c, CV_MAKETYPE(depth == ASW_32F ? CV_32F : CV_8U, ch)); }
#else
typedef asw::Mat MatT;
inline asw_image view(const asw::Mat& m) { return asw_image{m.data, m.rows, m.cols, m.channels_, m.depth_, m.step}; }
inline asw::Mat make(int r, int c, int depth, int ch) { return asw::Mat(r, c, depth, ch); }
#endif

template <typename Fn>
inline MatT aggregate(const MatT& l, const MatT& r, Fn fn, const char* what)
{
    if (l.empty() || r.empty()) return MatT();
    MatT disp = make(l.rows, l.cols, ASW_32F, 1);
    asw_image li = view(l), ri = view(r), di = view(disp);
    int rc = fn(context(), &li, &ri, &di);
    raise_unless_ok(rc, what);
    return rc == ASW_OK ? disp : MatT();  // empty Mat where the reference returns Mat()
}

template <typename T, typename Fn>
inline void cost_volume(const MatT& l, const MatT& r, std::vector<MatT>& out, int n, int depth, int pad, Fn fn, const char* what)
{
    if (!out.empty()) out.clear();  // M.cpp:222-225
    if (l.empty() || r.empty() || n <= 0) return;
    const int H = l.rows + 2 * pad, W = l.cols + 2 * pad;
    std::vector<T> vol((size_t)n * H * W);
    asw_image li = view(l), ri = view(r);
    int rc = fn(context(), &li, &ri, vol.data());
    raise_unless_ok(rc, what);
    if (rc != ASW_OK) return;  // silent return of the reference: cost_ds stays empty
    for (int k = 0; k < n; k++) {
        MatT m = make(H, W, depth, 1);
        asw_image mi = view(m);
        for (int y = 0; y < H; y++)
            std::memcpy((uint8_t*)mi.data + (size_t)y * mi.step, vol.data() + ((size_t)k * H + y) * W, (size_t)W * sizeof(T));
        out.push_back(m);
    }
}

}  // namespace detail
}  // namespace asw

// ---------------------------------------------------------------------------------------------------
// The reference's functions (M.h:91-182), global namespace like the reference
// ---------------------------------------------------------------------------------------------------
typedef asw::detail::MatT AswMat;

// M.h:91-92 / M.cpp:46-88
inline void stereoMatching(AswMat srcLeft, AswMat srcRight, AswMat& disparityMap, DisparityType disparityType,
                           StereoMatchingAlgorithms algorithmType, int winSize = 15, int minDisparity = 0, int numDisparity = 64)
{
    AswMat d = asw::detail::aggregate(srcLeft, srcRight, [&](asw_ctx* c, asw_image* l, asw_image* r, asw_image* o) {
        return asw_stereo_match(c, l, r, o, (int)disparityType, (int)algorithmType, winSize, minDisparity, numDisparity, nullptr, 0);
    }, "stereoMatching");
    disparityMap = d;  // `disparityMap = computeAdaptiveWeight...(...)`, M.cpp:58-82 (empty Mat on silent errors)
}

// M.h:101-102
inline void computeAD(AswMat leftImg, AswMat rightImg, std::vector<AswMat>& cost_ds, DisparityType dispType = DISPARITY_LEFT,
                      int minDisparity = 0, int numDisparity = 30)
{
    asw::detail::cost_volume<uint8_t>(leftImg, rightImg, cost_ds, numDisparity, ASW_8U, 0, [&](asw_ctx* c, asw_image* l, asw_image* r, uint8_t* v) {
        return asw_cost_ad(c, l, r, v, (int)dispType, minDisparity, numDisparity);
    }, "computeAD");
}

// M.h:105-106
inline void computeTAD(AswMat leftImg, AswMat rightImg, std::vector<AswMat>& cost_ds, DisparityType dispType = DISPARITY_LEFT,
                       int threshold_T = 30, int minDisparity = 0, int numDisparity = 30)
{
    asw::detail::cost_volume<uint8_t>(leftImg, rightImg, cost_ds, numDisparity, ASW_8U, 0, [&](asw_ctx* c, asw_image* l, asw_image* r, uint8_t* v) {
        return asw_cost_tad(c, l, r, v, (int)dispType, threshold_T, minDisparity, numDisparity);
    }, "computeTAD");
}

// M.h:117-118
inline void computeSD(AswMat leftImg, AswMat rightImg, std::vector<AswMat>& cost_ds, DisparityType dispType = DISPARITY_LEFT,
                      int minDisparity = 0, int numDisparity = 30)
{
    asw::detail::cost_volume<uint8_t>(leftImg, rightImg, cost_ds, numDisparity, ASW_8U, 0, [&](asw_ctx* c, asw_image* l, asw_image* r, uint8_t* v) {
        return asw_cost_sd(c, l, r, v, (int)dispType, minDisparity, numDisparity);
    }, "computeSD");
}

// M.h:109-111
inline void computeSimilarity(AswMat leftImg, AswMat rightImg, std::vector<AswMat>& cost_d_imgs, double regularity, double thresC,
                              double thresG, DisparityType dispType, int minDisparity, int numDisparity)
{
    asw::detail::cost_volume<float>(leftImg, rightImg, cost_d_imgs, numDisparity, ASW_32F, 0, [&](asw_ctx* c, asw_image* l, asw_image* r, float* v) {
        return asw_cost_similarity(c, l, r, v, regularity, thresC, thresG, (int)dispType, 0, minDisparity, numDisparity);
    }, "computeSimilarity");
}

// M.h:112-114 (padded overload)
inline void computeSimilarity(AswMat leftImg, AswMat rightImg, std::vector<AswMat>& cost_d_imgs, double regularity, double thresC,
                              double thresG, DisparityType dispType, int winSize, int minDisparity, int numDisparity)
{
    if (winSize % 2 == 0) return;  // M.cpp:654-657: returns before touching cost_d_imgs
    asw::detail::cost_volume<float>(leftImg, rightImg, cost_d_imgs, numDisparity, ASW_32F, winSize / 2, [&](asw_ctx* c, asw_image* l, asw_image* r, float* v) {
        return asw_cost_similarity(c, l, r, v, regularity, thresC, thresG, (int)dispType, winSize, minDisparity, numDisparity);
    }, "computeSimilarity(padded)");
}

// M.h:122-123: computeNCC -> disparity
inline AswMat computeNCC(AswMat leftImg, AswMat rightImg, DisparityType dispType = DISPARITY_LEFT, int winSize = 7,
                         int minDisparity = 0, int numDisparity = 30)
{
    return asw::detail::aggregate(leftImg, rightImg, [&](asw_ctx* c, asw_image* l, asw_image* r, asw_image* o) {
        return asw_ncc_disparity(c, l, r, o, (int)dispType, winSize, minDisparity, numDisparity);
    }, "computeNCC");
}

// M.h:124-126: computeNCC -> min-max normalised cost planes
inline void computeNCC(AswMat leftImg, AswMat rightImg, std::vector<AswMat>& cost_ds, DisparityType dispType = DISPARITY_LEFT,
                       int winSize = 7, int minDisparity = 0, int numDisparity = 30)
{
    asw::detail::cost_volume<float>(leftImg, rightImg, cost_ds, numDisparity, ASW_32F, 0, [&](asw_ctx* c, asw_image* l, asw_image* r, float* v) {
        return asw_cost_ncc(c, l, r, v, (int)dispType, winSize, minDisparity, numDisparity, 1);
    }, "computeNCC(costs)");
}

// M.h:133-134
inline AswMat computeAdaptiveWeight(AswMat leftImg, AswMat rightImg, double gamma_c = 30, double gamma_g = 2,
                                    DisparityType dispType = DISPARITY_LEFT, int winSize = 7, int minDisparity = 186,
                                    int numDisparity = 144)
{
    return asw::detail::aggregate(leftImg, rightImg, [&](asw_ctx* c, asw_image* l, asw_image* r, asw_image* o) {
        return asw_aggregate_bilateral(c, l, r, o, gamma_c, gamma_g, (int)dispType, winSize, minDisparity, numDisparity, nullptr, 0);
    }, "computeAdaptiveWeight");
}

// M.h:135-136 (DISPARITY_RIGHT: undefined behaviour in the reference, M.cpp:1291-1295 -> empty Mat here)
inline AswMat computeAdaptiveWeight_direct8(AswMat leftImg, AswMat rightImg, DisparityType dispType = DISPARITY_LEFT, int winSize = 7,
                                            int minDisparity = 186, int numDisparity = 144)
{
    return asw::detail::aggregate(leftImg, rightImg, [&](asw_ctx* c, asw_image* l, asw_image* r, asw_image* o) {
        return asw_aggregate_direct8(c, l, r, o, (int)dispType, winSize, minDisparity, numDisparity, nullptr, 0);
    }, "computeAdaptiveWeight_direct8");
}

// M.h:142-143
inline AswMat computeAdaptiveWeight_geodesic(AswMat leftImg, AswMat rightImg, DisparityType dispType = DISPARITY_LEFT, int winSize = 7,
                                             int minDisparity = 186, int numDisparity = 144)
{
    return asw::detail::aggregate(leftImg, rightImg, [&](asw_ctx* c, asw_image* l, asw_image* r, asw_image* o) {
        return asw_aggregate_geodesic(c, l, r, o, (int)dispType, winSize, minDisparity, numDisparity, nullptr, 0);
    }, "computeAdaptiveWeight_geodesic");
}

// M.h:174-176
inline AswMat computeAdaptiveWeight_GuidedF_3(AswMat leftImg, AswMat rightImg, DisparityType dispType = DISPARITY_LEFT, double eps = 1e-6,
                                              int winSize = 35, int minDisparity = 186, int numDisparity = 144)
{
    return asw::detail::aggregate(leftImg, rightImg, [&](asw_ctx* c, asw_image* l, asw_image* r, asw_image* o) {
        return asw_aggregate_guided3(c, l, r, o, (int)dispType, eps, winSize, minDisparity, numDisparity, nullptr, 0);
    }, "computeAdaptiveWeight_GuidedF_3");
}

// M.h:155-157
inline AswMat computeAdaptiveWeight_bilateralGrid(AswMat leftImg, AswMat rightImg, DisparityType dispType = DISPARITY_LEFT,
                                                  double sampleRateS = 10, double sampleRateR = 10, int minDisparity = 186,
                                                  int numDisparity = 144)
{
    return asw::detail::aggregate(leftImg, rightImg, [&](asw_ctx* c, asw_image* l, asw_image* r, asw_image* o) {
        return asw_aggregate_bilgrid(c, l, r, o, (int)dispType, sampleRateS, sampleRateR, minDisparity, numDisparity, nullptr, 0);
    }, "computeAdaptiveWeight_bilateralGrid");
}

// M.h:157-159
inline AswMat computeAdaptiveWeight_BLO1(AswMat leftImg, AswMat rightImg, DisparityType dispType = DISPARITY_LEFT, double sampleRateR = 10,
                                         int winSize = 35, int minDisparity = 186, int numDisparity = 144)
{
    return asw::detail::aggregate(leftImg, rightImg, [&](asw_ctx* c, asw_image* l, asw_image* r, asw_image* o) {
        return asw_aggregate_blo1(c, l, r, o, (int)dispType, sampleRateR, winSize, minDisparity, numDisparity, nullptr, 0);
    }, "computeAdaptiveWeight_BLO1");
}

// Not in the reference: the cross-check that consumes a DISPARITY_LEFT and a DISPARITY_RIGHT map (asw_lr_check)
inline AswMat leftRightCheck(AswMat dispLeft, AswMat dispRight, float maxDiff = 1.0f, float invalidValue = -1.0f, int* nInvalid = nullptr)
{
    if (dispLeft.rows != dispRight.rows || dispLeft.cols != dispRight.cols) return AswMat();
    AswMat out = asw::detail::make(dispLeft.rows, dispLeft.cols, ASW_32F, 1);
    asw_image a = asw::detail::view(dispLeft), b = asw::detail::view(dispRight), o = asw::detail::view(out);
    if (a.depth != ASW_32F || b.depth != ASW_32F || a.step != (size_t)a.cols * 4 || b.step != (size_t)b.cols * 4 || o.step != (size_t)o.cols * 4)
        throw std::runtime_error("leftRightCheck: continuous CV_32FC1 maps expected");
    int rc = asw_lr_check(asw::detail::context(), (const float*)a.data, (const float*)b.data, a.rows, a.cols, maxDiff, invalidValue,
                          (float*)o.data, nInvalid);
    asw::detail::raise_unless_ok(rc, "leftRightCheck");
    return rc == ASW_OK ? out : AswMat();
}

// M.h:156 / M.cpp:2442-2503: ONE disparity; the view that is not the reference view arrives bordered by the caller
// (copyMakeBorder by max_offset, M.cpp:2877-2878).  Empty Mat for an even window or a bordered view that is not wider.
inline AswMat getCostSAD_d(AswMat leftImg, AswMat rightImg, int disparity, DisparityType dispType = DISPARITY_LEFT, int winSize = 35)
{
    if (leftImg.empty() || rightImg.empty()) return AswMat();
    const AswMat& ref = dispType == DISPARITY_LEFT ? leftImg : rightImg;
    AswMat cost = asw::detail::make(ref.rows, ref.cols, ASW_32F, 1);
    asw_image li = asw::detail::view(leftImg), ri = asw::detail::view(rightImg), ci = asw::detail::view(cost);
    if (ci.step != (size_t)ci.cols * 4) throw std::runtime_error("getCostSAD_d: continuous output expected");
    int rc = asw_cost_sad_d(asw::detail::context(), &li, &ri, (float*)ci.data, disparity, (int)dispType, winSize);
    asw::detail::raise_unless_ok(rc, "getCostSAD_d");
    return rc == ASW_OK ? cost : AswMat();
}

// M.h:5-19: strict weak order of the map below (x first, then y)
#ifndef ASW_REFERENCE_COMPARATORS_DEFINED
#define ASW_REFERENCE_COMPARATORS_DEFINED
#ifdef ASW_WITH_OPENCV
typedef cv::Point AswPoint;
#else
typedef asw::Point AswPoint;
#endif
struct MY_COMP_Point2i {
    bool operator()(const AswPoint& left, const AswPoint& right) const
    {
        if (left.x < right.x) return true;
        if (left.x == right.x && left.y < right.y) return true;
        return false;
    }
};
#endif

// M.h:141 / M.cpp:1392-1424: weightGeoDist[Point(x, y)] = winSize x winSize CV_32FC1 window of geodesic distances of pixel
// (x, y); the map is cleared first; an even window returns without touching it (M.cpp:1394-1397).
inline void getGeodesicDist(AswMat originImg, std::map<AswPoint, AswMat, MY_COMP_Point2i>& weightGeoDist, int winSize = 15,
                            int iterTime = 3)
{
    if (winSize % 2 == 0 || originImg.empty()) return;
    const int H = originImg.rows, W = originImg.cols;
    const size_t cells = (size_t)winSize * winSize;
    std::vector<float> dense((size_t)H * W * cells);  // [y][x][win][win], asw_geodesic_dist's layout
    asw_image ii = asw::detail::view(originImg);
    int rc = asw_geodesic_dist(asw::detail::context(), &ii, dense.data(), winSize, iterTime);
    asw::detail::raise_unless_ok(rc, "getGeodesicDist");
    if (rc != ASW_OK) return;
    if (!weightGeoDist.empty()) weightGeoDist.clear();  // M.cpp:1406-1409
    for (int x = 0; x < W; x++)                         // insertion order of the reference: x outer, y inner (M.cpp:1410-1412)
        for (int y = 0; y < H; y++) {
            AswMat w = asw::detail::make(winSize, winSize, ASW_32F, 1);
            asw_image wi = asw::detail::view(w);
            for (int r = 0; r < winSize; r++)
                std::memcpy((uint8_t*)wi.data + (size_t)r * wi.step, dense.data() + ((size_t)y * W + x) * cells + (size_t)r * winSize,
                            (size_t)winSize * 4);
            weightGeoDist[AswPoint(x, y)] = w;
        }
}

// M.h:165
inline AswMat getGuidedFilter(AswMat guidedImg, AswMat inputP, int r, double eps)
{
    if (guidedImg.rows != inputP.rows || guidedImg.cols != inputP.cols) return AswMat();  // M.cpp:2768-2769
    AswMat q = asw::detail::make(inputP.rows, inputP.cols, ASW_32F, 1);
    asw_image gi = asw::detail::view(guidedImg), pi = asw::detail::view(inputP), qi = asw::detail::view(q);
    if (pi.depth != ASW_32F || pi.step != (size_t)pi.cols * 4 || qi.step != (size_t)qi.cols * 4)
        throw std::runtime_error("getGuidedFilter: inputP must be a continuous CV_32FC1 Mat");
    int rc = asw_guided_filter(asw::detail::context(), &gi, (const float*)pi.data, (float*)qi.data, r, eps);
    asw::detail::raise_unless_ok(rc, "getGuidedFilter");
    return rc == ASW_OK ? q : AswMat();
}

// M.h:166-168
inline AswMat computeAdaptiveWeight_GuidedF(AswMat leftImg, AswMat rightImg, DisparityType dispType = DISPARITY_LEFT, double eps = 1e-8,
                                            int winSize = 35, int minDisparity = 186, int numDisparity = 144)
{
    return asw::detail::aggregate(leftImg, rightImg, [&](asw_ctx* c, asw_image* l, asw_image* r, asw_image* o) {
        return asw_aggregate_guided(c, l, r, o, (int)dispType, eps, winSize, minDisparity, numDisparity, nullptr, 0);
    }, "computeAdaptiveWeight_GuidedF");
}

// M.h:169-171
inline AswMat computeAdaptiveWeight_GuidedF_2(AswMat leftImg, AswMat rightImg, DisparityType dispType = DISPARITY_LEFT, double eps = 1e-8,
                                              int winSize = 35, int minDisparity = 186, int numDisparity = 144)
{
    return asw::detail::aggregate(leftImg, rightImg, [&](asw_ctx* c, asw_image* l, asw_image* r, asw_image* o) {
        return asw_aggregate_guided2(c, l, r, o, (int)dispType, eps, winSize, minDisparity, numDisparity, nullptr, 0);
    }, "computeAdaptiveWeight_GuidedF_2");
}

// M.h:179-182
inline AswMat computeAdaptiveWeight_WeightedMedian(AswMat leftImg, AswMat rightImg, DisparityType dispType = DISPARITY_LEFT,
                                                   int winSize = 35, double sampleRateS = 10, double sampleRateR = 10,
                                                   int minDisparity = 186, int numDisparity = 144)
{
    return asw::detail::aggregate(leftImg, rightImg, [&](asw_ctx* c, asw_image* l, asw_image* r, asw_image* o) {
        return asw_aggregate_wmedian(c, l, r, o, (int)dispType, winSize, sampleRateS, sampleRateR, minDisparity, numDisparity, nullptr, 0);
    }, "computeAdaptiveWeight_WeightedMedian");
}
