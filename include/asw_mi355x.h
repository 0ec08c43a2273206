/*
 * asw_mi355x.h -- C-ABI of the MI355X (gfx950) adaptive-support-weight stereo matcher.
 *
 * Drop-in boundary for the hot path of ZhangYY12345/aswStereoMatch: every entry point below
 * replaces one function of aswStereoMatch/methods/aswMethods.{h,cpp} ("M.h"/"M.cpp"); the
 * reference interface it stands in for is cited next to it.  Plain pointers and sizes only;
 * no C++ / torch / OpenCV types cross this boundary.  INTEGRATION.md shows the cv::Mat shim a
 * maintainer of the reference would add on top (and the ctypes binding the tests use).
 *
 * Conventions
 *  - images are what cv::Mat holds: row-major, interleaved channels, `step` bytes per row;
 *  - cost volumes are d-major [d][y][x] and dense, plane k <-> disparity min_d + k, the same
 *    order as the reference's std::vector<cv::Mat>;
 *  - enum values equal the reference's (parametersStereo.h:4-24);
 *  - all host pointers are caller-owned; device scratch lives in the context and is reused;
 *  - every call is synchronous on return and returns an asw_status (never throws).
 */
#ifndef ASW_MI355X_H
#define ASW_MI355X_H

#include <stddef.h>
#include <stdint.h>

/* the library is built with -fvisibility=hidden: only the entry points below are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (the reference has none: it returns silently / an empty Mat / throws) ---- */
typedef enum asw_status {
    ASW_OK = 0,
    ASW_ERR_SIZE_MISMATCH = 1,      /* M.cpp:217-220, 313-316, 430-433: silent return          */
    ASW_ERR_EVEN_WINDOW = 2,        /* M.cpp:654-657, 1440-1443, 2458-2462, 3238-3241: Mat()  */
    ASW_ERR_UNSUPPORTED_METHOD = 3, /* enum values 0 and 1 (BM, SGBM: OpenCV's own matchers) */
    ASW_ERR_UNSUPPORTED_LAYOUT = 4, /* where the reference throws cv::Exception (SURVEY B-7)  */
    ASW_ERR_HIP = 5,                /* a HIP runtime call or kernel launch failed             */
    ASW_ERR_ALLOC = 6,
    ASW_ERR_BAD_ARGUMENT = 7,       /* null pointer, non-positive size, bad depth             */
    ASW_ERR_NO_FRAME = 8            /* resident API used before asw_upload_pair               */
} asw_status;

/* parametersStereo.h:4-8 */
enum { ASW_DISPARITY_LEFT = 0, ASW_DISPARITY_RIGHT = 1 };

/* parametersStereo.h:10-24 (StereoMatchingAlgorithms) */
enum {
    ASW_ALG_BM = 0,
    ASW_ALG_SGBM = 1,
    ASW_ALG_ADAPTIVE_WEIGHT = 2,
    ASW_ALG_ADAPTIVE_WEIGHT_8DIRECT = 3,
    ASW_ALG_ADAPTIVE_WEIGHT_GEODESIC = 4,
    ASW_ALG_ADAPTIVE_WEIGHT_BILATERAL_GRID = 5,
    ASW_ALG_ADAPTIVE_WEIGHT_BLO1 = 6,
    ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER = 7,
    ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_2 = 8,
    ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_3 = 9,
    ASW_ALG_ADAPTIVE_WEIGHT_MEDIAN = 10,
    ASW_ALG_NCC = 11
};

/* cv::Mat depth codes */
enum { ASW_8U = 0, ASW_32F = 5 };

/* The part of a cv::Mat header the path needs (M.h:91: cv::Mat srcLeft, srcRight, disparityMap) */
typedef struct asw_image {
    void* data;   /* host pointer                               */
    int rows;     /* cv::Mat::rows                              */
    int cols;     /* cv::Mat::cols                              */
    int channels; /* cv::Mat::channels()                        */
    int depth;    /* ASW_8U or ASW_32F                          */
    size_t step;  /* bytes per row (cv::Mat::step), >= cols*channels*elemsize */
} asw_image;

typedef struct asw_ctx asw_ctx; /* one per device; not shared between threads */

/* Per-call timing of the kernels of the last asw_match_resident / asw_stereo_match call,
 * measured with HIP events on the context's stream. */
typedef struct asw_timing {
    float total_ms;      /* all kernels of the call                                  */
    float aggregate_ms;  /* the dominant aggregation kernel(s) (ASW / guided / median) */
    float cost_ms;       /* cost-build kernels (gray, Scharr, min/max, ...)           */
    int aggregate_launches;
} asw_timing;

/* ---- context ---- */
int asw_create(int device_id, asw_ctx** out);
void asw_destroy(asw_ctx* ctx);
const char* asw_status_string(int status);
int asw_device_count(void);
/* cvtColor(COLOR_BGR2GRAY) on 8U is fixed-point in OpenCV and its constants changed between releases: 14 bits
 * {B 1868, G 9617, R 4899} in 4.1.0 -- the version the reference pins (aswStereoMatch.vcxproj:67,71), the default here --
 * and 15 bits {3735, 19235, 9798} in later 4.x releases; the two differ by +-1 on a few percent of the pixels.  A maintainer
 * who links the reference against a newer OpenCV selects 15 to keep bit-identical gray planes (classic, direct8, SAD, BLO1,
 * bilateral grid, NCC all start from them; M.cpp:1031-1033, 2448-2454, 835-840).  bits: 14 or 15. */
int asw_set_gray_bits(asw_ctx* ctx, int bits);

/* ---- whole-method entry point: stereoMatching(), M.h:91-92, M.cpp:46-88 ----
 * disp: ASW_32F, 1 channel, rows x cols, caller-allocated; receives ABSOLUTE disparity
 * (min_d + index), like the reference's CV_32FC1 result.  Per-method literals are the
 * selector's (gamma_c=30, gamma_g=20; eps=1e-6; rateS=rateR=10).
 * cost_volume_out (optional, may be NULL): aggregated cost volume, [n][rows][cols] f32 with
 * n = asw_volume_planes(algorithm, num_d): num_d, or num_d + 1 for ADAPTIVE_WEIGHT, 8DIRECT, GEODESIC and BILATERAL_GRID, whose
 * candidate range is inclusive (M.cpp:1021,1074; 1171; 1447,1467; 2256,2280).
 * cost_volume_floats: capacity of cost_volume_out in floats (ignored when it is NULL); a buffer shorter than
 * n * rows * cols is refused with ASW_ERR_BAD_ARGUMENT before anything is computed or written.  The same pair of
 * arguments ends every asw_aggregate_* entry point below.
 * These host-buffer entry points work on a private frame of the context: they never disturb the resident slots of
 * asw_upload_pair / asw_match_resident. */
int asw_stereo_match(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                     int disparity_type, int algorithm, int win_size, int min_disparity, int num_disparity,
                     float* cost_volume_out, size_t cost_volume_floats);

/* Planes of the cost volume `algorithm` produces for num_disparity candidates (what cost_volume_out must hold);
 * 0 for an algorithm the library does not serve. */
int asw_volume_planes(int algorithm, int num_disparity);

/* ---- the same, split so that inputs can stay resident in HBM (bench / pipelines) ----
 * Slots are caller-numbered (0..4095).  asw_upload_pair / asw_preprocess_pair replace a slot's pair and drop its previous
 * disparity and volume; asw_download_* return ASW_ERR_NO_FRAME until asw_match_resident has succeeded on the CURRENT
 * pair of the slot (a failed match drops the results too). */
int asw_upload_pair(asw_ctx* ctx, int slot, const asw_image* left, const asw_image* right);
int asw_match_resident(asw_ctx* ctx, int slot, int disparity_type, int algorithm, int win_size,
                       int min_disparity, int num_disparity, int keep_volume);
int asw_download_disparity(asw_ctx* ctx, int slot, asw_image* disp);
int asw_download_volume(asw_ctx* ctx, int slot, float* cost_volume_out, size_t n_floats);
int asw_synchronize(asw_ctx* ctx);
int asw_get_timing(asw_ctx* ctx, asw_timing* out);

/* ---- driver-side pre/post-processing on the device (aswStereoMatch.cpp, "main.cpp") ----
 * asw_preprocess_pair: main.cpp:30-31 resize(img, Size(out_width, out_height)) (INTER_LINEAR) and, if detail_boost != 0,
 * main.cpp:67-89 (BGR2HSV, V += 2*(V - bilateralFilter(V, 7, 10, 3, BORDER_REFLECT)), HSV2BGR) for both 8UC3 images; the
 * result becomes the resident pair of `slot` (as after asw_upload_pair).  asw_download_pair fetches it.
 * asw_download_disparity_u8: main.cpp:97-98 disparityMap.convertTo(CV_8UC1) and, if normalize != 0,
 * normalize(.., 0, 255, NORM_MINMAX) of the last asw_match_resident result of `slot`; disp_u8: ASW_8U, 1 channel. */
int asw_preprocess_pair(asw_ctx* ctx, int slot, const asw_image* left_full, const asw_image* right_full, int out_width,
                        int out_height, int detail_boost);
int asw_download_pair(asw_ctx* ctx, int slot, asw_image* left, asw_image* right);
int asw_download_disparity_u8(asw_ctx* ctx, int slot, asw_image* disp_u8, int normalize);

/* ---- per-method entry points with explicit parameters (M.h:133-184) ---- */
/* computeAdaptiveWeight, M.h:133-134, M.cpp:1016-1156 */
int asw_aggregate_bilateral(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                            double gamma_c, double gamma_g, int disparity_type, int win_size,
                            int min_disparity, int num_disparity, float* cost_volume_out, size_t cost_volume_floats);
/* computeAdaptiveWeight_direct8, M.h:135-136, M.cpp:1167-1319: the classic scheme on row + column + main diagonal of the
 * window, gamma_c = 30, gamma_g = win*2/3 (integer division).  DISPARITY_LEFT only: the reference's RIGHT branch indexes
 * its weight vectors with a negative tap coordinate (M.cpp:1291-1295) -> ASW_ERR_UNSUPPORTED_LAYOUT. */
int asw_aggregate_direct8(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                          int disparity_type, int win_size, int min_disparity, int num_disparity,
                          float* cost_volume_out, size_t cost_volume_floats);
/* computeAdaptiveWeight_geodesic, M.h:142-143, M.cpp:1436-1534 */
int asw_aggregate_geodesic(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                           int disparity_type, int win_size, int min_disparity, int num_disparity,
                           float* cost_volume_out, size_t cost_volume_floats);
/* computeAdaptiveWeight_GuidedF, M.h:166-168, M.cpp:2867-2963 */
int asw_aggregate_guided(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                         int disparity_type, double eps, int win_size, int min_disparity, int num_disparity,
                         float* cost_volume_out, size_t cost_volume_floats);
/* computeAdaptiveWeight_GuidedF_2, M.h:169-171, M.cpp:2976-3050 */
int asw_aggregate_guided2(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                          int disparity_type, double eps, int win_size, int min_disparity, int num_disparity,
                          float* cost_volume_out, size_t cost_volume_floats);
/* computeAdaptiveWeight_GuidedF_3, M.h:172-174, M.cpp:3063-3137: normalised NCC planes (computeNCC, M.cpp:924-1013) filtered
 * with the 6-channel guide [L, R shifted by d] (LEFT) or with the plain right image (RIGHT, M.cpp:3110) */
int asw_aggregate_guided3(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                          int disparity_type, double eps, int win_size, int min_disparity, int num_disparity,
                          float* cost_volume_out, size_t cost_volume_floats);
/* computeAdaptiveWeight_BLO1, M.h:157-159, M.cpp:2505-2725 (min_disparity must be 0: the reference indexes its
 * per-key slices with the absolute offset) */
int asw_aggregate_blo1(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                       int disparity_type, double sample_rate_r, int win_size, int min_disparity, int num_disparity,
                       float* cost_volume_out, size_t cost_volume_floats);
/* computeAdaptiveWeight_bilateralGrid, M.h:155-157, M.cpp:2253-2430 (grid: createBilGrid M.cpp:1831-2185, enum 5 calls it with
 * rates 10, 10): offsets min_d .. min_d+num_d inclusive, cost volume num_d+1 planes (NaN / inf where the interpolated count is 0).
 * DISPARITY_LEFT only -- the reference's RIGHT branch reads one column past the row (M.cpp:1929, 2356).
 * sample_rate_r >= 2.55 (at most 101 bins per range axis). */
int asw_aggregate_bilgrid(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                          int disparity_type, double sample_rate_s, double sample_rate_r, int min_disparity,
                          int num_disparity, float* cost_volume_out, size_t cost_volume_floats);
/* computeAdaptiveWeight_WeightedMedian, M.h:179-182, M.cpp:3228-3383 */
int asw_aggregate_wmedian(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                          int disparity_type, int win_size, double rate_s, double rate_r, int min_disparity,
                          int num_disparity, float* cost_volume_out, size_t cost_volume_floats);

/* ---- cost builders (M.h:101-113, 156) : outputs are dense d-major volumes ---- */
/* computeAD, M.cpp:208-292: cost u8 [num_d][rows][cols]; 1- or 3-channel 8U input */
int asw_cost_ad(asw_ctx* ctx, const asw_image* left, const asw_image* right, uint8_t* cost,
                int disparity_type, int min_disparity, int num_disparity);
/* computeTAD, M.cpp:304-401: 0/255 mask of AD > threshold_T */
int asw_cost_tad(asw_ctx* ctx, const asw_image* left, const asw_image* right, uint8_t* cost,
                 int disparity_type, int threshold_t, int min_disparity, int num_disparity);
/* computeSD, M.h:117-118, M.cpp:670-759: the AD value squared by a u8 Mat::mul, i.e. min(255, ad*ad) */
int asw_cost_sd(asw_ctx* ctx, const asw_image* left, const asw_image* right, uint8_t* cost,
                int disparity_type, int min_disparity, int num_disparity);
/* computeSimilarity (TAD C+G), M.cpp:415-636; win_size = 0 selects the unpadded overload,
 * win_size > 0 the padded one (M.cpp:651-668): planes are (rows+2h) x (cols+2h). */
int asw_cost_similarity(asw_ctx* ctx, const asw_image* left, const asw_image* right, float* cost,
                        double regularity, double thres_c, double thres_g, int disparity_type,
                        int win_size, int min_disparity, int num_disparity);
/* getCostSAD_d for every d as called from M.cpp:2884-2898: box mean of gray abs-diff */
int asw_cost_sad(asw_ctx* ctx, const asw_image* left, const asw_image* right, float* cost,
                 int disparity_type, int win_size, int min_disparity, int num_disparity);

/* getCostSAD_d itself, M.h:156, M.cpp:2442-2503: ONE disparity, and the image that is not the reference view arrives
 * already bordered by the caller (wider by the caller's max_offset; M.cpp:2877-2878, 2884-2898): DISPARITY_LEFT reads
 * right(Rect(right.cols - left.cols - disparity, 0, left.cols, rows)), DISPARITY_RIGHT reads left(Rect(disparity, 0,
 * right.cols, rows)).  1- or 3-channel 8U inputs (3: BGR2GRAY first).  cost: f32 rows x cols of the reference view.
 * ASW_ERR_SIZE_MISMATCH where the reference returns Mat() (bordered image not wider, M.cpp:2473-2476, 2488-2491),
 * ASW_ERR_BAD_ARGUMENT for a ROI outside the bordered image (cv::Exception in the reference). */
int asw_cost_sad_d(asw_ctx* ctx, const asw_image* left, const asw_image* right, float* cost, int disparity,
                   int disparity_type, int win_size);

/* computeNCC, volume overload, M.h:121-122, M.cpp:924-1013: cost = sum(l*r) / (sum(l*l)*sum(r*r)) on mean-removed windows of
 * the RGB2GRAY images; normalized != 0: every plane min-max normalised as the reference stores it, 0: the raw planes.
 * 1- or 3-channel 8U input. */
int asw_cost_ncc(asw_ctx* ctx, const asw_image* left, const asw_image* right, float* cost, int disparity_type,
                 int win_size, int min_disparity, int num_disparity, int normalized);
/* computeNCC, disparity overload, M.h:119-120, M.cpp:812-913 (enum NCC = 11): offsets min_d .. min_d+num_d-2, smallest cost
 * wins (LEFT); DISPARITY_RIGHT never writes a pixel in the reference -> all zeros here. */
int asw_ncc_disparity(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp, int disparity_type,
                      int win_size, int min_disparity, int num_disparity);

/* ---- building blocks that are public in the reference header ---- */
/* getGuidedFilter, M.h:165, M.cpp:2766-2854: guide 8U with 3 or 6 channels, p/q f32 rows x cols */
int asw_guided_filter(asw_ctx* ctx, const asw_image* guide, const float* p, float* q, int r, double eps);
/* getGeodesicDist, M.h:141, M.cpp:1392-1424: out f32 [rows][cols][win][win] */
int asw_geodesic_dist(asw_ctx* ctx, const asw_image* img, float* out, int win_size, int iter_time);
/* inline WTA of M.cpp:1144-1150 / 3032-3048 on a dense f32 volume [n][rows][cols] */
int asw_wta(asw_ctx* ctx, const float* cost_volume, int n, int rows, int cols, int min_disparity,
            float* disp);
/* Left-right consistency check, the consumer of the DISPARITY_RIGHT maps (M.cpp:1113-1142, 1498-1520, 2919-2935 produce them;
 * the reference itself never cross-checks, so the rule is this library's): out(y,x) = dl(y,x) if xr = x - (int)dl(y,x) lies in
 * [0, cols) and |dl(y,x) - dr(y,xr)| <= max_diff, else invalid_value.  n_invalid (optional) receives the number of rejected pixels. */
int asw_lr_check(asw_ctx* ctx, const float* disp_left, const float* disp_right, int rows, int cols, float max_diff,
                 float invalid_value, float* out, int* n_invalid);
/* cvtColor(COLOR_BGR2GRAY) as used at M.cpp:1031-1033 */
int asw_bgr2gray(asw_ctx* ctx, const asw_image* bgr, uint8_t* gray);

/* ---- batch over frames and devices (frames are independent; no collective) ----
 * Frame i goes to device device_ids[i % n_devices]; one host thread + one context per device. */
int asw_stereo_match_batch(int n_frames, const asw_image* lefts, const asw_image* rights, asw_image* disps,
                           int disparity_type, int algorithm, int win_size, int min_disparity,
                           int num_disparity, int n_devices, const int* device_ids);

#ifdef __cplusplus
}
#endif

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif /* ASW_MI355X_H */
