/*
 * asw_oracle.c -- CPU restatement of the aswStereoMatch hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X implementation in aswstereomatch_amd/.  It is
 * imported / linked only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * The product path never calls it.
 *
 * PARITY UNPINNED: the reference (ZhangYY12345/aswStereoMatch) ships no tests, fixtures or
 * golden images, and its arithmetic primitives live in OpenCV 4.1.0, which is neither vendored
 * in /root/reference nor installed here, so the reference cannot be compiled.  Every OpenCV
 * primitive is restated below from its published portable C++ algorithm;
 * the reference's own loops are followed line by line (citations are to
 * aswStereoMatch/methods/aswMethods.cpp, "M.cpp").  Hand-derivable known-answer tests
 *  pin what can be pinned without OpenCV.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: the reference is built by MSVC for x64/SSE2, which never fuses
 * a*b+c, and several results (M.cpp:1104-1108, 1488-1492, 22-31) depend on that.
 *
 * Layout conventions: images are row-major, interleaved channels (cv::Mat continuous layout);
 * cost volumes are d-major [d][y][x], index 0 <-> disparity minDisparity, like the
 * reference's std::vector<cv::Mat>.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_ERR_SIZE_MISMATCH 1  /* reference: silent return (M.cpp:217-220, 430-433) */
#define ORC_ERR_EVEN_WINDOW 2    /* reference: return Mat() (M.cpp:654-657, 1440-1443, 2458-2462) */
#define ORC_ERR_UNSUPPORTED_LAYOUT 4 /* reference: cv::Exception */
#define ORC_ERR_ALLOC 6

#define DISPARITY_LEFT 0  /* parametersStereo.h:4-8 */
#define DISPARITY_RIGHT 1

static int g_threads = 1;

void orc_set_threads(int n)
{
    g_threads = n < 1 ? 1 : n;
#ifdef _OPENMP
    omp_set_num_threads(g_threads);
#endif
}

int orc_get_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------------------------------
 * OpenCV primitive restatements
 * ------------------------------------------------------------------------------------- */

/* cv::borderInterpolate, BORDER_REFLECT (fedcba|abcdefgh|hgfedcb), repeated if far out. A-2 */
static inline int reflect_idx(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p - 1;
        else p = len - 1 - (p - len);
    }
    return p;
}

/* BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba): default of boxFilter / filter2D. A-3 */
static inline int reflect101_idx(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p;
        else p = len - 1 - (p - len) - 1;
    }
    return p;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* cvtColor(COLOR_BGR2GRAY) on 8U: fixed-point BT.601 (A-1).  M.cpp:1031,1033,2448,2454.
 * OpenCV 4.1.0 (the reference's pin) is believed to use the 14-bit constants; later 4.x releases use the 15-bit set.  The
 * choice cannot be verified offline, so it is ONE switch here (and one in the library, asw_set_gray_bits): 14 by default. */
static int g_gray_bits = 14;
void orc_set_gray_bits(int bits) { g_gray_bits = (bits == 15) ? 15 : 14; }
void orc_bgr2gray(const uint8_t* bgr, int H, int W, uint8_t* gray)
{
    const int wide = g_gray_bits == 15;
    const int B2Y = wide ? 3735 : 1868, G2Y = wide ? 19235 : 9617, R2Y = wide ? 9798 : 4899, shift = wide ? 15 : 14;
    for (long i = 0; i < (long)H * W; i++) {
        int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * B2Y + g * G2Y + r * R2Y + (1 << (shift - 1))) >> shift);
    }
}

/* u8 MatExpr (c0 + c1 + c2) / 3  ==  addWeighted(saturate(c0+c1), 1/3, c2, 1/3, 0)  (A-4).
 * OpenCV's 8u addWeighted works in float and rounds half-to-even (cvRound). */
static inline uint8_t mean3_u8(int c0, int c1, int c2)
{
    int t = c0 + c1;
    if (t > 255) t = 255;
    const float a = (float)(1.0 / 3.0);
    float v = (float)t * a + (float)c2 * a;
    long r = lrintf(v); /* default rounding mode = to nearest even, like cvRound */
    if (r > 255) r = 255;
    if (r < 0) r = 0;
    return (uint8_t)r;
}

static inline int absdiff_u8(int a, int b) { return a > b ? a - b : b - a; }

/* ---------------------------------------------------------------------------------------
 * Cost builders
 * ------------------------------------------------------------------------------------- */

/* computeAD, M.cpp:208-292.  C = 1 or 3 channels.  cost: numD planes of H x W u8. */
int orc_compute_ad(const uint8_t* L, const uint8_t* R, int H, int W, int C, int disp_type,
                   int minD, int numD, uint8_t* cost)
{
    if (C != 1 && C != 3) return ORC_ERR_UNSUPPORTED_LAYOUT; /* reference: no branch taken */
    const int min_off = minD; /* M.cpp:214-215 */
    for (int k = 0; k < numD; k++) {
        const int off = min_off + k;
        uint8_t* out = cost + (size_t)k * H * W;
        for (int y = 0; y < H; y++) {
            for (int x = 0; x < W; x++) {
                const uint8_t *a, *b;
                if (disp_type == DISPARITY_LEFT) {
                    /* right_border(Rect(max_offset - offset,...)) with a REFLECT left pad of
                     * max_offset columns: column x reads R[reflect(x - offset)].  M.cpp:232,237 */
                    a = L + ((size_t)y * W + x) * C;
                    b = R + ((size_t)y * W + reflect_idx(x - off, W)) * C;
                } else {
                    /* left_border(Rect(offset,...)) with a REFLECT right pad.  M.cpp:249,254 */
                    a = R + ((size_t)y * W + x) * C;
                    b = L + ((size_t)y * W + reflect_idx(x + off, W)) * C;
                }
                if (C == 3)
                    out[(size_t)y * W + x] = mean3_u8(absdiff_u8(a[0], b[0]), absdiff_u8(a[1], b[1]),
                                                      absdiff_u8(a[2], b[2])); /* M.cpp:240-241 */
                else
                    out[(size_t)y * W + x] = (uint8_t)absdiff_u8(a[0], b[0]); /* M.cpp:274 */
            }
        }
    }
    return ORC_OK;
}

/* computeTAD, M.cpp:304-401: AD then compare(> T) -> 0/255 mask (App. B-4). */
int orc_compute_tad(const uint8_t* L, const uint8_t* R, int H, int W, int C, int disp_type,
                    int threshold_T, int minD, int numD, uint8_t* cost)
{
    int rc = orc_compute_ad(L, R, H, W, C, disp_type, minD, numD, cost);
    if (rc != ORC_OK) return rc;
    /* cv::compare(u8, int scalar): scalar compared exactly; T<0 -> all 255, T>=255 -> all 0 */
    for (size_t i = 0; i < (size_t)numD * H * W; i++) cost[i] = ((int)cost[i] > threshold_T) ? 255 : 0;
    return ORC_OK;
}

/* computeSD, M.cpp:670-759: the AD plane, then color_.mul(color_) on u8 (M.cpp:701,718,735,749) -- OpenCV's 8u multiply
 * saturates, so every AD >= 16 becomes 255. */
int orc_compute_sd(const uint8_t* L, const uint8_t* R, int H, int W, int C, int disp_type, int minD, int numD, uint8_t* cost)
{
    int rc = orc_compute_ad(L, R, H, W, C, disp_type, minD, numD, cost);
    if (rc != ORC_OK) return rc;
    for (size_t i = 0; i < (size_t)numD * H * W; i++) {
        const int sq = (int)cost[i] * (int)cost[i];
        cost[i] = (uint8_t)(sq > 255 ? 255 : sq);
    }
    return ORC_OK;
}

/* filter2D(src 8UC3, CV_32F, Scharr-x char kernel), REFLECT_101, on an image given through an
 * accessor so that the right image can be the REFLECT-padded one.  M.cpp:446-450 (A-7).
 * px(y, c, ch) must be valid for y in [0,H), c in [0,Wimg). */
typedef struct {
    const uint8_t* img;
    int H, W;     /* original size */
    int pad_left; /* REFLECT pad on the left (bordered image is W + pad_left wide) */
} padded_img_t;

static inline int padded_px(const padded_img_t* p, int y, int c, int ch)
{
    return p->img[((size_t)y * p->W + reflect_idx(c - p->pad_left, p->W)) * 3 + ch];
}

static void scharr_x_padded(const padded_img_t* p, float* grad /* H x (W+pad) x 3 */)
{
    const int Wb = p->W + p->pad_left, H = p->H;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int y = 0; y < H; y++) {
        int ym = reflect101_idx(y - 1, H), yp = reflect101_idx(y + 1, H);
        for (int c = 0; c < Wb; c++) {
            int cm = reflect101_idx(c - 1, Wb), cp = reflect101_idx(c + 1, Wb);
            for (int ch = 0; ch < 3; ch++) {
                int v = 3 * (padded_px(p, ym, cp, ch) - padded_px(p, ym, cm, ch)) +
                        10 * (padded_px(p, y, cp, ch) - padded_px(p, y, cm, ch)) +
                        3 * (padded_px(p, yp, cp, ch) - padded_px(p, yp, cm, ch));
                grad[((size_t)y * Wb + c) * 3 + ch] = (float)v;
            }
        }
    }
}

/* One TAD C+G value, M.cpp:455-484, from the three u8 colour abs-diffs and the three f32
 * gradient abs-diffs.  Exposed for known-answer tests K3/K4. */
float orc_similarity_pixel(int c0, int c1, int c2, float g0, float g1, float g2, double regularity,
                           double thresC, double thresG)
{
    /* colour term, u8 (M.cpp:459-465, App. B-5) */
    int color = mean3_u8(c0, c1, c2);
    int maskC = ((double)color > thresC) ? 1 : 0; /* compare(>thresC)/255 */
    /* color_.mul(mask) + thresC*(mask): addWeighted(m1,1,mask255,thresC/255) in float, cvRound,
     * saturate_cast<uchar> */
    float tf = (float)(color * maskC) * 1.0f + 255.0f * (float)maskC * (float)(thresC * (1.0 / 255.0));
    long cc_l = lrintf(tf);
    if (cc_l > 255) cc_l = 255;
    if (cc_l < 0) cc_l = 0;
    float cc = (float)cc_l; /* convertTo(CV_32FC1) */

    /* gradient term, f32 (M.cpp:470-482, App. B-6) */
    const float third = (float)(1.0 / 3.0);
    float g01 = g0 + g1;                 /* cv::add, exact (integers) */
    float g = g01 * third + g2 * third;  /* addWeighted 32f, float scalars, not fused (A-8) */
    int maskG = ((double)g > thresG) ? 1 : 0;
    float bit = (float)maskG;                   /* compare/255 -> 0/1 */
    float bit_not = (float)(255 - maskG);       /* bitwise_not of a 0/1 u8 mask: 255 or 254 */
    float gm = g * bit;                         /* curGridient_.mul(bitImg) */
    float cg = bit_not * (float)thresG + gm;    /* scaleAdd(bit_not, thresG, gm) */

    /* regularityR*cc + regularity*cg: addWeighted 32f (M.cpp:484) */
    float rr = (float)(1.0 - regularity), rg = (float)regularity;
    return cc * rr + cg * rg;
}

/* computeSimilarity, M.cpp:415-636.  Only the DISPARITY_LEFT + 3-channel branch (437-487)
 * can execute in the reference (App. B-7).  cost: numD planes H x W f32. */
int orc_compute_similarity(const uint8_t* L, const uint8_t* R, int H, int W, int C, double regularity,
                           double thresC, double thresG, int disp_type, int minD, int numD, float* cost)
{
    if (C != 3 || disp_type != DISPARITY_LEFT) return ORC_ERR_UNSUPPORTED_LAYOUT;
    const int max_off = minD + numD - 1; /* M.cpp:422-423 */
    const int Wb = W + max_off;
    float* gL = (float*)malloc((size_t)H * W * 3 * sizeof(float));
    float* gR = (float*)malloc((size_t)H * Wb * 3 * sizeof(float));
    if (!gL || !gR) { free(gL); free(gR); return ORC_ERR_ALLOC; }
    padded_img_t pl = {L, H, W, 0}, pr = {R, H, W, max_off};
    scharr_x_padded(&pl, gL); /* M.cpp:449 */
    scharr_x_padded(&pr, gR); /* M.cpp:450: gradient of the PADDED right image */

    for (int k = 0; k < numD; k++) {
        const int off = minD + k;
        float* out = cost + (size_t)k * H * W;
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int y = 0; y < H; y++) {
            for (int x = 0; x < W; x++) {
                const uint8_t* a = L + ((size_t)y * W + x) * 3;
                const int cb = max_off - off + x; /* column in the padded right image, M.cpp:455 */
                const uint8_t* b = R + ((size_t)y * W + reflect_idx(cb - max_off, W)) * 3;
                const float* ga = gL + ((size_t)y * W + x) * 3;
                const float* gb = gR + ((size_t)y * Wb + cb) * 3;
                out[(size_t)y * W + x] = orc_similarity_pixel(
                    absdiff_u8(a[0], b[0]), absdiff_u8(a[1], b[1]), absdiff_u8(a[2], b[2]),
                    fabsf(ga[0] - gb[0]), fabsf(ga[1] - gb[1]), fabsf(ga[2] - gb[2]), regularity, thresC,
                    thresG);
            }
        }
    }
    free(gL);
    free(gR);
    return ORC_OK;
}

/* computeSimilarity padded overload, M.cpp:651-668: every plane REFLECT-padded by win/2.
 * cost: numD planes (H+2h) x (W+2h). */
int orc_compute_similarity_padded(const uint8_t* L, const uint8_t* R, int H, int W, int C, double regularity,
                                  double thresC, double thresG, int disp_type, int win, int minD, int numD,
                                  float* cost)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW;
    const int h = win / 2, Hp = H + 2 * h, Wp = W + 2 * h;
    float* raw = (float*)malloc((size_t)numD * H * W * sizeof(float));
    if (!raw) return ORC_ERR_ALLOC;
    int rc = orc_compute_similarity(L, R, H, W, C, regularity, thresC, thresG, disp_type, minD, numD, raw);
    if (rc == ORC_OK) {
        for (int k = 0; k < numD; k++)
            for (int y = 0; y < Hp; y++)
                for (int x = 0; x < Wp; x++)
                    cost[((size_t)k * Hp + y) * Wp + x] =
                        raw[((size_t)k * H + reflect_idx(y - h, H)) * W + reflect_idx(x - h, W)];
    }
    free(raw);
    return rc;
}

/* boxFilter(src CV_32F -> CV_32F, Size(k,k), normalize=true, BORDER_REFLECT_101), single plane
 * with a pixel stride so interleaved channels can be filtered in place (A-9).
 * mode 0: canonical definition used for parity -- f64 window sum, rows of horizontal sums added
 *         in ascending order, one multiply by 1/(k*k) in f64, one rounding to f32.
 * mode 1: OpenCV's portable RowSum<float,double> + ColumnSum<double,float> sliding sums,
 *         literally (kept to measure how far the canonical definition is from it). */
static int g_box_mode = 0;
void orc_set_box_mode(int mode) { g_box_mode = mode; }

static void box_filter_plane(const float* src, int sstride, float* dst, int dstride, int H, int W, int k)
{
    const int h = k / 2; /* anchor = centre; for even k OpenCV's anchor is k/2 as well */
    const double scale = 1.0 / ((double)k * k);
    double* rows = (double*)malloc((size_t)H * W * sizeof(double));
    if (g_box_mode == 0) {
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                double s = 0;
                for (int i = 0; i < k; i++) s += (double)src[((size_t)y * W + reflect101_idx(x - h + i, W)) * sstride];
                rows[(size_t)y * W + x] = s;
            }
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                double s = 0;
                for (int j = 0; j < k; j++) s += rows[(size_t)reflect101_idx(y - h + j, H) * W + x];
                dst[((size_t)y * W + x) * dstride] = (float)(s * scale);
            }
    } else {
        /* RowSum: s = sum of first k; then s += S[i+k] - S[i] */
        for (int y = 0; y < H; y++) {
            double s = 0;
            for (int i = 0; i < k; i++) s += (double)src[((size_t)y * W + reflect101_idx(-h + i, W)) * sstride];
            rows[(size_t)y * W] = s;
            for (int x = 0; x < W - 1; x++) {
                s += (double)src[((size_t)y * W + reflect101_idx(x - h + k, W)) * sstride] -
                     (double)src[((size_t)y * W + reflect101_idx(x - h, W)) * sstride];
                rows[(size_t)y * W + x + 1] = s;
            }
        }
        /* ColumnSum: SUM = first k-1 rows; per output row: s0 = SUM + Sp; D = s0*scale; SUM = s0 - Sm */
        double* SUM = (double*)calloc((size_t)W, sizeof(double));
        for (int j = 0; j < k - 1; j++) {
            const double* Sp = rows + (size_t)reflect101_idx(-h + j, H) * W;
            for (int x = 0; x < W; x++) SUM[x] += Sp[x];
        }
        for (int y = 0; y < H; y++) {
            const double* Sp = rows + (size_t)reflect101_idx(y - h + k - 1, H) * W;
            const double* Sm = rows + (size_t)reflect101_idx(y - h, H) * W;
            for (int x = 0; x < W; x++) {
                double s0 = SUM[x] + Sp[x];
                dst[((size_t)y * W + x) * dstride] = (float)(s0 * scale);
                SUM[x] = s0 - Sm[x];
            }
        }
        free(SUM);
    }
    free(rows);
}

/* public single-plane box filter for tests */
void orc_box_filter(const float* src, float* dst, int H, int W, int k)
{
    float* tmp = (float*)malloc((size_t)H * W * sizeof(float));
    box_filter_plane(src, 1, tmp, 1, H, W, k);
    memcpy(dst, tmp, (size_t)H * W * sizeof(float));
    free(tmp);
}

/* getCostSAD_d for DISPARITY_LEFT / RIGHT over all disparities, M.cpp:2442-2503 as called from
 * M.cpp:2884-2889 / 2893-2898: gray abs-diff (right image read through the REFLECT pad), to f32,
 * normalised win x win box mean.  cost: numD planes H x W f32. */
int orc_cost_sad(const uint8_t* L, const uint8_t* R, int H, int W, int disp_type, int win, int minD, int numD,
                 float* cost)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW; /* M.cpp:2458-2462 */
    uint8_t* gl = (uint8_t*)malloc((size_t)H * W);
    uint8_t* gr = (uint8_t*)malloc((size_t)H * W);
    float* diff = (float*)malloc((size_t)H * W * sizeof(float));
    if (!gl || !gr || !diff) { free(gl); free(gr); free(diff); return ORC_ERR_ALLOC; }
    /* cvtColor of the padded image == padding of the gray image (per-pixel op) */
    orc_bgr2gray(L, H, W, gl);
    orc_bgr2gray(R, H, W, gr);
    for (int k = 0; k < numD; k++) {
        const int d = minD + k;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                int a, b;
                if (disp_type == DISPARITY_LEFT) { /* M.cpp:2477 */
                    a = gl[(size_t)y * W + x];
                    b = gr[(size_t)y * W + reflect_idx(x - d, W)];
                } else { /* M.cpp:2492 */
                    a = gl[(size_t)y * W + reflect_idx(x + d, W)];
                    b = gr[(size_t)y * W + x];
                }
                diff[(size_t)y * W + x] = (float)absdiff_u8(a, b);
            }
        box_filter_plane(diff, 1, cost + (size_t)k * H * W, 1, H, W, win); /* M.cpp:2480 */
    }
    free(gl); free(gr); free(diff);
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------
 * WTA (M.cpp:1144-1150, 2945-2961, 3032-3048, 3365-3381): strict '<' against DBL_MAX in
 * ascending d; NaN never selected.  The reference leaves never-updated pixels uninitialised
 * (App. B-16); this build defines them as 0.
 * ------------------------------------------------------------------------------------- */
void orc_wta(const float* volume, int nD, int H, int W, int minD, float* disp)
{
    for (size_t p = 0; p < (size_t)H * W; p++) {
        double best = DBL_MAX;
        float bd = 0.0f;
        for (int k = 0; k < nD; k++) {
            double c = (double)volume[(size_t)k * H * W + p];
            if (c < best) { best = c; bd = (float)(k + minD); }
        }
        disp[p] = bd;
    }
}

/* ---------------------------------------------------------------------------------------
 * Classic bilateral ASW, computeAdaptiveWeight, M.cpp:1016-1156
 * ------------------------------------------------------------------------------------- */

/* The (win*win-1)-entry tap list of the reference, with both of its index conventions:
 * weight direction (dxw,dyw) from the build loops M.cpp:1044-1053, sample offset (dxs,dys) from
 * the consume loop M.cpp:1088-1102 (transposed + centre quirk, App. B-2). */
void orc_classic_taps(int ks, int* dxw, int* dyw, int* dxs, int* dys)
{
    int h = ks / 2, n = 0;
    for (int j = -h; j < h + 1; j++)
        for (int i = -h; i < h + 1; i++) {
            if (i == 0 && j == 0) continue;
            dxw[n] = i;
            dyw[n] = j;
            n++;
        }
    for (int i = 0; i < ks * ks - 1; i++) {
        int kx, ky;
        if (i > ks * ks / 2) { kx = (i + 1) / ks; ky = (i + 1) % ks; }
        else { kx = i / ks; ky = i % ks; }
        dxs[i] = -h + kx;
        dys[i] = -h + ky;
    }
}

/* Literal restatement: materialises the 2*(win^2-1) weight maps exactly as M.cpp:1041-1072 does.
 * Memory = 2*(win*win-1)*H*W*4 bytes, so small images only.  vol (optional) receives (float)E,
 * (numD+1) planes. */
int orc_asw_classic_literal(const uint8_t* Lbgr, const uint8_t* Rbgr, int H, int W, double gamma_c,
                            double gamma_g, int disp_type, int win, int minD, int numD, float* disp, float* vol)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW; /* build decision: the reference has no guard */
    const int ks = win, nt = ks * ks - 1;
    const int max_offset = minD + numD, min_offset = minD; /* M.cpp:1021-1022: inclusive, B-1 */
    const double k = 3;
    uint8_t* left = (uint8_t*)malloc((size_t)H * W);
    uint8_t* right = (uint8_t*)malloc((size_t)H * W);
    float* wl = (float*)malloc((size_t)nt * H * W * sizeof(float));
    float* wr = (float*)malloc((size_t)nt * H * W * sizeof(float));
    double* min_asw = (double*)malloc((size_t)H * W * sizeof(double));
    if (!left || !right || !wl || !wr || !min_asw) { free(left); free(right); free(wl); free(wr); free(min_asw); return ORC_ERR_ALLOC; }
    orc_bgr2gray(Lbgr, H, W, left);
    orc_bgr2gray(Rbgr, H, W, right);
    for (size_t p = 0; p < (size_t)H * W; p++) { min_asw[p] = DBL_MAX; disp[p] = 0.0f; }

    int n = 0;
    for (int j = -ks / 2; j < ks / 2 + 1; j++)
        for (int i = -ks / 2; i < ks / 2 + 1; i++) {
            if (i == 0 && j == 0) continue;
            double delta_g = sqrt((double)(i * i + j * j));
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    int nx = imin(imax(0, x + i), W - 1), ny = imin(imax(0, y + j), H - 1);
                    double dc1 = fabs((double)(left[(size_t)ny * W + nx] - left[(size_t)y * W + x]));
                    double dc2 = fabs((double)(right[(size_t)ny * W + nx] - right[(size_t)y * W + x]));
                    wl[((size_t)n * H + y) * W + x] = (float)(k * exp(-(dc1 / gamma_c + delta_g / gamma_g)));
                    wr[((size_t)n * H + y) * W + x] = (float)(k * exp(-(dc2 / gamma_c + delta_g / gamma_g)));
                }
            n++;
        }

    for (int offset = min_offset; offset <= max_offset; offset++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                double numerator = 0, denominator = 0;
                for (int i = 0; i < ks * ks - 1; i++) {
                    int kx, ky;
                    if (i > ks * ks / 2) { kx = (i + 1) / ks; ky = (i + 1) % ks; }
                    else { kx = i / ks; ky = i % ks; }
                    int nx = imin(imax(0, x - ks / 2 + kx), W - 1);
                    int ny = imin(imax(0, y - ks / 2 + ky), H - 1);
                    if (disp_type == DISPARITY_LEFT) { /* M.cpp:1104-1108 */
                        float a = wl[((size_t)i * H + y) * W + x];
                        float b = wr[((size_t)i * H + y) * W + imax(0, x - offset)];
                        float ab = a * b; /* float*float stays float in C++ */
                        numerator += ab * fabs((double)(left[(size_t)ny * W + nx] - right[(size_t)ny * W + imax(0, nx - offset)]));
                        denominator += ab;
                    } else { /* M.cpp:1134-1138 */
                        float a = wl[((size_t)i * H + y) * W + imin(x + offset, W - 1)];
                        float b = wr[((size_t)i * H + y) * W + x];
                        float ab = a * b;
                        numerator += ab * fabs((double)(right[(size_t)ny * W + nx] - left[(size_t)ny * W + imin(nx + offset, W - 1)]));
                        denominator += ab;
                    }
                }
                double E = numerator / denominator;
                if (vol) vol[((size_t)(offset - min_offset) * H + y) * W + x] = (float)E;
                if (E < min_asw[(size_t)y * W + x]) { /* M.cpp:1145-1150 */
                    min_asw[(size_t)y * W + x] = E;
                    disp[(size_t)y * W + x] = (float)offset;
                }
            }
    free(left); free(right); free(wl); free(wr); free(min_asw);
    return ORC_OK;
}

/* Second, independent restatement: weights looked up from a [dist-class][delta_c] table built
 * with the reference's expression (M.cpp:1054,1065), nothing materialised, rows in parallel.
 * Same tap order per (pixel, d), so E is bit-identical to the literal version.  This is the
 * version the CPU baseline times.  y0,y1: row range to compute (for bounded samples). */
int orc_asw_classic_rows(const uint8_t* Lbgr, const uint8_t* Rbgr, int H, int W, double gamma_c, double gamma_g,
                         int disp_type, int win, int minD, int numD, int y0, int y1, float* disp, float* vol)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW;
    const int ks = win, nt = ks * ks - 1, h = ks / 2;
    const int nD = numD + 1;
    uint8_t* left = (uint8_t*)malloc((size_t)H * W);
    uint8_t* right = (uint8_t*)malloc((size_t)H * W);
    int* taps = (int*)malloc((size_t)nt * 4 * sizeof(int));
    const int nr2 = 2 * h * h + 1; /* table indexed by i*i+j*j directly */
    float* lut = (float*)malloc((size_t)nr2 * 256 * sizeof(float));
    if (!left || !right || !taps || !lut) { free(left); free(right); free(taps); free(lut); return ORC_ERR_ALLOC; }
    int *dxw = taps, *dyw = taps + nt, *dxs = taps + 2 * nt, *dys = taps + 3 * nt;
    orc_classic_taps(ks, dxw, dyw, dxs, dys);
    orc_bgr2gray(Lbgr, H, W, left);
    orc_bgr2gray(Rbgr, H, W, right);
    for (int r2 = 0; r2 < nr2; r2++) {
        double delta_g = sqrt((double)r2);
        for (int dc = 0; dc < 256; dc++)
            lut[r2 * 256 + dc] = (float)(3.0 * exp(-((double)dc / gamma_c + delta_g / gamma_g)));
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < W; x++) {
            double best = DBL_MAX;
            float bd = 0.0f;
            for (int k = 0; k < nD; k++) {
                const int offset = minD + k;
                double numerator = 0, denominator = 0;
                for (int i = 0; i < nt; i++) {
                    const int r2 = dxw[i] * dxw[i] + dyw[i] * dyw[i];
                    const int nx = imin(imax(0, x + dxs[i]), W - 1), ny = imin(imax(0, y + dys[i]), H - 1);
                    float a, b;
                    double c;
                    if (disp_type == DISPARITY_LEFT) {
                        const int xr = imax(0, x - offset);
                        const int wnx = imin(imax(0, x + dxw[i]), W - 1), wny = imin(imax(0, y + dyw[i]), H - 1);
                        const int rnx = imin(imax(0, xr + dxw[i]), W - 1);
                        a = lut[r2 * 256 + absdiff_u8(left[(size_t)wny * W + wnx], left[(size_t)y * W + x])];
                        b = lut[r2 * 256 + absdiff_u8(right[(size_t)wny * W + rnx], right[(size_t)y * W + xr])];
                        c = (double)absdiff_u8(left[(size_t)ny * W + nx], right[(size_t)ny * W + imax(0, nx - offset)]);
                    } else {
                        const int xl = imin(x + offset, W - 1);
                        const int wny = imin(imax(0, y + dyw[i]), H - 1);
                        const int lnx = imin(imax(0, xl + dxw[i]), W - 1), rnx = imin(imax(0, x + dxw[i]), W - 1);
                        a = lut[r2 * 256 + absdiff_u8(left[(size_t)wny * W + lnx], left[(size_t)y * W + xl])];
                        b = lut[r2 * 256 + absdiff_u8(right[(size_t)wny * W + rnx], right[(size_t)y * W + x])];
                        c = (double)absdiff_u8(right[(size_t)ny * W + nx], left[(size_t)ny * W + imin(nx + offset, W - 1)]);
                    }
                    float ab = a * b;
                    numerator += ab * c;
                    denominator += ab;
                }
                double E = numerator / denominator;
                if (vol) vol[((size_t)k * H + y) * W + x] = (float)E;
                if (E < best) { best = E; bd = (float)offset; }
            }
            disp[(size_t)y * W + x] = bd;
        }
    }
    free(left); free(right); free(taps); free(lut);
    return ORC_OK;
}

int orc_asw_classic(const uint8_t* Lbgr, const uint8_t* Rbgr, int H, int W, double gamma_c, double gamma_g,
                    int disp_type, int win, int minD, int numD, float* disp, float* vol)
{
    return orc_asw_classic_rows(Lbgr, Rbgr, H, W, gamma_c, gamma_g, disp_type, win, minD, numD, 0, H, disp, vol);
}

/* ---------------------------------------------------------------------------------------
 * Geodesic ASW, M.cpp:1321-1534
 * ------------------------------------------------------------------------------------- */

static inline float color_dist(const uint8_t* a, const uint8_t* b) /* getColorDist M.cpp:1321-1326 */
{
    return (float)(fabs((double)(a[0] - b[0])) + fabs((double)(a[1] - b[1])) + fabs((double)(a[2] - b[2])));
}

/* getWinGeoDist, M.cpp:1328-1390, on one (win+2)^2 window.  img(r,c) -> pointer to BGR. */
static void win_geo_dist(const uint8_t* img, int H, int W, int px, int py, int win, int iterTime, float* wd)
{
    const int h = win / 2, n = win + 2;
#define WIMG(r, c) (img + ((size_t)reflect_idx(py - h - 1 + (r), H) * W + reflect_idx(px - h - 1 + (c), W)) * 3)
#define WD(r, c) wd[(r) * n + (c)]
    for (int it = 0; it < iterTime; it++) {
        if (it / 2 == 1) { /* forward raster: L, UL, U, UR  (M.cpp:1341-1364) */
            for (int r = 1; r <= win; r++)
                for (int c = 1; c <= win; c++) {
                    float v;
                    v = WD(r, c - 1) + color_dist(WIMG(r, c - 1), WIMG(r, c));
                    WD(r, c) = fminf(WD(r, c), v);
                    v = WD(r - 1, c - 1) + color_dist(WIMG(r - 1, c - 1), WIMG(r, c));
                    WD(r, c) = fminf(WD(r, c), v);
                    v = WD(r - 1, c) + color_dist(WIMG(r - 1, c), WIMG(r, c));
                    WD(r, c) = fminf(WD(r, c), v);
                    v = WD(r - 1, c + 1) + color_dist(WIMG(r - 1, c + 1), WIMG(r, c));
                    WD(r, c) = fminf(WD(r, c), v);
                }
        } else if (it / 2 == 0) { /* backward raster: R, BR, B, BL  (M.cpp:1365-1388) */
            for (int r = win; r > 0; r--)
                for (int c = win; c > 0; c--) {
                    float v;
                    v = WD(r, c + 1) + color_dist(WIMG(r, c + 1), WIMG(r, c));
                    WD(r, c) = fminf(WD(r, c), v);
                    v = WD(r + 1, c + 1) + color_dist(WIMG(r + 1, c + 1), WIMG(r, c));
                    WD(r, c) = fminf(WD(r, c), v);
                    v = WD(r + 1, c) + color_dist(WIMG(r + 1, c), WIMG(r, c));
                    WD(r, c) = fminf(WD(r, c), v);
                    v = WD(r + 1, c - 1) + color_dist(WIMG(r + 1, c - 1), WIMG(r, c));
                    WD(r, c) = fminf(WD(r, c), v);
                }
        }
    }
#undef WIMG
#undef WD
}

/* getGeodesicDist, M.cpp:1392-1424.  out[(y*W+x)*win*win + j*win + i]. */
int orc_geodesic_dist(const uint8_t* img, int H, int W, int win, int iterTime, float* out)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW;
    const int h = win / 2, n = win + 2;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int y = 0; y < H; y++) {
        float* wd = (float*)malloc((size_t)n * n * sizeof(float));
        for (int x = 0; x < W; x++) {
            for (int q = 0; q < n * n; q++) wd[q] = FLT_MAX; /* M.cpp:1416 */
            wd[(h + 1) * n + (h + 1)] = 0;                    /* M.cpp:1417 */
            win_geo_dist(img, H, W, x, y, win, iterTime, wd);
            float* o = out + ((size_t)y * W + x) * win * win;
            for (int j = 0; j < win; j++)
                for (int i = 0; i < win; i++) o[j * win + i] = wd[(j + 1) * n + (i + 1)]; /* M.cpp:1420-1421 */
        }
        free(wd);
    }
    return ORC_OK;
}

/* computeAdaptiveWeight_geodesic, M.cpp:1436-1534. */
int orc_asw_geodesic(const uint8_t* L, const uint8_t* R, int H, int W, int disp_type, int win, int minD, int numD,
                     float* disp, float* vol)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW; /* M.cpp:1440-1443 */
    const int ks = win, h = ks / 2, nD = numD + 1; /* inclusive range, M.cpp:1447,1467 */
    float* wL = (float*)malloc((size_t)H * W * ks * ks * sizeof(float));
    float* wR = (float*)malloc((size_t)H * W * ks * ks * sizeof(float));
    if (!wL || !wR) { free(wL); free(wR); return ORC_ERR_ALLOC; }
    orc_geodesic_dist(L, H, W, win, 3, wL); /* M.cpp:1464 */
    orc_geodesic_dist(R, H, W, win, 3, wR); /* M.cpp:1465 */
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            double best = DBL_MAX;
            float bd = 0.0f;
            for (int k = 0; k < nD; k++) {
                const int offset = minD + k;
                double numerator = 0, denominator = 0;
                const float *pl, *pr;
                if (disp_type == DISPARITY_LEFT) {
                    pl = wL + ((size_t)y * W + x) * ks * ks;
                    pr = wR + ((size_t)y * W + imax(0, x - offset)) * ks * ks;
                } else {
                    pl = wL + ((size_t)y * W + imin(x + offset, W - 1)) * ks * ks;
                    pr = wR + ((size_t)y * W + x) * ks * ks;
                }
                for (int j = 0; j < ks; j++)
                    for (int i = 0; i < ks; i++) {
                        int nx = imin(imax(0, x - h + i), W - 1), ny = imin(imax(0, y - h + j), H - 1);
                        float c;
                        if (disp_type == DISPARITY_LEFT)
                            c = color_dist(L + ((size_t)ny * W + nx) * 3, R + ((size_t)ny * W + imax(0, nx - offset)) * 3);
                        else
                            c = color_dist(R + ((size_t)ny * W + nx) * 3, L + ((size_t)ny * W + imin(W - 1, nx + offset)) * 3);
                        float ab = pl[j * ks + i] * pr[j * ks + i]; /* f32 product */
                        float abc = ab * c;                         /* f32 product, M.cpp:1488-1490 */
                        numerator += abc;
                        denominator += ab;
                    }
                double E = numerator / denominator;
                if (vol) vol[((size_t)k * H + y) * W + x] = (float)E;
                if (E < best) { best = E; bd = (float)offset; }
            }
            disp[(size_t)y * W + x] = bd;
        }
    free(wL); free(wR);
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------
 * Guided filter, M.cpp:2727-2854, and the two ASW variants built on it
 * ------------------------------------------------------------------------------------- */

/* normalize(src, dst, 0, 1, NORM_MINMAX, CV_32F) parameters (A-10): returns float scale/shift */
static void minmax_scale(double smin, double smax, float* a, float* b)
{
    double scale = (smax - smin > DBL_EPSILON) ? 1.0 / (smax - smin) : 0.0;
    double shift = 0.0 - smin * scale;
    *a = (float)scale;
    *b = (float)shift;
}

/* getGuidedFilter(guidedImg 8U C-channel, inputP f32, r, eps) -> q f32.  C must be 3 or 6
 * (multiChl_to_oneChl_mul returns an empty Mat otherwise, M.cpp:2732-2734). */
int orc_guided_filter(const uint8_t* guide, int C, const float* P_in, int H, int W, int r, double eps, float* q)
{
    if (C != 3 && C != 6) return ORC_ERR_UNSUPPORTED_LAYOUT;
    const size_t N = (size_t)H * W;
    float* I = (float*)malloc(N * C * sizeof(float));
    float* P = (float*)malloc(N * sizeof(float));
    float* meanI = (float*)malloc(N * C * sizeof(float));
    float* meanP = (float*)malloc(N * sizeof(float));
    float* corrIP = (float*)malloc(N * C * sizeof(float));
    float* corrI = (float*)malloc(N * C * sizeof(float));
    float* tmp = (float*)malloc(N * sizeof(float));
    float* a = (float*)malloc(N * C * sizeof(float));
    float* b = (float*)malloc(N * sizeof(float));
    if (!I || !P || !meanI || !meanP || !corrIP || !corrI || !tmp || !a || !b) {
        free(I); free(P); free(meanI); free(meanP); free(corrIP); free(corrI); free(tmp); free(a); free(b);
        return ORC_ERR_ALLOC;
    }
    /* M.cpp:2774: min/max over ALL channels of the guide */
    {
        int mn = 255, mx = 0;
        for (size_t i = 0; i < N * C; i++) { if (guide[i] < mn) mn = guide[i]; if (guide[i] > mx) mx = guide[i]; }
        float sa, sb;
        minmax_scale(mn, mx, &sa, &sb);
        for (size_t i = 0; i < N * C; i++) I[i] = (float)guide[i] * sa + sb; /* convertTo 8u->32f, float a,b */
    }
    /* M.cpp:2775: per-call (= per disparity slice) min/max of P */
    {
        /* NaN entries (they only arise from the NCC costs of GuidedF_3: 0/0 on flat windows) are skipped, as
         * minMaxIdx's ordered comparisons skip them; an all-NaN slice gives min = +inf, max = -inf -> scale 0 */
        float mn = INFINITY, mx = -INFINITY;
        for (size_t i = 0; i < N; i++) { if (P_in[i] < mn) mn = P_in[i]; if (P_in[i] > mx) mx = P_in[i]; }
        float sa, sb;
        minmax_scale((double)mn, (double)mx, &sa, &sb);
        for (size_t i = 0; i < N; i++) P[i] = P_in[i] * sa + sb;
    }
    for (int c = 0; c < C; c++) box_filter_plane(I + c, C, meanI + c, C, H, W, r); /* M.cpp:2778 */
    box_filter_plane(P, 1, meanP, 1, H, W, r);                                     /* M.cpp:2780 */
    for (int c = 0; c < C; c++) {                                                  /* M.cpp:2787-2792 */
        for (size_t i = 0; i < N; i++) tmp[i] = I[i * C + c] * P[i];
        box_filter_plane(tmp, 1, corrIP + c, C, H, W, r);
    }
    for (int c = 0; c < C; c++) {                                                  /* M.cpp:2796 */
        for (size_t i = 0; i < N; i++) tmp[i] = I[i * C + c] * I[i * C + c];
        box_filter_plane(tmp, 1, corrI + c, C, H, W, r);
    }
    const float epsf = (float)eps;
    for (size_t i = 0; i < N; i++) {
        float dot = 0.0f;
        for (int c = 0; c < C; c++) {
            float mI = meanI[i * C + c];
            float mm = mI * mI;
            float var = corrI[i * C + c] - mm;            /* M.cpp:2799 */
            float mp = mI * meanP[i];                     /* M.cpp:2809 */
            float cov = corrIP[i * C + c] - mp;           /* M.cpp:2815 */
            float den = 1.0f * epsf + var;                /* scaleAdd(ones, eps, var) */
            float ac = cov / den;                         /* M.cpp:2846 (A-11) */
            a[i * C + c] = ac;
            float pr = ac * mI;                           /* operator* Vec, M.cpp:22-31: left-to-right */
            dot = (c == 0) ? pr : dot + pr;
        }
        b[i] = meanP[i] - dot;                            /* M.cpp:2847 */
    }
    for (int c = 0; c < C; c++) {                                                  /* M.cpp:2849 */
        for (size_t i = 0; i < N; i++) tmp[i] = a[i * C + c];
        box_filter_plane(tmp, 1, a + c, C, H, W, r);
    }
    memcpy(tmp, b, N * sizeof(float));
    box_filter_plane(tmp, 1, b, 1, H, W, r);                                       /* M.cpp:2850 */
    for (size_t i = 0; i < N; i++) {                                               /* M.cpp:2852 */
        float dot = 0.0f;
        for (int c = 0; c < C; c++) {
            float pr = a[i * C + c] * I[i * C + c];
            dot = (c == 0) ? pr : dot + pr;
        }
        q[i] = dot + b[i];
    }
    free(I); free(P); free(meanI); free(meanP); free(corrIP); free(corrI); free(tmp); free(a); free(b);
    return ORC_OK;
}

/* computeAdaptiveWeight_GuidedF_2, M.cpp:2976-3050: TAD C+G cost, guide = left image. */
int orc_asw_guided2(const uint8_t* L, const uint8_t* R, int H, int W, int disp_type, double eps, int win, int minD,
                    int numD, float* disp, float* vol)
{
    if (disp_type != DISPARITY_LEFT) return ORC_ERR_UNSUPPORTED_LAYOUT; /* App. B-7 */
    const size_t N = (size_t)H * W;
    float* costs = (float*)malloc((size_t)numD * N * sizeof(float));
    float* qv = vol ? vol : (float*)malloc((size_t)numD * N * sizeof(float));
    if (!costs || !qv) { free(costs); if (!vol) free(qv); return ORC_ERR_ALLOC; }
    int rc = orc_compute_similarity(L, R, H, W, 3, 0.4, 10, 50, disp_type, minD, numD, costs); /* M.cpp:2990 */
    if (rc == ORC_OK) {
        for (int i = 0; i < numD && rc == ORC_OK; i++) /* M.cpp:2995-3006 */
            rc = orc_guided_filter(L, 3, costs + (size_t)i * N, H, W, win, eps, qv + (size_t)i * N);
        if (rc == ORC_OK) orc_wta(qv, numD, H, W, minD, disp); /* M.cpp:3032-3048 */
    }
    free(costs);
    if (!vol) free(qv);
    return rc;
}

/* computeAdaptiveWeight_GuidedF, M.cpp:2867-2963: SAD cost, 6-channel guide [L, R shifted by d]. */
int orc_asw_guided(const uint8_t* L, const uint8_t* R, int H, int W, int disp_type, double eps, int win, int minD,
                   int numD, float* disp, float* vol)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW;
    const size_t N = (size_t)H * W;
    float* costs = (float*)malloc((size_t)numD * N * sizeof(float));
    float* qv = vol ? vol : (float*)malloc((size_t)numD * N * sizeof(float));
    uint8_t* guide = (uint8_t*)malloc(N * 6);
    if (!costs || !qv || !guide) { free(costs); if (!vol) free(qv); free(guide); return ORC_ERR_ALLOC; }
    int rc = orc_cost_sad(L, R, H, W, disp_type, win, minD, numD, costs); /* M.cpp:2884-2898 */
    for (int i = 0; i < numD && rc == ORC_OK; i++) {
        const int d = minD + i;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const uint8_t *pl, *pr;
                if (disp_type == DISPARITY_LEFT) { /* M.cpp:2908-2912: Rect(numDisparity-i-1) == shift by d */
                    pl = L + ((size_t)y * W + x) * 3;
                    pr = R + ((size_t)y * W + reflect_idx(x - d, W)) * 3;
                } else { /* M.cpp:2925-2929 */
                    pl = L + ((size_t)y * W + reflect_idx(x + d, W)) * 3;
                    pr = R + ((size_t)y * W + x) * 3;
                }
                uint8_t* g = guide + ((size_t)y * W + x) * 6;
                g[0] = pl[0]; g[1] = pl[1]; g[2] = pl[2]; g[3] = pr[0]; g[4] = pr[1]; g[5] = pr[2];
            }
        rc = orc_guided_filter(guide, 6, costs + (size_t)i * N, H, W, win, eps, qv + (size_t)i * N);
    }
    if (rc == ORC_OK) orc_wta(qv, numD, H, W, minD, disp);
    free(costs); free(guide);
    if (!vol) free(qv);
    return rc;
}

/* ---------------------------------------------------------------------------------------
 * NCC cost, computeNCC (both overloads) + getInputImgNCC, M.cpp:767-1013, and computeAdaptiveWeight_GuidedF_3,
 * M.cpp:3063-3137.
 *  - gray conversion is COLOR_RGB2GRAY on BGR data (M.cpp:835,840,948,953): the R and B coefficients swap;
 *  - support window of (y,x): REFLECT-padded gray image (as f32) minus the box mean at (y,x); the box mean is
 *    boxFilter's default BORDER_REFLECT_101 -- the two borders differ (M.cpp:782-786);
 *  - the "other" image is padded by max_offset REFLECT columns first (left side for DISPARITY_LEFT, right side for
 *    DISPARITY_RIGHT) and windows / means are taken on that padded image (M.cpp:852-857, 882-887);
 *  - cost = sum(l*r) / (sum(l*l) * sum(r*r)): NO square root (M.cpp:867-868); products in f32 (Mat::mul), sums in f64.
 *    The order in which cv::sum adds the 225 products cannot be verified offline: restated row-major;
 *  - volume overload: every offset minD..max_offset, each plane cast to f32 and min-max normalised (M.cpp:968-983);
 *  - disparity overload: offsets minD..max_offset-1 only (M.cpp:864: `<`), DISPARITY_LEFT keeps the SMALLEST cost,
 *    DISPARITY_RIGHT compares `cost > DBL_MAX`, never true: the reference returns uninitialised memory, 0 here.
 * ------------------------------------------------------------------------------------- */
void orc_rgb2gray(const uint8_t* bgr, int H, int W, uint8_t* gray)
{
    const int wide = g_gray_bits == 15; /* channel 0 is taken for R */
    const int B2Y = wide ? 3735 : 1868, G2Y = wide ? 19235 : 9617, R2Y = wide ? 9798 : 4899, shift = wide ? 15 : 14;
    for (long i = 0; i < (long)H * W; i++) {
        int r = bgr[3 * i], g = bgr[3 * i + 1], b = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * B2Y + g * G2Y + r * R2Y + (1 << (shift - 1))) >> shift);
    }
}

typedef struct { int H, W; uint8_t* g; float* mean; double* ss; } ncc_img_t; /* gray image, box means, sum of squares */

/* window element (r,c) of getInputImgNCC's support window at (y,x): REFLECT border, f32 subtraction (M.cpp:782-795) */
static inline float ncc_elem(const ncc_img_t* im, int y, int x, int r, int c, int h)
{
    const int yy = reflect_idx(y - h + r, im->H), xx = reflect_idx(x - h + c, im->W);
    return (float)im->g[(size_t)yy * im->W + xx] - im->mean[(size_t)y * im->W + x];
}

static int ncc_prepare(ncc_img_t* im, const uint8_t* gray, int H, int W, int padL, int padR, int win)
{
    const int Wp = W + padL + padR, h = win / 2;
    im->H = H; im->W = Wp;
    im->g = (uint8_t*)malloc((size_t)H * Wp);
    im->mean = (float*)malloc((size_t)H * Wp * sizeof(float));
    im->ss = (double*)malloc((size_t)H * Wp * sizeof(double));
    float* f = (float*)malloc((size_t)H * Wp * sizeof(float));
    if (!im->g || !im->mean || !im->ss || !f) { free(f); return ORC_ERR_ALLOC; }
    for (int y = 0; y < H; y++)
        for (int c = 0; c < Wp; c++) { /* copyMakeBorder(..., BORDER_REFLECT), M.cpp:852,882 */
            im->g[(size_t)y * Wp + c] = gray[(size_t)y * W + reflect_idx(c - padL, W)];
            f[(size_t)y * Wp + c] = (float)im->g[(size_t)y * Wp + c];
        }
    box_filter_plane(f, 1, im->mean, 1, H, Wp, win); /* boxFilter(src, CV_32FC1, Size(win,win)), M.cpp:785-786 */
    free(f);
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < Wp; x++) {
            double s = 0;
            for (int r = 0; r < win; r++)
                for (int c = 0; c < win; c++) { float v = ncc_elem(im, y, x, r, c, h); float p = v * v; s += (double)p; }
            im->ss[(size_t)y * Wp + x] = s;
        }
    return ORC_OK;
}
static void ncc_free(ncc_img_t* im) { free(im->g); free(im->mean); free(im->ss); }

static inline double ncc_cost(const ncc_img_t* ref, int y, int x, const ncc_img_t* oth, int xo, int win)
{
    const int h = win / 2;
    double s = 0;
    for (int r = 0; r < win; r++)
        for (int c = 0; c < win; c++) {
            float p = ncc_elem(ref, y, x, r, c, h) * ncc_elem(oth, y, xo, r, c, h); /* Mat::mul on CV_32F */
            s += (double)p;
        }
    return s / (ref->ss[(size_t)y * ref->W + x] * oth->ss[(size_t)y * oth->W + xo]);
}

/* raw = 0: the volume overload (M.cpp:924-1013), numD planes, f32, each min-max normalised;
 * raw = 1: the same planes before normalize() (what the GPU parity tests also look at). */
int orc_cost_ncc(const uint8_t* Lbgr, const uint8_t* Rbgr, int H, int W, int disp_type, int win, int minD, int numD, int raw,
                 float* cost)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW; /* M.cpp:939-942 */
    if (disp_type != DISPARITY_LEFT && disp_type != DISPARITY_RIGHT) return ORC_OK; /* no branch taken */
    const int max_offset = minD + numD - 1;
    const size_t N = (size_t)H * W;
    uint8_t* gl = (uint8_t*)malloc(N);
    uint8_t* gr = (uint8_t*)malloc(N);
    orc_rgb2gray(Lbgr, H, W, gl);
    orc_rgb2gray(Rbgr, H, W, gr);
    ncc_img_t ref, oth;
    int rc;
    if (disp_type == DISPARITY_LEFT) { rc = ncc_prepare(&ref, gl, H, W, 0, 0, win); if (!rc) rc = ncc_prepare(&oth, gr, H, W, max_offset, 0, win); }
    else { rc = ncc_prepare(&ref, gr, H, W, 0, 0, win); if (!rc) rc = ncc_prepare(&oth, gl, H, W, 0, max_offset, win); }
    if (rc) return rc;
    for (int offset = minD; offset <= max_offset; offset++) {
        float* plane = cost + (size_t)(offset - minD) * N;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const int xo = disp_type == DISPARITY_LEFT ? x + max_offset - offset : x + offset; /* M.cpp:975, 1002 */
                plane[(size_t)y * W + x] = (float)ncc_cost(&ref, y, x, &oth, xo, win);
            }
        if (!raw) { /* normalize(curCost_, curCost_norm, 0, 1, NORM_MINMAX), M.cpp:981-983 */
            float mn = INFINITY, mx = -INFINITY;
            for (size_t i = 0; i < N; i++) { if (plane[i] < mn) mn = plane[i]; if (plane[i] > mx) mx = plane[i]; }
            float sa, sb;
            minmax_scale((double)mn, (double)mx, &sa, &sb);
            for (size_t i = 0; i < N; i++) plane[i] = plane[i] * sa + sb;
        }
    }
    ncc_free(&ref); ncc_free(&oth); free(gl); free(gr);
    return ORC_OK;
}

/* computeNCC -> disparity, M.cpp:812-913 */
int orc_ncc_disparity(const uint8_t* Lbgr, const uint8_t* Rbgr, int H, int W, int disp_type, int win, int minD, int numD,
                      float* disp)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW; /* M.cpp:828-831 */
    const int max_offset = minD + numD - 1;
    const size_t N = (size_t)H * W;
    for (size_t i = 0; i < N; i++) disp[i] = 0.0f; /* never-written pixels: 0 (reference: uninitialised Mat) */
    if (disp_type != DISPARITY_LEFT) return ORC_OK;  /* RIGHT: `cost_d > DBL_MAX` never holds (M.cpp:896) */
    uint8_t* gl = (uint8_t*)malloc(N);
    uint8_t* gr = (uint8_t*)malloc(N);
    orc_rgb2gray(Lbgr, H, W, gl);
    orc_rgb2gray(Rbgr, H, W, gr);
    ncc_img_t ref, oth;
    int rc = ncc_prepare(&ref, gl, H, W, 0, 0, win);
    if (!rc) rc = ncc_prepare(&oth, gr, H, W, max_offset, 0, win);
    if (rc) return rc;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            double best = DBL_MAX;
            for (int offset = minD; offset < max_offset; offset++) { /* exclusive upper bound, M.cpp:864 */
                double c = ncc_cost(&ref, y, x, &oth, x + max_offset - offset, win);
                if (c < best) { best = c; disp[(size_t)y * W + x] = (float)offset; }
            }
        }
    ncc_free(&ref); ncc_free(&oth); free(gl); free(gr);
    return ORC_OK;
}

/* computeAdaptiveWeight_GuidedF_3, M.cpp:3063-3137: normalised NCC planes filtered with the 6-channel guide
 * [L, R shifted by d] (LEFT, M.cpp:3085-3096) or with the plain right image (RIGHT: the 6-channel guide is built but
 * `rightImg` is what getGuidedFilter receives, M.cpp:3100-3112). */
int orc_asw_guided3(const uint8_t* L, const uint8_t* R, int H, int W, int disp_type, double eps, int win, int minD,
                    int numD, float* disp, float* vol)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW; /* computeNCC returns without costs, costs_ds[i] then throws */
    if (disp_type != DISPARITY_LEFT && disp_type != DISPARITY_RIGHT) return ORC_ERR_UNSUPPORTED_LAYOUT;
    const size_t N = (size_t)H * W;
    float* costs = (float*)malloc((size_t)numD * N * sizeof(float));
    float* qv = vol ? vol : (float*)malloc((size_t)numD * N * sizeof(float));
    uint8_t* guide = (uint8_t*)malloc(N * 6);
    if (!costs || !qv || !guide) { free(costs); if (!vol) free(qv); free(guide); return ORC_ERR_ALLOC; }
    int rc = orc_cost_ncc(L, R, H, W, disp_type, win, minD, numD, 0, costs);
    for (int i = 0; i < numD && rc == ORC_OK; i++) {
        if (disp_type == DISPARITY_LEFT) {
            const int d = minD + i; /* Rect(numDisparity-i-1) on a pad of max_offset columns == shift by minD+i */
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    const uint8_t* pl = L + ((size_t)y * W + x) * 3;
                    const uint8_t* pr = R + ((size_t)y * W + reflect_idx(x - d, W)) * 3;
                    uint8_t* g = guide + ((size_t)y * W + x) * 6;
                    g[0] = pl[0]; g[1] = pl[1]; g[2] = pl[2]; g[3] = pr[0]; g[4] = pr[1]; g[5] = pr[2];
                }
            rc = orc_guided_filter(guide, 6, costs + (size_t)i * N, H, W, win, eps, qv + (size_t)i * N);
        } else {
            rc = orc_guided_filter(R, 3, costs + (size_t)i * N, H, W, win, eps, qv + (size_t)i * N);
        }
    }
    if (rc == ORC_OK) orc_wta(qv, numD, H, W, minD, disp); /* M.cpp:3116-3134 */
    free(costs); free(guide);
    if (!vol) free(qv);
    return rc;
}

/* ---------------------------------------------------------------------------------------
 * Weighted-median ASW, M.cpp:3139-3383 (DISPARITY_LEFT; RIGHT is UB in the reference, B-13)
 * ------------------------------------------------------------------------------------- */

/* colour weight of computeColorWeightGau, M.cpp:3170-3179: exp((d0+d1+d2)/rateR*(-1)) evaluated as
 * addWeighted(d0+d1, -1/rateR, d2, -1/rateR) in f32, then exp.  cv::exp is restated as expf (A-12). */
float orc_wm_color_weight(int d0, int d1, int d2, double rateR)
{
    const float al = (float)((1.0 / rateR) * (-1.0));
    float arg = (float)(d0 + d1) * al + (float)d2 * al;
    return expf(arg);
}

/* space kernel computeSpaceWeightGau, M.cpp:3207-3226 */
void orc_wm_space_kernel(int win, double rateS, float* k)
{
    const int h = win / 2;
    const float al = (float)((1.0 / rateS) * (-1.0));
    for (int y = 0; y < win; y++)
        for (int x = 0; x < win; x++) {
            float v = (float)((x - h) * (x - h)) + (float)((y - h) * (y - h));
            k[x * win + y] = expf(v * al); /* at(x,y): transposed but symmetric, B-14 */
        }
}

typedef struct { float c, w; } cw_t;

/* one weighted-median pick, M.cpp:3276-3304.  cw is overwritten (sorted). */
static float wmedian_pick(cw_t* cw, cw_t* scratch, int n, double halfSum)
{
    /* stable insertion sort by cost == std::multimap<float,float> insertion order */
    (void)scratch;
    for (int i = 1; i < n; i++) {
        cw_t v = cw[i];
        int j = i - 1;
        while (j >= 0 && cw[j].c > v.c) { cw[j + 1] = cw[j]; j--; }
        cw[j + 1] = v;
    }
    double partial = 0.0;
    for (int i = 0; i < n; i++) {
        partial += (double)cw[i].w;
        if (partial > halfSum) return i == 0 ? cw[0].c : cw[i - 1].c;
    }
    return 0.0f; /* never crossed: reference leaves the pixel uninitialised; build value 0 */
}

float orc_wmedian_pick(const float* cost, const float* weight, int n)
{
    cw_t* cw = (cw_t*)malloc((size_t)n * sizeof(cw_t));
    double s = 0;
    for (int i = 0; i < n; i++) { cw[i].c = cost[i]; cw[i].w = weight[i]; s += (double)weight[i]; }
    float r = wmedian_pick(cw, NULL, n, s / 2);
    free(cw);
    return r;
}

int orc_asw_wmedian(const uint8_t* L, const uint8_t* R, int H, int W, int disp_type, int win, double rateS,
                    double rateR, int minD, int numD, float* disp, float* vol)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW; /* M.cpp:3238-3241 */
    if (disp_type != DISPARITY_LEFT) return ORC_ERR_UNSUPPORTED_LAYOUT;
    const int h = win / 2, n = win * win, Hp = H + 2 * h, Wp = W + 2 * h;
    const int max_off = minD + numD - 1, Wb = W + max_off;
    const size_t N = (size_t)H * W;
    float* costs = (float*)malloc((size_t)numD * Hp * Wp * sizeof(float));
    float* wv = vol ? vol : (float*)malloc((size_t)numD * N * sizeof(float));
    float* wd = (float*)malloc((size_t)n * sizeof(float));
    float* wLw = (float*)malloc(N * n * sizeof(float));
    float* wRw = (float*)malloc((size_t)H * Wb * n * sizeof(float));
    if (!costs || !wv || !wd || !wLw || !wRw) { free(costs); if (!vol) free(wv); free(wd); free(wLw); free(wRw); return ORC_ERR_ALLOC; }
    int rc = orc_compute_similarity_padded(L, R, H, W, 3, 0.4, 10, 50, disp_type, win, minD, numD, costs); /* M.cpp:3250 */
    if (rc != ORC_OK) { free(costs); if (!vol) free(wv); free(wd); free(wLw); free(wRw); return rc; }
    orc_wm_space_kernel(win, rateS, wd); /* M.cpp:3255 */

    /* computeColorWeightGau(leftImg): window over the REFLECT(h)-padded image, M.cpp:3156,3165 */
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const uint8_t* c0 = L + ((size_t)y * W + x) * 3;
            for (int j = 0; j < win; j++)
                for (int i = 0; i < win; i++) {
                    const uint8_t* p = L + ((size_t)reflect_idx(y - h + j, H) * W + reflect_idx(x - h + i, W)) * 3;
                    wLw[((size_t)y * W + x) * n + j * win + i] =
                        orc_wm_color_weight(absdiff_u8(p[0], c0[0]), absdiff_u8(p[1], c0[1]), absdiff_u8(p[2], c0[2]), rateR);
                }
        }
    /* computeColorWeightGau(rightImg_border): the image is the (W+max_off)-wide REFLECT-left-padded
     * right image (M.cpp:3246), padded AGAIN by h with REFLECT inside the function (M.cpp:3156). */
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int y = 0; y < H; y++)
        for (int cb = 0; cb < Wb; cb++) {
            const uint8_t* c0 = R + ((size_t)y * W + reflect_idx(cb - max_off, W)) * 3;
            for (int j = 0; j < win; j++)
                for (int i = 0; i < win; i++) {
                    int cc = reflect_idx(cb - h + i, Wb);
                    const uint8_t* p = R + ((size_t)reflect_idx(y - h + j, H) * W + reflect_idx(cc - max_off, W)) * 3;
                    wRw[((size_t)y * Wb + cb) * n + j * win + i] =
                        orc_wm_color_weight(absdiff_u8(p[0], c0[0]), absdiff_u8(p[1], c0[1]), absdiff_u8(p[2], c0[2]), rateR);
                }
        }

    for (int off = 0; off < numD; off++) { /* M.cpp:3264 */
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
        for (int y = 0; y < H; y++) {
            cw_t* cw = (cw_t*)malloc((size_t)n * sizeof(cw_t));
            for (int x = 0; x < W; x++) {
                const float* pl = wLw + ((size_t)y * W + x) * n;
                const float* pr = wRw + ((size_t)y * Wb + (x - off + numD - 1)) * n; /* M.cpp:3274 */
                double sum = 0.0;
                for (int j = 0; j < win; j++)
                    for (int i = 0; i < win; i++) {
                        float w = pl[j * win + i] * wd[j * win + i]; /* .mul(weightDist) */
                        w = w * pr[j * win + i];                     /* .mul(weightWinsR) */
                        cw[j * win + i].w = w;
                        cw[j * win + i].c = costs[((size_t)off * Hp + (y + j)) * Wp + (x + i)]; /* M.cpp:3273 */
                        sum += (double)w;                            /* cv::sum, f64 (A-13) */
                    }
                wv[(size_t)off * N + (size_t)y * W + x] = wmedian_pick(cw, NULL, n, sum / 2);
            }
            free(cw);
        }
    }
    orc_wta(wv, numD, H, W, minD, disp); /* M.cpp:3365-3381 */
    free(costs); free(wd); free(wLw); free(wRw);
    if (!vol) free(wv);
    return ORC_OK;
}


/* ---------------------------------------------------------------------------------------
 * O(1)-bilateral ASW, computeAdaptiveWeight_BLO1, M.cpp:2505-2725
 * Quirks reproduced: weights are |I - k| (a distance, not a similarity); abs(img - k) is folded by
 * MatExpr into absdiff(img, k); the right view is shifted by the loop INDEX i while costs_ds[i] is
 * the cost at disparity minD+i; the normaliser box(M) is taken from the LAST disparity only
 * (M.cpp:2582 is outside the i-loop) and divides every slice; the interpolation weights are
 * swapped ((cur-lower) multiplies the LOWER slice, M.cpp:2659-2660).  The WTA indexes the per-key
 * vectors with the ABSOLUTE offset (M.cpp:2659), which is out of range unless minDisparity == 0:
 * only minDisparity == 0 is accepted.
 * ------------------------------------------------------------------------------------- */
int orc_asw_blo1(const uint8_t* L, const uint8_t* R, int H, int W, int disp_type, double sampleRateR, int win, int minD,
                 int numD, float* disp, float* vol)
{
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW; /* getCostSAD_d returns Mat(), M.cpp:2458-2462 */
    if (minD != 0) return 7;                       /* see header: UB in the reference */
    const int step = (int)(256 * sampleRateR);     /* M.cpp:2550 */
    if (step <= 0) return 7;                       /* the reference's key loop would never terminate */
    const size_t N = (size_t)H * W;
    int keys[257], nk = 0;
    for (int i = 0; i < 256; i += step) keys[nk++] = i; /* M.cpp:2551-2556 */
    if (keys[nk - 1] != 255) keys[nk++] = 255;          /* M.cpp:2557-2560 */
    uint8_t* gl = (uint8_t*)malloc(N);
    uint8_t* gr = (uint8_t*)malloc(N);
    float* costs = (float*)malloc((size_t)numD * N * sizeof(float));
    float* JB = (float*)malloc((size_t)nk * numD * N * sizeof(float));
    if (!gl || !gr || !costs || !JB) { free(gl); free(gr); free(costs); free(JB); return ORC_ERR_ALLOC; }
    orc_bgr2gray(L, H, W, gl);
    orc_bgr2gray(R, H, W, gr);
    int rc = orc_cost_sad(L, R, H, W, disp_type, win, minD, numD, costs); /* M.cpp:2529-2547 */
    if (rc != ORC_OK) { free(gl); free(gr); free(costs); free(JB); return rc; }
    int key_index[256];
    for (int i = 0; i < 256; i++) key_index[i] = -1;
    for (int ki = 0; ki < nk; ki++) key_index[keys[ki]] = ki;

#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int ki = 0; ki < nk; ki++) {
        const int k = keys[ki];
        float* M = (float*)malloc(N * sizeof(float));
        float* J = (float*)malloc(N * sizeof(float));
        float* bM = (float*)malloc(N * sizeof(float));
        for (int i = 0; i < numD; i++) {
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    int a, b;
                    if (disp_type == DISPARITY_LEFT) { /* M.cpp:2571-2580 */
                        a = absdiff_u8(gl[(size_t)y * W + x], k);
                        b = absdiff_u8(gr[(size_t)y * W + reflect_idx(x - i, W)], k);
                    } else { /* M.cpp:2599-2608 */
                        b = absdiff_u8(gr[(size_t)y * W + x], k);
                        a = absdiff_u8(gl[(size_t)y * W + reflect_idx(x + i, W)], k);
                    }
                    float m = (float)b * (float)a; /* M_k_y_r.mul(M_k_y_l) */
                    M[(size_t)y * W + x] = m;
                    J[(size_t)y * W + x] = m * costs[(size_t)i * N + (size_t)y * W + x];
                }
            box_filter_plane(J, 1, JB + ((size_t)ki * numD + i) * N, 1, H, W, win); /* M.cpp:2578 */
        }
        box_filter_plane(M, 1, bM, 1, H, W, win); /* M.cpp:2582: M of the LAST disparity */
        for (int i = 0; i < numD; i++) {
            float* jb = JB + ((size_t)ki * numD + i) * N;
            for (size_t p = 0; p < N; p++) jb[p] = jb[p] / bM[p]; /* M.cpp:2588, IEEE f32 division */
        }
        free(M); free(J); free(bM);
    }

    const uint8_t* ref = disp_type == DISPARITY_LEFT ? gl : gr;
    for (size_t p = 0; p < N; p++) {
        double best = DBL_MAX;
        float bd = 0.0f;
        const int cur = ref[p];
        for (int off = 0; off < numD; off++) { /* min_offset..max_offset with minD == 0 */
            double c;
            if (key_index[cur] < 0) { /* M.cpp:2650-2661 */
                int lower = cur / step * step, upper = lower + step;
                if (upper > 255) upper = 255;
                float lo = JB[((size_t)key_index[lower] * numD + off) * N + p];
                float hi = JB[((size_t)key_index[upper] * numD + off) * N + p];
                float t = (float)(cur - lower) * lo + (float)(upper - cur) * hi;
                c = (double)t;
            } else {
                c = (double)JB[((size_t)key_index[cur] * numD + off) * N + p];
            }
            if (vol) vol[(size_t)off * N + p] = (float)c;
            if (c < best) { best = c; bd = (float)off; }
        }
        disp[p] = bd;
    }
    free(gl); free(gr); free(costs); free(JB);
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------
 * computeAdaptiveWeight_direct8, M.cpp:1167-1319.
 * The classic scheme on a sparse support: only the taps with i==j, i==0, j==0 or i+j==ks-1 (M.cpp:1201, 1245) -- the
 * last test was meant to be the anti-diagonal but only ever matches (h,h), so the support is row + column + main
 * diagonal = 3*(ks-1) taps.  gamma_c = 30, gamma_g = winSize*2/3 in INTEGER arithmetic (M.cpp:1175), k = 3.
 * Weight maps are built and consumed in the same order (no transposition here, unlike the classic method).
 * Range inclusive: minD..minD+numD (M.cpp:1171,1223).  DISPARITY_RIGHT indexes the weight vectors with the signed
 * tap coordinate i (M.cpp:1291-1295) -- undefined behaviour in the reference -> ORC_ERR_UNSUPPORTED_LAYOUT here.
 * vol (optional): (float)E, numD+1 planes.
 * ------------------------------------------------------------------------------------- */
int orc_asw_direct8(const uint8_t* Lbgr, const uint8_t* Rbgr, int H, int W, int disp_type, int win, int minD, int numD,
                    float* disp, float* vol)
{
    if (disp_type != DISPARITY_LEFT) return ORC_ERR_UNSUPPORTED_LAYOUT;
    if (win % 2 == 0) return ORC_ERR_EVEN_WINDOW; /* build decision, as for the classic method */
    const int ks = win, h = ks / 2;
    const int max_offset = minD + numD, min_offset = minD;
    const double k = 3, gamma_c = 30, gamma_g = (double)(win * 2 / 3);
    int nt = 0;
    int* ti = (int*)malloc(sizeof(int) * (size_t)(ks * ks + 1));
    int* tj = (int*)malloc(sizeof(int) * (size_t)(ks * ks + 1));
    for (int j = -h; j < h + 1; j++)
        for (int i = -h; i < h + 1; i++) {
            if (i == 0 && j == 0) continue;
            if (i == j || i == 0 || j == 0 || (i + j) == ks - 1) { ti[nt] = i; tj[nt] = j; nt++; }
        }
    /* weight of tap t for colour distance dc: the map entry of M.cpp:1214-1215 as a function of (t, dc) */
    float* lut = (float*)malloc(sizeof(float) * 256 * (size_t)(nt > 0 ? nt : 1));
    for (int t = 0; t < nt; t++) {
        double delta_g = sqrt((double)(ti[t] * ti[t] + tj[t] * tj[t]));
        for (int dc = 0; dc < 256; dc++) lut[t * 256 + dc] = (float)(k * exp(-((double)dc / gamma_c + delta_g / gamma_g)));
    }
    uint8_t* left = (uint8_t*)malloc((size_t)H * W);
    uint8_t* right = (uint8_t*)malloc((size_t)H * W);
    orc_bgr2gray(Lbgr, H, W, left);
    orc_bgr2gray(Rbgr, H, W, right);
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            double best = DBL_MAX;
            float bd = 0.0f; /* never-written pixels: 0 (the reference leaves them uninitialised) */
            for (int offset = min_offset; offset <= max_offset; offset++) {
                double numerator = 0, denominator = 0;
                const int xr = imax(0, x - offset);
                for (int t = 0; t < nt && t < 4 * (ks - 1); t++) {
                    const int i = ti[t], j = tj[t];
                    const int nx = imin(imax(0, x + i), W - 1), ny = imin(imax(0, y + j), H - 1);
                    const int nxr = imin(imax(0, xr + i), W - 1);
                    float a = lut[t * 256 + absdiff_u8(left[(size_t)ny * W + nx], left[(size_t)y * W + x])];
                    float b = lut[t * 256 + absdiff_u8(right[(size_t)ny * W + nxr], right[(size_t)y * W + xr])];
                    float ab = a * b;
                    numerator += ab * fabs((double)(left[(size_t)ny * W + nx] - right[(size_t)ny * W + imax(0, nx - offset)]));
                    denominator += ab;
                }
                double E = numerator / denominator;
                if (vol) vol[((size_t)(offset - min_offset) * H + y) * W + x] = (float)E;
                if (E < best) { best = E; bd = (float)offset; }
            }
            disp[(size_t)y * W + x] = bd;
        }
    free(ti); free(tj); free(lut); free(left); free(right);
    return ORC_OK;
}

/* Left-right consistency check (the build's own rule, include/asw_mi355x.h asw_lr_check; the reference has no such step):
 * keep dl(y,x) when the right map agrees at x - (int)dl within max_diff, else `invalid`.  Returns the number rejected. */
int orc_lr_check(const float* dl, const float* dr, int H, int W, float max_diff, float invalid, float* out)
{
    int bad = 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float d = dl[(size_t)y * W + x];
            const int sane = fabsf(d) < 16777216.0f; /* NaN, inf, absurd values: rejected before the int cast */
            const int xr = x - (sane ? (int)d : 0);
            const int ok = sane && xr >= 0 && xr < W && fabsf(d - dr[(size_t)y * W + (xr < 0 ? 0 : xr >= W ? W - 1 : xr)]) <= max_diff;
            out[(size_t)y * W + x] = ok ? d : invalid;
            bad += !ok;
        }
    return bad;
}

/* ---------------------------------------------------------------------------------------
 * Bilateral-grid ASW: computeAdaptiveWeight_bilateralGrid, M.cpp:2253-2430 (enum 5, called with sampleRateS =
 * sampleRateR = 10 at M.cpp:67), grid builder createBilGrid M.cpp:1831-2185, quadrlinear_blGrid M.cpp:2227-2251.
 *
 * Per candidate offset the reference rebuilds a 4-D grid bilGrid[x][y][zL][zR] of pair<double,int> in nested
 * std::maps: every pixel adds |gL - gR_shifted| and a count of 1 at its rounded (cvRound = half-to-even) grid key,
 * four IN-PLACE 5-tap passes (axes zR, zL, y, x, in that order; each pass walks its axis in ascending order and
 * overwrites as it goes, so taps -1/-2 see already-filtered values; the count is a C int, i.e. every assignment
 * truncates it) smooth the grid, and a pixel's cost is quadrilinear(first) / quadrilinear(second) over the 16
 * neighbours at key +-1 around cvCeil(coordinate / rate).  Keys outside the allocated 0..gridSize range are read
 * through std::map::operator[], which inserts (0.0, 0): reads outside the grid are zeros (App. B, inventory #12).
 * Here the grid is a dense array with exactly that zero rule.
 * DISPARITY_RIGHT reads column `width` of the left image (min(x + offset, width), M.cpp:1929,2356: one past the
 * row) -- undefined behaviour in the reference -> ORC_ERR_UNSUPPORTED_LAYOUT here.
 * vol (optional): (float)cost, numD+1 planes (NaN where the interpolated count is 0).
 * ------------------------------------------------------------------------------------- */
typedef struct {
    double* f; /* .first  */
    int* s;    /* .second */
    int nx, ny, nl, nr; /* gridSize_width/height/rangeL/rangeR: LAST valid index of each axis */
} blgrid_t;

static inline size_t blg_at(const blgrid_t* g, int x, int y, int l, int r)
{
    return (((size_t)x * (size_t)(g->ny + 1) + (size_t)y) * (size_t)(g->nl + 1) + (size_t)l) * (size_t)(g->nr + 1) + (size_t)r;
}
static inline int blg_in(const blgrid_t* g, int x, int y, int l, int r)
{
    return x >= 0 && x <= g->nx && y >= 0 && y <= g->ny && l >= 0 && l <= g->nl && r >= 0 && r <= g->nr;
}

/* one line of one smoothing pass, in place, ascending (M.cpp:1936-1993 and its three repetitions) */
static void blg_pass_line(double* f, int* s, size_t stride, int n)
{
#define BF(i) ((i) <= n ? f[(size_t)(i) * stride] : 0.0)
#define BS(i) ((i) <= n ? (double)s[(size_t)(i) * stride] : 0.0)
    for (int w = 0; w <= n; w++) {
        double rf, rs;
        if (w == 0) {
            rf = 0.6 * BF(w) + 0.3 * BF(w + 1) + 0.1 * BF(w + 2);
            rs = 0.6 * BS(w) + 0.3 * BS(w + 1) + 0.1 * BS(w + 2);
        } else if (w == 1) {
            rf = 0.2 * BF(w - 1) + 0.5 * BF(w) + 0.2 * BF(w + 1) + 0.1 * BF(w + 2);
            rs = 0.2 * BS(w - 1) + 0.5 * BS(w) + 0.2 * BS(w + 1) + 0.1 * BS(w + 2);
        } else if (w == n - 1) {
            rf = 0.1 * BF(w - 2) + 0.2 * BF(w - 1) + 0.5 * BF(w) + 0.2 * BF(w + 1);
            rs = 0.1 * BS(w - 2) + 0.2 * BS(w - 1) + 0.5 * BS(w) + 0.2 * BS(w + 1);
        } else if (w == n) {
            rf = 0.1 * BF(w - 2) + 0.3 * BF(w - 1) + 0.6 * BF(w);
            rs = 0.1 * BS(w - 2) + 0.3 * BS(w - 1) + 0.6 * BS(w);
        } else {
            rf = 0.0625 * BF(w - 2) + 0.25 * BF(w - 1) + 0.375 * BF(w) + 0.25 * BF(w + 1) + 0.0625 * BF(w + 2);
            rs = 0.0625 * BS(w - 2) + 0.25 * BS(w - 1) + 0.375 * BS(w) + 0.25 * BS(w + 1) + 0.0625 * BS(w + 2);
        }
        f[(size_t)w * stride] = rf;
        s[(size_t)w * stride] = (int)rs; /* pair<double,double> -> pair<double,int> */
    }
#undef BF
#undef BS
}

/* quadrlinear_blGrid, M.cpp:2227-2251: d = {x,y,zL,zR} fractions, n[16] with x the slowest and zR the fastest bit */
static double blg_quadrilinear(const double d[4], const double n[16])
{
    double a[8], b[4];
    for (int i = 0; i < 8; i++) a[i] = n[2 * i] * (1 - d[3]) + n[2 * i + 1] * d[3];
    for (int i = 0; i < 4; i++) b[i] = a[2 * i] * (1 - d[2]) + a[2 * i + 1] * d[2];
    const double c1 = b[0] * (1 - d[1]) + b[1] * d[1];
    const double c2 = b[2] * (1 - d[1]) + b[3] * d[1];
    return c1 * (1 - d[0]) + c2 * d[0];
}

static inline int cv_round_d(double v) { return (int)lrint(v); } /* cvRound: to nearest, ties to even */
static inline int cv_ceil_d(double v) { int i = (int)v; return i + ((double)i < v); }

int orc_asw_bilgrid(const uint8_t* Lbgr, const uint8_t* Rbgr, int H, int W, int disp_type, double sampleRateS,
                    double sampleRateR, int minD, int numD, float* disp, float* vol)
{
    if (disp_type != DISPARITY_LEFT) return ORC_ERR_UNSUPPORTED_LAYOUT;
    if (!(sampleRateS > 0) || !(sampleRateR > 0)) return ORC_ERR_UNSUPPORTED_LAYOUT; /* the slicing divides by them */
    const int max_offset = minD + numD, min_offset = minD;
    uint8_t* left = (uint8_t*)malloc((size_t)H * W);
    uint8_t* right = (uint8_t*)malloc((size_t)H * W);
    orc_bgr2gray(Lbgr, H, W, left);   /* M.cpp:2271-2278 */
    orc_bgr2gray(Rbgr, H, W, right);
    blgrid_t g;
    g.nl = cv_round_d(255.0 / sampleRateR); /* M.cpp:1868-1871 */
    g.nr = cv_round_d(255.0 / sampleRateR);
    g.nx = cv_round_d((W - 1) / sampleRateS);
    g.ny = cv_round_d((H - 1) / sampleRateS);
    const size_t cells = (size_t)(g.nx + 1) * (size_t)(g.ny + 1) * (size_t)(g.nl + 1) * (size_t)(g.nr + 1);
    g.f = (double*)malloc(cells * sizeof(double));
    g.s = (int*)malloc(cells * sizeof(int));
    double* best = (double*)malloc((size_t)H * W * sizeof(double));
    for (size_t i = 0; i < (size_t)H * W; i++) { best[i] = DBL_MAX; disp[i] = 0.0f; } /* never-written pixels: 0 */
    const size_t sr = 1, sl = (size_t)(g.nr + 1), sy = sl * (size_t)(g.nl + 1), sx = sy * (size_t)(g.ny + 1);

    for (int offset = min_offset; offset <= max_offset; offset++) {
        memset(g.f, 0, cells * sizeof(double));
        memset(g.s, 0, cells * sizeof(int));
        /* grid filling, M.cpp:1897-1915 (sums of integers: exact, any order) */
        for (int i = 0; i < W; i++)
            for (int j = 0; j < H; j++) {
                const float vl = (float)left[(size_t)j * W + i], vr = (float)right[(size_t)j * W + imax(0, i - offset)];
                const size_t c = blg_at(&g, cv_round_d(i / sampleRateS), cv_round_d(j / sampleRateS),
                                        cv_round_d(vl / sampleRateR), cv_round_d(vr / sampleRateR));
                g.f[c] = g.f[c] + fabsf(vl - vr);
                g.s[c] = g.s[c] + 1;
            }
        /* the four passes, in the reference's order: zR, zL, y, x */
#pragma omp parallel for collapse(2) schedule(static) num_threads(g_threads)
        for (int x = 0; x <= g.nx; x++)
            for (int y = 0; y <= g.ny; y++) {
                for (int l = 0; l <= g.nl; l++) blg_pass_line(g.f + blg_at(&g, x, y, l, 0), g.s + blg_at(&g, x, y, l, 0), sr, g.nr);
                for (int r = 0; r <= g.nr; r++) blg_pass_line(g.f + blg_at(&g, x, y, 0, r), g.s + blg_at(&g, x, y, 0, r), sl, g.nl);
            }
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int x = 0; x <= g.nx; x++)
            for (size_t lr = 0; lr < sy; lr++) blg_pass_line(g.f + (size_t)x * sx + lr, g.s + (size_t)x * sx + lr, sy, g.ny);
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (long long ylr = 0; ylr < (long long)sx; ylr++) blg_pass_line(g.f + ylr, g.s + ylr, sx, g.nx);

        /* slicing + WTA, M.cpp:2284-2350 */
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const double c[4] = {x / sampleRateS, y / sampleRateS, left[(size_t)y * W + x] / sampleRateR,
                                     right[(size_t)y * W + imax(0, x - offset)] / sampleRateR};
                int k[4];
                double d[4];
                for (int a = 0; a < 4; a++) { k[a] = cv_ceil_d(c[a]); d[a] = k[a] - c[a]; }
                double nf[16], ns[16];
                for (int n = 0; n < 16; n++) {
                    const int gx = k[0] + ((n & 8) ? 1 : -1), gy = k[1] + ((n & 4) ? 1 : -1);
                    const int gl = k[2] + ((n & 2) ? 1 : -1), gr = k[3] + ((n & 1) ? 1 : -1);
                    const int in = blg_in(&g, gx, gy, gl, gr);
                    nf[n] = in ? g.f[blg_at(&g, gx, gy, gl, gr)] : 0.0;
                    ns[n] = in ? (double)g.s[blg_at(&g, gx, gy, gl, gr)] : 0.0;
                }
                const double cur = blg_quadrilinear(d, nf) / blg_quadrilinear(d, ns);
                if (vol) vol[((size_t)(offset - min_offset) * H + y) * W + x] = (float)cur;
                if (cur < best[(size_t)y * W + x]) { best[(size_t)y * W + x] = cur; disp[(size_t)y * W + x] = (float)offset; }
            }
    }
    free(left); free(right); free(g.f); free(g.s); free(best);
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------
 * Driver-side pre/post-processing, aswStereoMatch.cpp ("main.cpp") :30-31, 67-89, 97-98.
 * Five more OpenCV 4.1.0 primitives, restated from their portable C++ paths; none can be verified offline (no OpenCV,
 * no fixture): resize(INTER_LINEAR) on 8UC3 with 11-bit fixed-point coefficients (and its silent switch to INTER_AREA for
 * an exact 2x downscale), cvtColor BGR2HSV / HSV2BGR on 8U (H in [0,180)), bilateralFilter on 8UC1 (f32 accumulation in
 * tap order, BORDER_REFLECT), saturating u8 MatExpr arithmetic, convertTo(CV_8U) + normalize(0,255,NORM_MINMAX).
 * Builds with IPP (the reference's Windows OpenCV) may take other code paths for resize and bilateralFilter.
 * ------------------------------------------------------------------------------------- */
static inline int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline int cv_round_f(float v) { return (int)lrintf(v); }   /* cvRound: to nearest, ties to even */
static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }

/* resize(src, dst, Size(dw, dh)) with the default INTER_LINEAR, main.cpp:30-31 */
void orc_resize_linear_u8c3(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw)
{
    const int cn = 3;
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double scale_x = 1.0 / inv_x, scale_y = 1.0 / inv_y;
    /* cv::resize: INTER_LINEAR with an exact 2x2 integer downscale is executed as INTER_AREA (fast 2x2 average) */
    {
        int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y); /* saturate_cast<int>(scale) */
        int fast = fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON;
        if (fast && isx == 2 && isy == 2) {
            for (int y = 0; y < dh; y++)
                for (int x = 0; x < dw; x++)
                    for (int c = 0; c < cn; c++) {
                        const uint8_t* p = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
                        dst[((size_t)y * dw + x) * cn + c] = (uint8_t)((p[0] + p[cn] + p[(size_t)sw * cn] + p[(size_t)sw * cn + cn] + 2) >> 2);
                    }
            return;
        }
    }
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* ialpha = (short*)malloc(sizeof(short) * dw * 2);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { if (dx < xmax) xmax = dx; if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
        xofs[dx] = sx;
        int a0 = cv_round_f((1.f - fx) * 2048.f), a1 = cv_round_f(fx * 2048.f);
        ialpha[dx * 2] = (short)a0; ialpha[dx * 2 + 1] = (short)a1;
    }
    int* rows = (int*)malloc(sizeof(int) * (size_t)dw * cn * 2);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        /* no reset of fy at the borders in y: resizeGeneric_ clips the two source ROWS to the image instead */
        int b0 = cv_round_f((1.f - fy) * 2048.f), b1 = cv_round_f(fy * 2048.f);
        for (int k = 0; k < 2; k++) {
            int yy = sy + k; if (yy < 0) yy = 0; if (yy > sh - 1) yy = sh - 1;
            const uint8_t* S = src + (size_t)yy * sw * cn;
            int* D = rows + (size_t)k * dw * cn;
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    int sx = xofs[dx] * cn + c;
                    D[dx * cn + c] = dx < xmax ? S[sx] * ialpha[dx * 2] + S[sx + cn] * ialpha[dx * 2 + 1] : S[sx] * 2048;
                }
        }
        const int *S0 = rows, *S1 = rows + (size_t)dw * cn;
        for (int x = 0; x < dw * cn; x++) /* VResizeLinear<uchar,int,short,...>: the truncating 8u form */
            dst[(size_t)dy * dw * cn + x] = (uint8_t)((((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(rows);
}

/* cvtColor(COLOR_BGR2HSV) on 8U, main.cpp:70-71: V = max, S = diff*255/V, H = 30*sector arithmetic, 12-bit tables */
void orc_bgr2hsv_u8(const uint8_t* bgr, size_t n, uint8_t* hsv)
{
    const int hsv_shift = 12;
    static int sdiv[256], hdiv[256], init = 0;
    if (!init) {
        sdiv[0] = hdiv[0] = 0;
        for (int i = 1; i < 256; i++) {
            sdiv[i] = (int)lrint((255 << hsv_shift) / (1. * i));
            hdiv[i] = (int)lrint((180 << hsv_shift) / (6. * i));
        }
        init = 1;
    }
    for (size_t i = 0; i < n; i++) {
        int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        int v = b, vmin = b;
        if (g > v) v = g;
        if (r > v) v = r;
        if (g < vmin) vmin = g;
        if (r < vmin) vmin = r;
        int diff = v - vmin;
        int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
        int s = (diff * sdiv[v] + (1 << (hsv_shift - 1))) >> hsv_shift;
        int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
        h = (h * hdiv[diff] + (1 << (hsv_shift - 1))) >> hsv_shift;
        h += h < 0 ? 180 : 0;
        hsv[3 * i] = (uint8_t)sat_u8(h); hsv[3 * i + 1] = (uint8_t)s; hsv[3 * i + 2] = (uint8_t)v;
    }
}

/* cvtColor(COLOR_HSV2BGR) on 8U, main.cpp:80,88: through f32 (HSV2RGB_f), *255, cvRound */
void orc_hsv2bgr_u8(const uint8_t* hsv, size_t n, uint8_t* bgr)
{
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    const float hscale = 6.f / 180.f;
    for (size_t i = 0; i < n; i++) {
        float h = (float)hsv[3 * i], s = hsv[3 * i + 1] * (1.f / 255.f), v = hsv[3 * i + 2] * (1.f / 255.f);
        float b, g, r;
        if (s == 0) b = g = r = v;
        else {
            float tab[4];
            h *= hscale;
            if (h < 0) do h += 6; while (h < 0);
            else if (h >= 6) do h -= 6; while (h >= 6);
            int sector = cv_floor_f(h);
            h -= sector;
            if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
            tab[0] = v;
            tab[1] = v * (1.f - s);
            tab[2] = v * (1.f - s * h);
            tab[3] = v * (1.f - s * (1.f - h));
            b = tab[sector_data[sector][0]]; g = tab[sector_data[sector][1]]; r = tab[sector_data[sector][2]];
        }
        bgr[3 * i] = (uint8_t)sat_u8(cv_round_f(b * 255.f));
        bgr[3 * i + 1] = (uint8_t)sat_u8(cv_round_f(g * 255.f));
        bgr[3 * i + 2] = (uint8_t)sat_u8(cv_round_f(r * 255.f));
    }
}

/* taps of bilateralFilter(d, sigmaColor, sigmaSpace): circular support, raster order; returns the number of taps */
int orc_bilateral_taps(int d, double sigma_space, int* dy, int* dx, float* w)
{
    if (sigma_space <= 0) sigma_space = 1;
    int radius = d <= 0 ? (int)lrint(sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    const double gs = -0.5 / (sigma_space * sigma_space);
    int n = 0;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            dy[n] = i; dx[n] = j; w[n] = (float)exp(r * r * gs);
            n++;
        }
    return n;
}
void orc_bilateral_color_lut(double sigma_color, float* lut /* 256 */)
{
    if (sigma_color <= 0) sigma_color = 1;
    const double gc = -0.5 / (sigma_color * sigma_color);
    for (int i = 0; i < 256; i++) lut[i] = (float)exp((double)i * i * gc);
}

/* bilateralFilter(src 8UC1, dst, d, sigmaColor, sigmaSpace, BORDER_REFLECT), main.cpp:76,84 */
void orc_bilateral_u8c1(const uint8_t* src, int H, int W, int d, double sigma_color, double sigma_space, uint8_t* dst)
{
    int dy[1024], dx[1024];
    float sw[1024], cw[256];
    const int nt = orc_bilateral_taps(d, sigma_space, dy, dx, sw);
    orc_bilateral_color_lut(sigma_color, cw);
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float sum = 0, wsum = 0;
            const int val0 = src[(size_t)y * W + x];
            for (int k = 0; k < nt; k++) {
                const int val = src[(size_t)reflect_idx(y + dy[k], H) * W + reflect_idx(x + dx[k], W)];
                const float w = sw[k] * cw[val > val0 ? val - val0 : val0 - val];
                sum += val * w;
                wsum += w;
            }
            dst[(size_t)y * W + x] = (uint8_t)cv_round_f(sum / wsum);
        }
}

/* the HSV-V detail boost of main.cpp:67-89 on one BGR image: V += 2 * (V - bilateral(V, 7, 10, 3)) with u8 saturation */
void orc_detail_boost(const uint8_t* bgr, int H, int W, uint8_t* out)
{
    const size_t N = (size_t)H * W;
    uint8_t* hsv = (uint8_t*)malloc(N * 3);
    uint8_t* v = (uint8_t*)malloc(N);
    uint8_t* blur = (uint8_t*)malloc(N);
    orc_bgr2hsv_u8(bgr, N, hsv);
    for (size_t i = 0; i < N; i++) v[i] = hsv[3 * i + 2];
    orc_bilateral_u8c1(v, H, W, 7, 10, 3, blur);
    for (size_t i = 0; i < N; i++) {
        int detail = sat_u8(v[i] - blur[i]);                 /* mat3cn[2] - blurL */
        hsv[3 * i + 2] = (uint8_t)sat_u8(v[i] + 2 * detail); /* mat3cn[2] + detailL * 2 (addWeighted: exact integers) */
    }
    orc_hsv2bgr_u8(hsv, N, out);
    free(hsv); free(v); free(blur);
}

/* main.cpp:30-31 + 67-89 for one image */
void orc_preprocess(const uint8_t* src, int sh, int sw, int dh, int dw, int boost, uint8_t* out)
{
    uint8_t* small = (uint8_t*)malloc((size_t)dh * dw * 3);
    orc_resize_linear_u8c3(src, sh, sw, small, dh, dw);
    if (boost) orc_detail_boost(small, dh, dw, out);
    else memcpy(out, small, (size_t)dh * dw * 3);
    free(small);
}

/* main.cpp:97-98: convertTo(CV_8UC1), then (normalize != 0) normalize(0, 255, NORM_MINMAX) */
void orc_disparity_to_u8(const float* disp, size_t n, int normalize, uint8_t* out)
{
    int mn = 255, mx = 0;
    for (size_t i = 0; i < n; i++) {
        int v = sat_u8(cv_round_f(disp[i]));
        out[i] = (uint8_t)v;
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
    if (!normalize) return;
    double scale = 255.0 * ((double)(mx - mn) > DBL_EPSILON ? 1.0 / (double)(mx - mn) : 0.0);
    double shift = 0.0 - (double)mn * scale;
    const float fs = (float)scale, fb = (float)shift;
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)sat_u8(cv_round_f(out[i] * fs + fb));
}

/* ---------------------------------------------------------------------------------------
 * stereoMatching selector, M.cpp:46-88, with the literals it hard-codes.
 * ------------------------------------------------------------------------------------- */
int orc_stereo_matching(const uint8_t* L, const uint8_t* R, int H, int W, int disparity_type, int algorithm, int win,
                        int minD, int numD, float* disp)
{
    switch (algorithm) {
    case 2: return orc_asw_classic(L, R, H, W, 30, 20, disparity_type, win, minD, numD, disp, NULL);  /* M.cpp:58 */
    case 3: return orc_asw_direct8(L, R, H, W, disparity_type, win, minD, numD, disp, NULL);          /* M.cpp:61 */
    case 4: return orc_asw_geodesic(L, R, H, W, disparity_type, win, minD, numD, disp, NULL);          /* M.cpp:64 */
    case 5: return orc_asw_bilgrid(L, R, H, W, disparity_type, 10, 10, minD, numD, disp, NULL);        /* M.cpp:67 */
    case 6: return orc_asw_blo1(L, R, H, W, disparity_type, 0.015, win, minD, numD, disp, NULL);       /* M.cpp:70 */
    case 7: return orc_asw_guided(L, R, H, W, disparity_type, 1e-6, win, minD, numD, disp, NULL);      /* M.cpp:73 */
    case 8: return orc_asw_guided2(L, R, H, W, disparity_type, 1e-6, win, minD, numD, disp, NULL);     /* M.cpp:76 */
    case 9: return orc_asw_guided3(L, R, H, W, disparity_type, 1e-6, win, minD, numD, disp, NULL);     /* M.cpp:79 */
    case 10: return orc_asw_wmedian(L, R, H, W, disparity_type, win, 10, 10, minD, numD, disp, NULL);  /* M.cpp:82 */
    case 11: return orc_ncc_disparity(L, R, H, W, disparity_type, win, minD, numD, disp);              /* M.cpp:85 */
    default: return 3; /* unsupported method (out of scope) */
    }
}
