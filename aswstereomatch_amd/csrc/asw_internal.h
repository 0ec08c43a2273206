// Internal declarations shared by the HIP translation units of libasw_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/asw_mi355x.h"

#define ASW_HIP_TRY(expr)                                  \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) {                            \
            asw_note_hip_error(_e, #expr, __FILE__, __LINE__); \
            return ASW_ERR_HIP;                            \
        }                                                  \
    } while (0)

void asw_note_hip_error(hipError_t e, const char* what, const char* file, int line);

// Grow-only device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Frame {
    DevBuf L, R;  // dense interleaved 8U images as uploaded
    int rows = 0, cols = 0, channels = 0;
    DevBuf disp;  // f32 rows x cols
    DevBuf vol;   // f32 aggregated cost volume of the last match (if kept)
    size_t vol_floats = 0;
    bool valid = false;
    // the disparity (and volume) in `disp` / `vol` belong to the CURRENT pair: set by a successful match, cleared by every
    // upload / pre-processing into the slot and by a failed match, so a download can never return another frame's result
    // or read a smaller, older allocation
    bool has_disp = false;
    int disp_rows = 0, disp_cols = 0;
    void invalidate_results() { has_disp = false; disp_rows = disp_cols = 0; vol_floats = 0; }
};

struct BilateralTables {  // cached per (kind, win, gamma_c, gamma_g, mirror)
    int kind = -1;  // 0: classic (all win*win-1 taps, transposed consume order), 1: direct8 (row + column + diagonal)
    int win = 0;
    int mirror = -1;
    double gamma_c = 0, gamma_g = 0;
    int ntaps = 0, ncls = 0;
    DevBuf taps;  // int4 per tap
    DevBuf lut;   // float [ncls][256]
    DevBuf cells; // kind 0, win 15, not mirrored: int4 per window cell (kx = -3..17, ky = 0..14) for k_asw_bilateral_xq
};

// Measurement / test switches, read from the environment ONCE when a context is created (asw_create): a call never looks at
// the environment.  -1 / 0 = the library's own choice.
struct AswTuning {
    int bilateral_xq = -1;       // ASW_BILATERAL_XQ: 0 = one-kernel form only (k_asw_bilateral), 1 = xq form wherever it applies
    int geodesic_xq = -1;        // ASW_GEODESIC_XQ
    int wmedian_tile = -1;       // ASW_WMEDIAN_TILE: 0 = per-pixel sort (k_wmedian)
    int wmedian_tile_chunk = 0;  // ASW_WMEDIAN_TILE_CHUNK: slices per list chunk (tests: odd chunkings)
    int wmedian_tile_split = 0;  // ASW_WMEDIAN_TILE_SPLIT: workgroups per pixel block
    int wmedian_gen_rows = 0;    // ASW_WMEDIAN_GEN_ROWS: block rows per workgroup of the general tile form (windows 17..37): 1 | 2 | 4 | 8
    int band_ab = 0, band_q = 0; // ASW_BAND_AB / ASW_BAND_Q: rows per band of the guided filter's passes
    int ring_ab = 1, ring_q = 1; // ASW_RING_AB / ASW_RING_Q: register-ring form of the two passes (k_guided.hip: launch_guided3), 0 = re-fetch
    int guided_fused = -1;       // ASW_GUIDED_FUSED: fused a/b -> q walk of GuidedF_2 at 15x15 (k_guided_pair3: no a/b volume, a third of the traffic): 1 always, 0 never, -1 = frames large enough for it (guided_uses_fused)
    int ab6_pair = -1;           // ASW_AB6_PAIR: 0 = the a/b pass of the 6-channel guide through k_box_walk instead of k_ab6_pair
    int q6_pair = -1;            // ASW_Q6_PAIR: 0 = the q pass of the 6-channel guide through k_box_walk (re-fetching form) instead of k_q6_pair
    int q_wg_strips = 1;         // ASW_Q_WG_STRIPS: workgroup of the q pass = 4 neighbouring strips (1) / 4 slices of a strip (0)
    void read_environment();
};

struct asw_ctx {
    int device = 0;
    AswTuning tune;
    int prep_ntaps = 0;  // taps of the pre-processing bilateral filter (tables in buf("prep_tables"))
    int gray_bits = 14;  // cvtColor(BGR2GRAY) constant set (asw_set_gray_bits): 14 = OpenCV 4.1.0, 15 = later 4.x
    hipStream_t stream = nullptr;
    std::vector<Frame> frames;  // resident slots of asw_upload_pair / asw_match_resident (caller-numbered)
    Frame host_frame;           // private frame of the host-buffer entry points (asw_stereo_match, asw_aggregate_*): never a slot
    std::vector<unsigned char> host_pack;  // dense staging for host images whose rows carry padding (asw_context.hip: copy_rows)
    std::map<std::string, DevBuf> scratch;  // named grow-only scratch buffers
    BilateralTables bil;
    // weighted-median tables: exp() LUT of the colour weight per rateR, space kernel per (win, rateS)
    DevBuf wm_lut2, wm_wd;
    double wm_rate_r = -1, wm_rate_s = -1;
    int wm_win = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // total start/stop, aggregate start/stop
    // side streams for small launches that are independent of a method's main kernel (border tiles and the one-candidate tail
    // of the classic bilateral method: 0.5 ms each when serialised, latency-bound) + fork / join events
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t aux_ev[3] = {nullptr, nullptr, nullptr};
    asw_timing timing = {0, 0, 0, 0};
    DevBuf& buf(const char* name) { return scratch[name]; }
};

// ---- kernel launchers (each returns an asw_status; all work is enqueued on `s`) ----
// bits: fixed-point width of the BT.601 constants, 14 (OpenCV 4.1.0, the reference's pin) or 15 (later 4.x): SURVEY App. A-1
int launch_bgr2gray(hipStream_t s, const uint8_t* bgr, int H, int W, uint8_t* gray, int bits);
// cvtColor(COLOR_RGB2GRAY) applied to BGR data, as computeNCC does (M.cpp:835,840): the R and B coefficients swap
int launch_rgb2gray(hipStream_t s, const uint8_t* bgr, int H, int W, uint8_t* gray, int bits);
int launch_cost_ad(hipStream_t s, const uint8_t* L, const uint8_t* R, int H, int W, int C, int disp_type, int minD,
                   int numD, int do_thresh /* 0: AD, 1: TAD mask, 2: SD */, int threshold, uint8_t* cost);
int launch_wta(hipStream_t s, const float* vol, int n, int H, int W, int minD, float* disp);

struct BilateralLaunch {
    const uint8_t* gL;
    const uint8_t* gR;
    int H, W, win, minD, nD;  // nD = number of candidates (numD + 1 for the reference's inclusive range)
    const int4* taps;         // {sample cell offset dys*LW+dxs, weight dx, weight dy, class*256}
    const float* lut;         // [ncls][256]
    int ntaps;
    int flip;    // 1: mirrored problem (DISPARITY_RIGHT), taps must come from the mirrored table
    float* vol;  // optional [nD][H][W]
    float* disp; // [H][W]
    double* partE;  // optional scratch [max_slices][H][W]: per-slice winners when the d range is split over grid.z
    float* partD;
    int max_slices;
    int c_begin = 0;  // > 0: only candidates [c_begin, nD)
    int resume = 0;   // 1: the running minimum starts from partE / partD ([H][W]) instead of DBL_MAX
    int out_slice = -1;  // >= 0: write this launch's winners to slice out_slice of partE / partD (merged later) instead of disp
};
// xq form of the classic kernel (k_bilateral_xq.hip): candidates [0, bilateral_xq_candidates()) of a DISPARITY_LEFT, win = 15
// problem with at least that many candidates; writes the running minimum to bestE / bestD for the tail launch, or -- when
// there is no tail (disp != nullptr) -- the disparity itself
int bilateral_xq_candidates(int nwave);  // nwave = 8 / 4 wavefronts per workgroup: 128 / 64 candidates
int launch_bilateral_xq(hipStream_t s, hipStream_t s_border, int nwave, const uint8_t* gL, const uint8_t* gR, int H, int W, int minD,
                        const int4* cells, const float* lut, float* vol, double* bestE, float* bestD, float* disp, bool right = false);
int launch_bilateral(hipStream_t s, const BilateralLaunch& a);
// winners of per-slice partial WTAs (candidate range split over grid.z for small frames) -> disparity, strict '<' in ascending d
int launch_merge_slices(hipStream_t s, const double* partE, const float* partD, int nz, size_t plane, float* disp);
int bilateral_lds_row_stride(int win);  // LW of the kernel's sample tile (taps[].x is expressed in it)

// ---- cost kernels (k_cost.hip) ----
int launch_scharr_x(hipStream_t s, const uint8_t* img, int H, int W, int pad, short* grad /* [H][W+pad][3] */);
// ord_scratch (similarity_parts_words() u32 words) / scales (optional): fused per-slice min/max -> normalize()
// parameters of every cost plane
size_t similarity_parts_words(int H, int W, int numD);
int launch_similarity(hipStream_t s, const uint8_t* L, const uint8_t* R, const short* gL, const short* gR, int H, int W,
                      int minD, int numD, double regularity, double thresC, double thresG, float* cost, uint32_t* ord_scratch,
                      float2* scales);
int launch_pad_reflect(hipStream_t s, const float* src, int n, int H, int W, int h, float* dst);
// per-slice normalize(NORM_MINMAX) parameters {scale, shift} of a dense f32 volume [n][plane]
int launch_slice_scales(hipStream_t s, const float* vol, int n, size_t plane, uint32_t* ord_scratch /* 2n */, float2* scales);
int launch_u8_scale(hipStream_t s, const uint8_t* img, size_t nbytes, uint32_t* ord_scratch /* 2 */, float2* scale1);
int launch_guide_scales_lr(hipStream_t s, const uint8_t* ref_img, const uint8_t* shifted_img, int H, int W, int minD, int numD,
                           int disp_type, uint32_t* ord_scratch, int* colmm_scratch /* 2W */, float2* scales /* numD */);

// ---- NCC cost (k_ncc.hip), computeNCC / getInputImgNCC, M.cpp:767-1013; launch_box_mean_u8 lives in k_guided.hip ----
struct NccLaunch {
    const uint8_t* gref;  // reference gray image [H][W]
    const float* mref;    // its box means
    const double* sref;   // its window sums of squares
    const uint8_t* goth;  // other gray image, REFLECT-padded to [H][Wp]
    const float* moth;
    const double* soth;
    int H, W, Wp, win, minD, numD, right, nwta;
    float* vol;   // optional [numD][H][W], un-normalised
    float* disp;  // optional [H][W], WTA over the first nwta candidates
};
int launch_pad_gray(hipStream_t s, const uint8_t* g, int H, int W, int padL, int padR, uint8_t* out);
int launch_box_mean_u8(hipStream_t s, const uint8_t* img, int H, int W, int win, float* mean);  // boxFilter(8U -> 32F)
int launch_ncc_selfsum(hipStream_t s, const uint8_t* g, const float* mean, int H, int W, int win, double* ss);
int launch_ncc(hipStream_t s, const NccLaunch& a);
int launch_apply_scales(hipStream_t s, float* vol, int n, size_t plane, const float2* scales);

// ---- driver-side pre/post-processing (k_prep.hip), SURVEY 8f row f3 ----
int launch_resize_linear(hipStream_t s, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw);
int launch_bgr2hsv(hipStream_t s, const uint8_t* bgr, size_t n, const int* sdiv, const int* hdiv, uint8_t* hsv);
int launch_boost_hsv2bgr(hipStream_t s, const uint8_t* hsv, int H, int W, const int* taps, int ntaps, const float* color_lut,
                         uint8_t* bgr);
int launch_disp_to_u8(hipStream_t s, const float* disp, size_t n, int normalize, uint8_t* out, int* mm_scratch);

// ---- box means / guided filter (k_guided.hip) ----
int launch_cost_sad(hipStream_t s, const uint8_t* gl, const uint8_t* gr, int H, int W, int disp_type, int win, int minD,
                    int numD, float* cost);
struct GuidedLaunch {
    const uint32_t* guideA; // BGRX plane holding guide channels 0..2
    const uint32_t* guideB; // BGRX plane holding guide channels 3..5 (C = 6), else null
    int shiftA, shiftB;     // plane A / B is read at reflect(x + shift*d): GuidedF LEFT = (0,-1), RIGHT = (+1,0); else 0
    int C;                  // 3 or 6
    int guide_per_slice;    // 1: guide (hence its statistics and scales) changes with the slice
    const float2* gscales;  // guide normalize() parameters (1 or n entries)
    const float* P;         // raw cost volume [n][H][W]
    const float2* pscales;  // per-slice normalize() parameters of P
    int H, W, n, r, minD;
    double eps;
    int nan_safe;           // 1: P may hold NaN (NCC costs of flat windows; caller-supplied P): window sums are rebuilt when a running sum is poisoned
    float* stats;           // scratch, guided_stats_floats(): {meanI_c, var_c+eps} interleaved per pixel and BGRX word
    int* rep_scratch;       // scratch, n ints (or null): scale-group representative of every slice (6-channel per-slice guides)
    float* ab;              // scratch, guided_ab_floats(): {a_c, b} per pixel (3-channel guide: strip-major float2 tiles, k_guided.hip ABTiles)
    float* q;               // out [n][H][W]
    const AswTuning* tune;  // the context's switches
};
size_t guided_stats_floats(int C, int nstat, int H, int W);
size_t guided_ab_floats(int C, int n, int H, int W, int r);
// true: launch_guided will run the fused a/b -> q walk for this problem (no a/b scratch is touched)
bool guided_uses_fused(const AswTuning& t, int C, int guide_per_slice, int shifted, int nan_safe, int H, int W, int n, int r);
int launch_pack_words(hipStream_t s, const uint8_t* img, int H, int W, int C, int w, uint32_t* out);
int launch_guided(hipStream_t s, const GuidedLaunch& a);

// ---- geodesic support weights (k_geodesic.hip) ----
int launch_pack_bgrx(hipStream_t s, const uint8_t* bgr, int H, int W, uint32_t* out);
int launch_geodesic_weights_u16(hipStream_t s, const uint32_t* img, int H, int W, int win, int iter, uint16_t* planes);
int launch_geodesic_weights_f32(hipStream_t s, const uint32_t* img, int H, int W, int win, int iter, float* planes);
int launch_planes_to_windows(hipStream_t s, const float* planes, int H, int W, int cells, float* out);
// c_begin / out_slice: only candidates [c_begin, nD), winners to slice out_slice of partE / partD (-1: to disp)
int launch_asw_geodesic(hipStream_t s, const uint32_t* imgL, const uint32_t* imgR, const uint16_t* wL, const uint16_t* wR,
                        int H, int W, int win, int minD, int nD, int flip, float* vol, float* disp, double* partE /* optional
                        scratch [8][H][W] */, float* partD, int c_begin = 0, int out_slice = -1);
// a few candidates [c_begin, nD) of a 15x15 problem (the remainder behind the xq passes), un-mirrored: imgF / wF = the fixed image
int launch_asw_geodesic_few(hipStream_t s, const uint32_t* imgF, const uint32_t* imgO, const uint16_t* wF, const uint16_t* wO, int H,
                            int W, int minD, int c_begin, int nD, bool right, float* vol, double* outE, float* outD);
// xq form (k_geodesic_xq.hip): one pass = candidates [cbase, cbase + 16 * nwave), nwave = 8 or 4, DISPARITY_LEFT, win = 15
int geodesic_xq_pass_candidates(int nwave);
int launch_geodesic_xq(hipStream_t s, hipStream_t s_border, int nwave, const uint32_t* imgL, const uint32_t* imgR, const uint16_t* wL,
                       const uint16_t* wR, int H, int W, int minD, int cbase, float* vol, double* outE, float* outD, bool right = false);

// ---- weighted median (k_wmedian.hip) ----
int launch_wm_weights(hipStream_t s, const uint8_t* img, int H, int W, int pad, int win, const float* lut2, const float* wd,
                      float* out);
int launch_wmedian(hipStream_t s, const float* cost, const float* wLd, const float* wRb, int H, int W, int win, int numD,
                   int max_off, float* out);
size_t wmedian_tile_list_slots(int H, int W, int d_count);
int launch_wmedian_tile(hipStream_t s, const float* cost, const float* wLd, const float* wRb, int H, int W, int numD, int max_off,
                        int d_begin, int d_count, uint32_t* listC, uint16_t* listP, float* out, int nsplit /* A/B: AswTuning */);
// windows 17x17 .. 37x37 (k_wmedian_tile_gen.hip)
bool wmedian_tile_gen_supported(int win);
int wmedian_tile_gen_slots(int win);
int launch_wmedian_tile_gen(hipStream_t s, const float* cost, const float* wLd, const float* wRb, int H, int W, int win, int numD,
                            int max_off, int d_begin, int d_count, uint32_t* listC, uint16_t* listP, float* out, int rows_per_part);

// ---- O(1)-bilateral ASW (BLO1), k_blo1.hip ----
int launch_blo1(hipStream_t s, const uint8_t* gl, const uint8_t* gr, const float* cost, int step, int H, int W, int disp_type,
                int win, int numD, float* vol, float* disp);

int launch_lr_check(hipStream_t s, const float* dl, const float* dr, int H, int W, float max_diff, float invalid, float* out,
                    unsigned* n_invalid);

// ---- bilateral-grid ASW, k_bilgrid.hip ----
// grid extents (last index per axis; both range axes share nz); ASW_ERR_BAD_ARGUMENT for rates <= 0 or a range axis too fine for LDS
int bilgrid_dims(int H, int W, double rate_s, double rate_r, int* nx, int* ny, int* nz);
int launch_bilgrid(hipStream_t s, const uint8_t* gl, const uint8_t* gr, int H, int W, double rate_s, double rate_r, int minD,
                   int numD, double* F, int* S, double* best, float* vol, float* disp);

// ---- hooks for the batch scheduler (batch.hip) ----
int asw_internal_stage_slot(asw_ctx* ctx, int slot, int rows, int cols, int channels, Frame** out);
int asw_internal_enqueue_match(asw_ctx* ctx, int slot, int disparity_type, int algorithm, int win_size, int min_disparity,
                               int num_disparity);  // enqueues on ctx->stream, does not wait
int asw_internal_check_pair(const asw_image* l, const asw_image* r, const asw_image* d);
