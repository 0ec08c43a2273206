// Host-side pieces shared by the three translation units behind the C-ABI:
//   asw_context.hip  context, frame slots, host <-> HBM plumbing, pre/post-processing entry points
//   asw_methods.hip  the method runners (tables, scratch, launch sequences) and the selector
//   asw_api.hip      the per-method / cost-builder / building-block entry points of include/asw_mi355x.h
#pragma once
#include "asw_internal.h"

#define ASW_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != ASW_OK) return _rc; \
    } while (0)

struct MatchParams {
    int disparity_type, win, minD, numD;
    double gamma_c = 30, gamma_g = 20;  // M.cpp:58
    double eps = 1e-6;                  // M.cpp:73,76
    double rate_s = 10, rate_r = 10;    // M.cpp:82
    double blo_rate_r = 0.015;          // M.cpp:70
    double grid_rate_s = 10, grid_rate_r = 10;  // M.cpp:67
};

int check_u8_image(const asw_image* im);
int check_pair(const asw_image* L, const asw_image* R);
int upload_image(asw_ctx* ctx, const asw_image* im, DevBuf& dst);
int check_disp_out(const asw_image* d, int rows, int cols);
Frame* frame_slot(asw_ctx* ctx, int slot, bool create);
int upload_pair_into(asw_ctx* ctx, Frame* f, const asw_image* left, const asw_image* right);
int download_disparity_from(asw_ctx* ctx, Frame* f, asw_image* disp);
int download_volume_from(asw_ctx* ctx, Frame* f, float* out, size_t n_floats);
int build_similarity_volume(asw_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int H, int W, int minD, int numD,
                                   double regularity, double thresC, double thresG, float* cost,
                                   uint32_t* ord_scratch = nullptr, float2* scales = nullptr);
int run_ncc_cost(asw_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int H, int W, int disparity_type, int win, int minD,
                        int numD, float* vol /* optional, un-normalised */, float* disp /* optional */, int nwta,
                        int channels = 3);
int run_method(asw_ctx* ctx, Frame* f, int algorithm, const MatchParams& mp, bool keep_volume, bool sync = true);
int match_host(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp, int algorithm,
                      const MatchParams& mp, float* cost_volume_out, size_t cost_volume_floats);
