// Per-method, cost-builder and building-block entry points of the C-ABI (include/asw_mi355x.h): argument checking, staging
// of caller-owned host buffers, dispatch to the method runners.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "asw_internal.h"
#include "asw_host.h"

// Host buffers in, host buffers out.  Works on the context's private frame: the caller's resident slots (asw_upload_pair /
// asw_match_resident) are never touched by a one-call entry point.
int match_host(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp, int algorithm,
                      const MatchParams& mp, float* cost_volume_out, size_t cost_volume_floats)
{
    if (!ctx) return ASW_ERR_BAD_ARGUMENT;
    ASW_TRY(check_pair(left, right));
    ASW_TRY(check_disp_out(disp, left->rows, left->cols));
    if (cost_volume_out) {  // the caller states what its buffer holds; a short one is refused before anything is written
        const int planes = asw_volume_planes(algorithm, mp.numD);
        if (planes > 0 && cost_volume_floats < (size_t)planes * left->rows * left->cols) return ASW_ERR_BAD_ARGUMENT;
    }
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    Frame* f = &ctx->host_frame;
    ASW_TRY(upload_pair_into(ctx, f, left, right));
    ASW_TRY(run_method(ctx, f, algorithm, mp, cost_volume_out != nullptr));
    ASW_TRY(download_disparity_from(ctx, f, disp));
    if (cost_volume_out) {
        if (f->vol_floats > cost_volume_floats) return ASW_ERR_BAD_ARGUMENT;
        ASW_TRY(download_volume_from(ctx, f, cost_volume_out, f->vol_floats));
    }
    return ASW_OK;
}

extern "C" int asw_volume_planes(int algorithm, int num_disparity)
{
    switch (algorithm) {
    case ASW_ALG_ADAPTIVE_WEIGHT:            // offset <= max_offset, M.cpp:1021,1074
    case ASW_ALG_ADAPTIVE_WEIGHT_8DIRECT:    // M.cpp:1171
    case ASW_ALG_ADAPTIVE_WEIGHT_GEODESIC:   // M.cpp:1447,1467
    case ASW_ALG_ADAPTIVE_WEIGHT_BILATERAL_GRID:  // M.cpp:2256,2280
        return num_disparity + 1;
    case ASW_ALG_ADAPTIVE_WEIGHT_BLO1:
    case ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER:
    case ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_2:
    case ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_3:
    case ASW_ALG_ADAPTIVE_WEIGHT_MEDIAN:
    case ASW_ALG_NCC:
        return num_disparity;
    default:
        return 0;
    }
}

extern "C" int asw_stereo_match(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                int disparity_type, int algorithm, int win_size, int min_disparity,
                                int num_disparity, float* cost_volume_out, size_t cost_volume_floats)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    return match_host(ctx, left, right, disp, algorithm, mp, cost_volume_out, cost_volume_floats);
}

extern "C" int asw_aggregate_bilateral(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                       double gamma_c, double gamma_g, int disparity_type, int win_size,
                                       int min_disparity, int num_disparity, float* cost_volume_out, size_t cost_volume_floats)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    mp.gamma_c = gamma_c; mp.gamma_g = gamma_g;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT, mp, cost_volume_out, cost_volume_floats);
}

extern "C" int asw_aggregate_direct8(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                     int disparity_type, int win_size, int min_disparity, int num_disparity,
                                     float* cost_volume_out, size_t cost_volume_floats)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_8DIRECT, mp, cost_volume_out, cost_volume_floats);
}

// ------------------------------------------------------------------------------------------
// cost builders and small building blocks
// ------------------------------------------------------------------------------------------
static int cost_ad_common(asw_ctx* ctx, const asw_image* left, const asw_image* right, uint8_t* cost, int disparity_type,
                          int do_thresh, int threshold, int minD, int numD)
{
    if (!ctx || !cost) return ASW_ERR_BAD_ARGUMENT;
    ASW_TRY(check_pair(left, right));
    if (numD <= 0 || minD < 0) return ASW_ERR_BAD_ARGUMENT;
    if (left->channels != 1 && left->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;  // no branch in M.cpp:227,264
    if (disparity_type != ASW_DISPARITY_LEFT && disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    const int H = left->rows, W = left->cols, C = left->channels;
    DevBuf& dl = ctx->buf("stageL");
    DevBuf& dr = ctx->buf("stageR");
    DevBuf& dc = ctx->buf("cost_u8");
    ASW_TRY(upload_image(ctx, left, dl));
    ASW_TRY(upload_image(ctx, right, dr));
    size_t bytes = (size_t)numD * H * W;
    ASW_TRY(dc.ensure(bytes));
    ASW_TRY(launch_cost_ad(ctx->stream, dl.as<uint8_t>(), dr.as<uint8_t>(), H, W, C, disparity_type, minD, numD, do_thresh,
                           threshold, dc.as<uint8_t>()));
    ASW_HIP_TRY(hipMemcpyAsync(cost, dc.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_cost_ad(asw_ctx* ctx, const asw_image* left, const asw_image* right, uint8_t* cost,
                           int disparity_type, int min_disparity, int num_disparity)
{
    return cost_ad_common(ctx, left, right, cost, disparity_type, 0, 0, min_disparity, num_disparity);
}

extern "C" int asw_cost_tad(asw_ctx* ctx, const asw_image* left, const asw_image* right, uint8_t* cost,
                            int disparity_type, int threshold_t, int min_disparity, int num_disparity)
{
    return cost_ad_common(ctx, left, right, cost, disparity_type, 1, threshold_t, min_disparity, num_disparity);
}

extern "C" int asw_cost_sd(asw_ctx* ctx, const asw_image* left, const asw_image* right, uint8_t* cost,
                           int disparity_type, int min_disparity, int num_disparity)
{
    return cost_ad_common(ctx, left, right, cost, disparity_type, 2, 0, min_disparity, num_disparity);
}

extern "C" int asw_bgr2gray(asw_ctx* ctx, const asw_image* bgr, uint8_t* gray)
{
    if (!ctx || !gray) return ASW_ERR_BAD_ARGUMENT;
    ASW_TRY(check_u8_image(bgr));
    if (bgr->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    DevBuf& d = ctx->buf("stageL");
    DevBuf& g = ctx->buf("grayL");
    ASW_TRY(upload_image(ctx, bgr, d));
    size_t n = (size_t)bgr->rows * bgr->cols;
    ASW_TRY(g.ensure(n));
    ASW_TRY(launch_bgr2gray(ctx->stream, d.as<uint8_t>(), bgr->rows, bgr->cols, g.as<uint8_t>(), ctx->gray_bits));
    ASW_HIP_TRY(hipMemcpyAsync(gray, g.p, n, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_wta(asw_ctx* ctx, const float* cost_volume, int n, int rows, int cols, int min_disparity, float* disp)
{
    if (!ctx || !cost_volume || !disp || n <= 0 || rows <= 0 || cols <= 0) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    DevBuf& v = ctx->buf("wta_vol");
    DevBuf& d = ctx->buf("wta_disp");
    size_t plane = (size_t)rows * cols;
    ASW_TRY(v.ensure(plane * n * 4));
    ASW_TRY(d.ensure(plane * 4));
    ASW_HIP_TRY(hipMemcpyAsync(v.p, cost_volume, plane * n * 4, hipMemcpyHostToDevice, ctx->stream));
    ASW_TRY(launch_wta(ctx->stream, v.as<float>(), n, rows, cols, min_disparity, d.as<float>()));
    ASW_HIP_TRY(hipMemcpyAsync(disp, d.p, plane * 4, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_lr_check(asw_ctx* ctx, const float* disp_left, const float* disp_right, int rows, int cols, float max_diff,
                            float invalid_value, float* out, int* n_invalid)
{
    if (!ctx || !disp_left || !disp_right || !out || rows <= 0 || cols <= 0 || !(max_diff >= 0)) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    const size_t plane = (size_t)rows * cols;
    DevBuf& a = ctx->buf("lr_left");
    DevBuf& b = ctx->buf("lr_right");
    DevBuf& o = ctx->buf("lr_out");
    DevBuf& c = ctx->buf("lr_count");
    ASW_TRY(a.ensure(plane * 4));
    ASW_TRY(b.ensure(plane * 4));
    ASW_TRY(o.ensure(plane * 4));
    ASW_TRY(c.ensure(sizeof(unsigned)));
    ASW_HIP_TRY(hipMemcpyAsync(a.p, disp_left, plane * 4, hipMemcpyHostToDevice, ctx->stream));
    ASW_HIP_TRY(hipMemcpyAsync(b.p, disp_right, plane * 4, hipMemcpyHostToDevice, ctx->stream));
    ASW_TRY(launch_lr_check(ctx->stream, a.as<float>(), b.as<float>(), rows, cols, max_diff, invalid_value, o.as<float>(), c.as<unsigned>()));
    unsigned bad = 0;
    ASW_HIP_TRY(hipMemcpyAsync(out, o.p, plane * 4, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipMemcpyAsync(&bad, c.p, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n_invalid) *n_invalid = (int)bad;
    return ASW_OK;
}

extern "C" int asw_aggregate_guided(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                    int disparity_type, double eps, int win_size, int min_disparity, int num_disparity,
                                    float* cost_volume_out, size_t cost_volume_floats)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity; mp.eps = eps;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER, mp, cost_volume_out, cost_volume_floats);
}

extern "C" int asw_aggregate_guided2(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                     int disparity_type, double eps, int win_size, int min_disparity, int num_disparity,
                                     float* cost_volume_out, size_t cost_volume_floats)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity; mp.eps = eps;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_2, mp, cost_volume_out, cost_volume_floats);
}

extern "C" int asw_cost_similarity(asw_ctx* ctx, const asw_image* left, const asw_image* right, float* cost,
                                   double regularity, double thres_c, double thres_g, int disparity_type, int win_size,
                                   int min_disparity, int num_disparity)
{
    if (!ctx || !cost) return ASW_ERR_BAD_ARGUMENT;
    if (win_size != 0 && win_size % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // M.cpp:654-657 (before anything else)
    ASW_TRY(check_pair(left, right));
    if (num_disparity <= 0 || min_disparity < 0 || win_size < 0) return ASW_ERR_BAD_ARGUMENT;
    // only DISPARITY_LEFT + 3 channels executes in the reference; the other branches throw (App. B-7)
    if (left->channels != 3 || disparity_type != ASW_DISPARITY_LEFT) return ASW_ERR_UNSUPPORTED_LAYOUT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    const int H = left->rows, W = left->cols, n = num_disparity, h = win_size / 2;
    DevBuf& dl = ctx->buf("stageL");
    DevBuf& dr = ctx->buf("stageR");
    DevBuf& raw = ctx->buf("g_raw");
    ASW_TRY(upload_image(ctx, left, dl));
    ASW_TRY(upload_image(ctx, right, dr));
    ASW_TRY(raw.ensure((size_t)n * H * W * 4));
    ASW_TRY(build_similarity_volume(ctx, dl.as<uint8_t>(), dr.as<uint8_t>(), H, W, min_disparity, n, regularity, thres_c, thres_g,
                                    raw.as<float>()));
    const float* src = raw.as<float>();
    size_t out_floats = (size_t)n * H * W;
    if (win_size > 0) {
        DevBuf& pad = ctx->buf("g_pad");
        out_floats = (size_t)n * (H + 2 * h) * (W + 2 * h);
        ASW_TRY(pad.ensure(out_floats * 4));
        ASW_TRY(launch_pad_reflect(ctx->stream, raw.as<float>(), n, H, W, h, pad.as<float>()));
        src = pad.as<float>();
    }
    ASW_HIP_TRY(hipMemcpyAsync(cost, src, out_floats * 4, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_cost_sad(asw_ctx* ctx, const asw_image* left, const asw_image* right, float* cost, int disparity_type,
                            int win_size, int min_disparity, int num_disparity)
{
    if (!ctx || !cost) return ASW_ERR_BAD_ARGUMENT;
    ASW_TRY(check_pair(left, right));
    if (win_size % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // M.cpp:2458-2462
    if (num_disparity <= 0 || min_disparity < 0 || win_size < 1 || win_size > 128) return ASW_ERR_BAD_ARGUMENT;
    if (left->channels != 3 && left->channels != 1) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (disparity_type != ASW_DISPARITY_LEFT && disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    const int H = left->rows, W = left->cols, n = num_disparity;
    DevBuf& dl = ctx->buf("stageL");
    DevBuf& dr = ctx->buf("stageR");
    DevBuf& gl = ctx->buf("grayL");
    DevBuf& gr = ctx->buf("grayR");
    DevBuf& raw = ctx->buf("g_raw");
    ASW_TRY(upload_image(ctx, left, dl));
    ASW_TRY(upload_image(ctx, right, dr));
    ASW_TRY(raw.ensure((size_t)n * H * W * 4));
    const uint8_t *pl = dl.as<uint8_t>(), *pr = dr.as<uint8_t>();
    if (left->channels == 3) {  // M.cpp:2446-2456
        ASW_TRY(gl.ensure((size_t)H * W));
        ASW_TRY(gr.ensure((size_t)H * W));
        ASW_TRY(launch_bgr2gray(ctx->stream, pl, H, W, gl.as<uint8_t>(), ctx->gray_bits));
        ASW_TRY(launch_bgr2gray(ctx->stream, pr, H, W, gr.as<uint8_t>(), ctx->gray_bits));
        pl = gl.as<uint8_t>(); pr = gr.as<uint8_t>();
    }
    ASW_TRY(launch_cost_sad(ctx->stream, pl, pr, H, W, disparity_type, win_size, min_disparity, n, raw.as<float>()));
    ASW_HIP_TRY(hipMemcpyAsync(cost, raw.p, (size_t)n * H * W * 4, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

// getCostSAD_d (M.cpp:2442-2503) as the reference declares it: one disparity, the other view pre-bordered by the caller
extern "C" int asw_cost_sad_d(asw_ctx* ctx, const asw_image* left, const asw_image* right, float* cost, int disparity,
                              int disparity_type, int win_size)
{
    if (!ctx || !cost) return ASW_ERR_BAD_ARGUMENT;
    ASW_TRY(check_u8_image(left));
    ASW_TRY(check_u8_image(right));
    if ((left->channels != 1 && left->channels != 3) || (right->channels != 1 && right->channels != 3)) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (win_size % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // M.cpp:2458-2462
    if (win_size < 1 || win_size > 128) return ASW_ERR_BAD_ARGUMENT;
    if (disparity_type != ASW_DISPARITY_LEFT && disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    const bool lref = disparity_type == ASW_DISPARITY_LEFT;
    const asw_image* ref = lref ? left : right;   // the view the cost plane belongs to
    const asw_image* bord = lref ? right : left;  // the bordered (wider) other view
    if (ref->rows != bord->rows) return ASW_ERR_BAD_ARGUMENT;  // absdiff of unequal sizes throws in the reference
    const int H = ref->rows, W = ref->cols, Wb = bord->cols;
    if (Wb <= W) return ASW_ERR_SIZE_MISMATCH;  // M.cpp:2473-2476 / 2488-2491: return Mat()
    const int x0 = lref ? Wb - W - disparity : disparity;  // M.cpp:2478 / 2493
    if (x0 < 0 || x0 + W > Wb) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    DevBuf& dref = ctx->buf("stageL");
    DevBuf& dbord = ctx->buf("stageR");
    DevBuf& gref = ctx->buf("grayL");
    DevBuf& gbord = ctx->buf("sadd_gray_wide");
    DevBuf& gcrop = ctx->buf("grayR");
    DevBuf& raw = ctx->buf("g_raw");
    ASW_TRY(upload_image(ctx, ref, dref));
    ASW_TRY(upload_image(ctx, bord, dbord));
    ASW_TRY(gref.ensure((size_t)H * W));
    ASW_TRY(gbord.ensure((size_t)H * Wb));
    ASW_TRY(gcrop.ensure((size_t)H * W));
    ASW_TRY(raw.ensure((size_t)H * W * 4));
    const uint8_t* pref = dref.as<uint8_t>();
    const uint8_t* pbord = dbord.as<uint8_t>();
    if (ref->channels == 3) {  // M.cpp:2446-2456
        ASW_TRY(launch_bgr2gray(ctx->stream, pref, H, W, gref.as<uint8_t>(), ctx->gray_bits));
        pref = gref.as<uint8_t>();
    }
    if (bord->channels == 3) {
        ASW_TRY(launch_bgr2gray(ctx->stream, pbord, H, Wb, gbord.as<uint8_t>(), ctx->gray_bits));
        pbord = gbord.as<uint8_t>();
    }
    // the ROI of the bordered view as a dense plane; then |ref - roi| -> f32 -> boxFilter mean is launch_cost_sad at offset 0
    ASW_HIP_TRY(hipMemcpy2DAsync(gcrop.p, (size_t)W, pbord + x0, (size_t)Wb, (size_t)W, H, hipMemcpyDeviceToDevice, ctx->stream));
    ASW_TRY(launch_cost_sad(ctx->stream, pref, gcrop.as<uint8_t>(), H, W, ASW_DISPARITY_LEFT, win_size, 0, 1, raw.as<float>()));
    ASW_HIP_TRY(hipMemcpyAsync(cost, raw.p, (size_t)H * W * 4, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_cost_ncc(asw_ctx* ctx, const asw_image* left, const asw_image* right, float* cost, int disparity_type,
                            int win_size, int min_disparity, int num_disparity, int normalized)
{
    if (!ctx || !cost) return ASW_ERR_BAD_ARGUMENT;
    ASW_TRY(check_pair(left, right));
    if (win_size % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // M.cpp:939-942
    if (num_disparity <= 0 || min_disparity < 0 || win_size < 1 || win_size > 63) return ASW_ERR_BAD_ARGUMENT;
    if (left->channels != 3 && left->channels != 1) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (disparity_type != ASW_DISPARITY_LEFT && disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    const int H = left->rows, W = left->cols, n = num_disparity;
    const size_t plane = (size_t)H * W;
    DevBuf& dl = ctx->buf("stageL");
    DevBuf& dr = ctx->buf("stageR");
    DevBuf& raw = ctx->buf("g_raw");
    ASW_TRY(upload_image(ctx, left, dl));
    ASW_TRY(upload_image(ctx, right, dr));
    ASW_TRY(raw.ensure(plane * n * 4));
    ASW_TRY(run_ncc_cost(ctx, dl.as<uint8_t>(), dr.as<uint8_t>(), H, W, disparity_type, win_size, min_disparity, n, raw.as<float>(),
                         nullptr, 0, left->channels));
    if (normalized) {  // normalize(curCost_, curCost_norm, 0, 1, NORM_MINMAX), M.cpp:981-983
        DevBuf& ord = ctx->buf("g_ord");
        DevBuf& psc = ctx->buf("g_pscales");
        ASW_TRY(ord.ensure((size_t)(2 * n + 2) * 4));
        ASW_TRY(psc.ensure((size_t)n * sizeof(float2)));
        ASW_TRY(launch_slice_scales(ctx->stream, raw.as<float>(), n, plane, ord.as<uint32_t>(), psc.as<float2>()));
        ASW_TRY(launch_apply_scales(ctx->stream, raw.as<float>(), n, plane, psc.as<float2>()));
    }
    ASW_HIP_TRY(hipMemcpyAsync(cost, raw.p, plane * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_ncc_disparity(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp, int disparity_type,
                                 int win_size, int min_disparity, int num_disparity)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    return match_host(ctx, left, right, disp, ASW_ALG_NCC, mp, nullptr, 0);
}

extern "C" int asw_aggregate_guided3(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                     int disparity_type, double eps, int win_size, int min_disparity, int num_disparity,
                                     float* cost_volume_out, size_t cost_volume_floats)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity; mp.eps = eps;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_3, mp, cost_volume_out, cost_volume_floats);
}

extern "C" int asw_guided_filter(asw_ctx* ctx, const asw_image* guide, const float* p, float* q, int r, double eps)
{
    if (!ctx || !p || !q) return ASW_ERR_BAD_ARGUMENT;
    ASW_TRY(check_u8_image(guide));
    if (guide->channels != 3 && guide->channels != 6) return ASW_ERR_UNSUPPORTED_LAYOUT;  // M.cpp:2732-2734
    if (r < 1 || r > 128) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    const int H = guide->rows, W = guide->cols, C = guide->channels;
    const size_t plane = (size_t)H * W;
    DevBuf& dg = ctx->buf("stageL");
    DevBuf& raw = ctx->buf("g_raw");
    DevBuf& ord = ctx->buf("g_ord");
    DevBuf& psc = ctx->buf("g_pscales");
    DevBuf& gsc = ctx->buf("g_gscales");
    DevBuf& stats = ctx->buf("g_stats");
    DevBuf& ab = ctx->buf("g_ab");
    DevBuf& qv = ctx->buf("g_q1");
    DevBuf& pxa = ctx->buf("bgrxL");
    DevBuf& pxb = ctx->buf("bgrxR");
    ASW_TRY(upload_image(ctx, guide, dg));
    ASW_TRY(raw.ensure(plane * 4));
    ASW_TRY(ord.ensure(4 * 4));
    ASW_TRY(psc.ensure(sizeof(float2)));
    ASW_TRY(gsc.ensure(sizeof(float2)));
    ASW_TRY(stats.ensure(guided_stats_floats(C, 1, H, W) * 4));
    ASW_TRY(ab.ensure(guided_ab_floats(C, 1, H, W, r) * 4));
    ASW_TRY(qv.ensure(plane * 4));
    ASW_TRY(pxa.ensure((plane + 4) * 4));  // + slack: the q pass reads the guide words of a lane's two columns as one pair, the last one may start at column W-1
    ASW_TRY(pxb.ensure((plane + 4) * 4));
    ASW_TRY(launch_pack_words(ctx->stream, dg.as<uint8_t>(), H, W, C, 0, pxa.as<uint32_t>()));
    if (C == 6) ASW_TRY(launch_pack_words(ctx->stream, dg.as<uint8_t>(), H, W, C, 1, pxb.as<uint32_t>()));
    ASW_HIP_TRY(hipMemcpyAsync(raw.p, p, plane * 4, hipMemcpyHostToDevice, ctx->stream));
    ASW_TRY(launch_u8_scale(ctx->stream, dg.as<uint8_t>(), plane * C, ord.as<uint32_t>() + 2, gsc.as<float2>()));  // M.cpp:2774
    ASW_TRY(launch_slice_scales(ctx->stream, raw.as<float>(), 1, plane, ord.as<uint32_t>(), psc.as<float2>()));     // M.cpp:2775
    GuidedLaunch a;
    a.shiftA = 0; a.shiftB = 0; a.C = C; a.guide_per_slice = 0;
    a.nan_safe = 1;  // p is the caller's: it may hold NaN
    a.guideA = pxa.as<uint32_t>(); a.guideB = C == 6 ? pxb.as<uint32_t>() : nullptr;
    a.gscales = gsc.as<float2>(); a.P = raw.as<float>(); a.pscales = psc.as<float2>();
    a.H = H; a.W = W; a.n = 1; a.r = r; a.minD = 0; a.eps = eps;
    a.stats = stats.as<float>(); a.rep_scratch = nullptr; a.ab = ab.as<float>(); a.q = qv.as<float>();
    a.tune = &ctx->tune;
    ASW_TRY(launch_guided(ctx->stream, a));
    ASW_HIP_TRY(hipMemcpyAsync(q, qv.p, plane * 4, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_aggregate_geodesic(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                      int disparity_type, int win_size, int min_disparity, int num_disparity,
                                      float* cost_volume_out, size_t cost_volume_floats)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_GEODESIC, mp, cost_volume_out, cost_volume_floats);
}

extern "C" int asw_geodesic_dist(asw_ctx* ctx, const asw_image* img, float* out, int win_size, int iter_time)
{
    if (!ctx || !out) return ASW_ERR_BAD_ARGUMENT;
    ASW_TRY(check_u8_image(img));
    if (win_size % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // M.cpp:1394-1397
    if (img->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (win_size < 1 || win_size > 35 || iter_time < 0) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    const int H = img->rows, W = img->cols, cells = win_size * win_size;
    const size_t plane = (size_t)H * W;
    DevBuf& di = ctx->buf("stageL");
    DevBuf& px = ctx->buf("bgrxL");
    DevBuf& pf = ctx->buf("geoPlanesF");
    DevBuf& wo = ctx->buf("geoWindows");
    ASW_TRY(upload_image(ctx, img, di));
    ASW_TRY(px.ensure(plane * 4));
    ASW_TRY(pf.ensure(plane * cells * 4));
    ASW_TRY(wo.ensure(plane * cells * 4));
    ASW_TRY(launch_pack_bgrx(ctx->stream, di.as<uint8_t>(), H, W, px.as<uint32_t>()));
    ASW_TRY(launch_geodesic_weights_f32(ctx->stream, px.as<uint32_t>(), H, W, win_size, iter_time, pf.as<float>()));
    ASW_TRY(launch_planes_to_windows(ctx->stream, pf.as<float>(), H, W, cells, wo.as<float>()));
    ASW_HIP_TRY(hipMemcpyAsync(out, wo.p, plane * cells * 4, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_aggregate_blo1(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                  int disparity_type, double sample_rate_r, int win_size, int min_disparity,
                                  int num_disparity, float* cost_volume_out, size_t cost_volume_floats)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    mp.blo_rate_r = sample_rate_r;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_BLO1, mp, cost_volume_out, cost_volume_floats);
}

extern "C" int asw_aggregate_bilgrid(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                     int disparity_type, double sample_rate_s, double sample_rate_r, int min_disparity,
                                     int num_disparity, float* cost_volume_out, size_t cost_volume_floats)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = 1; mp.minD = min_disparity; mp.numD = num_disparity;
    mp.grid_rate_s = sample_rate_s; mp.grid_rate_r = sample_rate_r;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_BILATERAL_GRID, mp, cost_volume_out, cost_volume_floats);
}

extern "C" int asw_aggregate_wmedian(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                     int disparity_type, int win_size, double rate_s, double rate_r, int min_disparity,
                                     int num_disparity, float* cost_volume_out, size_t cost_volume_floats)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    mp.rate_s = rate_s; mp.rate_r = rate_r;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_MEDIAN, mp, cost_volume_out, cost_volume_floats);
}
