// C-ABI entry points (include/asw_mi355x.h): argument checking, host<->HBM staging, method dispatch.
// Host code follows the reference's selector (M.cpp:46-88) and its error behaviour.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "asw_internal.h"

// ------------------------------------------------------------------------------------------
// small runtime pieces
// ------------------------------------------------------------------------------------------
void asw_note_hip_error(hipError_t e, const char* what, const char* file, int line)
{
    if (getenv("ASW_QUIET") == nullptr)
        fprintf(stderr, "[asw_mi355x] HIP error %d (%s) at %s:%d in %s\n", (int)e, hipGetErrorString(e), file, line, what);
}

int DevBuf::ensure(size_t bytes)
{
    if (bytes <= cap && p) return ASW_OK;
    if (p) {
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    if (hipMalloc(&p, want) != hipSuccess) {
        p = nullptr;
        return ASW_ERR_ALLOC;
    }
    cap = want;
    return ASW_OK;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

#define ASW_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != ASW_OK) return _rc; \
    } while (0)

extern "C" const char* asw_status_string(int status)
{
    switch (status) {
    case ASW_OK: return "ok";
    case ASW_ERR_SIZE_MISMATCH: return "left/right size mismatch";
    case ASW_ERR_EVEN_WINDOW: return "window size must be odd";
    case ASW_ERR_UNSUPPORTED_METHOD: return "algorithm not on the accelerated path";
    case ASW_ERR_UNSUPPORTED_LAYOUT: return "unsupported channel layout / disparity type for this method";
    case ASW_ERR_HIP: return "HIP runtime error";
    case ASW_ERR_ALLOC: return "device allocation failed";
    case ASW_ERR_BAD_ARGUMENT: return "bad argument";
    case ASW_ERR_NO_FRAME: return "no resident frame in this slot";
    default: return "unknown status";
    }
}

extern "C" int asw_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int asw_create(int device_id, asw_ctx** out)
{
    if (!out) return ASW_ERR_BAD_ARGUMENT;
    *out = nullptr;
    int n = 0;
    ASW_HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(device_id));
    asw_ctx* c = new asw_ctx();
    c->device = device_id;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return ASW_ERR_HIP;
    }
    for (int i = 0; i < 4; i++)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) {
            delete c;
            return ASW_ERR_HIP;
        }
    *out = c;
    return ASW_OK;
}

extern "C" void asw_destroy(asw_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& f : ctx->frames) {
        f.L.release(); f.R.release(); f.disp.release(); f.vol.release();
    }
    for (auto& kv : ctx->scratch) kv.second.release();
    ctx->bil.taps.release();
    ctx->bil.lut.release();
    ctx->wm_lut2.release();
    ctx->wm_wd.release();
    for (int i = 0; i < 4; i++)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int asw_synchronize(asw_ctx* ctx)
{
    if (!ctx) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_get_timing(asw_ctx* ctx, asw_timing* out)
{
    if (!ctx || !out) return ASW_ERR_BAD_ARGUMENT;
    *out = ctx->timing;
    return ASW_OK;
}

// ------------------------------------------------------------------------------------------
// host <-> device image plumbing
// ------------------------------------------------------------------------------------------
static int check_u8_image(const asw_image* im)
{
    if (!im || !im->data || im->rows <= 0 || im->cols <= 0) return ASW_ERR_BAD_ARGUMENT;
    if (im->depth != ASW_8U) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (im->channels != 1 && im->channels != 3 && im->channels != 6) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (im->step < (size_t)im->cols * im->channels) return ASW_ERR_BAD_ARGUMENT;
    return ASW_OK;
}

static int check_pair(const asw_image* L, const asw_image* R)
{
    ASW_TRY(check_u8_image(L));
    ASW_TRY(check_u8_image(R));
    // leftImg.size != rightImg.size -> silent return in the reference (M.cpp:217-220)
    if (L->rows != R->rows || L->cols != R->cols) return ASW_ERR_SIZE_MISMATCH;
    if (L->channels != R->channels) return ASW_ERR_UNSUPPORTED_LAYOUT;
    return ASW_OK;
}

static int upload_image(asw_ctx* ctx, const asw_image* im, DevBuf& dst)
{
    size_t rowbytes = (size_t)im->cols * im->channels;
    ASW_TRY(dst.ensure(rowbytes * im->rows));
    ASW_HIP_TRY(hipMemcpy2DAsync(dst.p, rowbytes, im->data, im->step, rowbytes, im->rows, hipMemcpyHostToDevice, ctx->stream));
    return ASW_OK;
}

static int check_disp_out(const asw_image* d, int rows, int cols)
{
    if (!d || !d->data) return ASW_ERR_BAD_ARGUMENT;
    if (d->depth != ASW_32F || d->channels != 1) return ASW_ERR_BAD_ARGUMENT;
    if (d->rows != rows || d->cols != cols || d->step < (size_t)cols * 4) return ASW_ERR_BAD_ARGUMENT;
    return ASW_OK;
}

static Frame* frame_slot(asw_ctx* ctx, int slot, bool create)
{
    if (slot < 0 || slot >= 4096) return nullptr;
    if ((size_t)slot >= ctx->frames.size()) {
        if (!create) return nullptr;
        ctx->frames.resize(slot + 1);
    }
    return &ctx->frames[slot];
}

extern "C" int asw_upload_pair(asw_ctx* ctx, int slot, const asw_image* left, const asw_image* right)
{
    if (!ctx) return ASW_ERR_BAD_ARGUMENT;
    ASW_TRY(check_pair(left, right));
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    Frame* f = frame_slot(ctx, slot, true);
    if (!f) return ASW_ERR_BAD_ARGUMENT;
    f->valid = false;
    ASW_TRY(upload_image(ctx, left, f->L));
    ASW_TRY(upload_image(ctx, right, f->R));
    f->rows = left->rows; f->cols = left->cols; f->channels = left->channels;
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));  // host buffers are caller-owned: done with them on return
    f->valid = true;
    return ASW_OK;
}

// ------------------------------------------------------------------------------------------
// driver-side pre/post-processing on the device (aswStereoMatch.cpp:30-31, 67-89, 97-98)
// ------------------------------------------------------------------------------------------
static int ensure_prep_tables(asw_ctx* ctx)
{
    DevBuf& t = ctx->buf("prep_tables");
    if (t.p) return ASW_OK;
    // [0..255] sdiv, [256..511] hdiv (RGB2HSV_b, hsv_shift = 12), then the bilateralFilter(d=7, sigmaColor=10,
    // sigmaSpace=3) tables of main.cpp:76: 256 colour weights (f32 bits), tap count, taps {dy, dx, weight bits}
    std::vector<int> h(512 + 256 + 1 + 3 * 64, 0);
    for (int i = 1; i < 256; i++) {
        h[i] = (int)lrint((255 << 12) / (1. * i));
        h[256 + i] = (int)lrint((180 << 12) / (6. * i));
    }
    const double sigma_color = 10, sigma_space = 3;
    const int radius = 7 / 2;
    const double gc = -0.5 / (sigma_color * sigma_color), gs = -0.5 / (sigma_space * sigma_space);
    for (int i = 0; i < 256; i++) {
        float w = (float)exp((double)i * i * gc);
        memcpy(&h[512 + i], &w, 4);
    }
    int n = 0;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            float w = (float)exp(r * r * gs);
            h[769 + 3 * n] = i; h[769 + 3 * n + 1] = j;
            memcpy(&h[769 + 3 * n + 2], &w, 4);
            n++;
        }
    h[768] = n;
    ctx->prep_ntaps = n;
    ASW_TRY(t.ensure(h.size() * sizeof(int)));
    ASW_HIP_TRY(hipMemcpyAsync(t.p, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

static int prep_one(asw_ctx* ctx, const asw_image* full, int out_w, int out_h, int boost, DevBuf& dst)
{
    DevBuf& stage = ctx->buf("prep_full");
    DevBuf& small = ctx->buf("prep_small");
    DevBuf& hsv = ctx->buf("prep_hsv");
    const size_t n = (size_t)out_w * out_h;
    ASW_TRY(upload_image(ctx, full, stage));
    ASW_TRY(dst.ensure(n * 3));
    if (boost) ASW_TRY(small.ensure(n * 3));
    uint8_t* resized = boost ? small.as<uint8_t>() : dst.as<uint8_t>();
    ASW_TRY(launch_resize_linear(ctx->stream, stage.as<uint8_t>(), full->rows, full->cols, resized, out_h, out_w));
    if (boost) {
        const int* tab = ctx->buf("prep_tables").as<int>();
        ASW_TRY(hsv.ensure(n * 3));
        ASW_TRY(launch_bgr2hsv(ctx->stream, resized, n, tab, tab + 256, hsv.as<uint8_t>()));
        ASW_TRY(launch_boost_hsv2bgr(ctx->stream, hsv.as<uint8_t>(), out_h, out_w, tab + 769, ctx->prep_ntaps,
                                     reinterpret_cast<const float*>(tab + 512), dst.as<uint8_t>()));
    }
    return ASW_OK;
}

extern "C" int asw_preprocess_pair(asw_ctx* ctx, int slot, const asw_image* left_full, const asw_image* right_full, int out_width,
                                   int out_height, int detail_boost)
{
    if (!ctx || out_width < 1 || out_height < 1) return ASW_ERR_BAD_ARGUMENT;
    ASW_TRY(check_pair(left_full, right_full));
    if (left_full->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;  // COLOR_BGR2HSV asserts 3 or 4 channels
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    ASW_TRY(ensure_prep_tables(ctx));
    Frame* f = frame_slot(ctx, slot, true);
    if (!f) return ASW_ERR_BAD_ARGUMENT;
    f->valid = false;
    ASW_TRY(prep_one(ctx, left_full, out_width, out_height, detail_boost, f->L));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));  // the staging buffer is reused for the right image
    ASW_TRY(prep_one(ctx, right_full, out_width, out_height, detail_boost, f->R));
    f->rows = out_height; f->cols = out_width; f->channels = 3;
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    f->valid = true;
    return ASW_OK;
}

extern "C" int asw_download_pair(asw_ctx* ctx, int slot, asw_image* left, asw_image* right)
{
    if (!ctx || !left || !right || !left->data || !right->data) return ASW_ERR_BAD_ARGUMENT;
    Frame* f = frame_slot(ctx, slot, false);
    if (!f || !f->valid) return ASW_ERR_NO_FRAME;
    const size_t rowbytes = (size_t)f->cols * f->channels;
    for (asw_image* im : {left, right})
        if (im->depth != ASW_8U || im->rows != f->rows || im->cols != f->cols || im->channels != f->channels || im->step < rowbytes)
            return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    ASW_HIP_TRY(hipMemcpy2DAsync(left->data, left->step, f->L.p, rowbytes, rowbytes, f->rows, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipMemcpy2DAsync(right->data, right->step, f->R.p, rowbytes, rowbytes, f->rows, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_download_disparity_u8(asw_ctx* ctx, int slot, asw_image* disp_u8, int normalize)
{
    if (!ctx || !disp_u8 || !disp_u8->data) return ASW_ERR_BAD_ARGUMENT;
    Frame* f = frame_slot(ctx, slot, false);
    if (!f || !f->valid || !f->disp.p) return ASW_ERR_NO_FRAME;
    if (disp_u8->depth != ASW_8U || disp_u8->channels != 1 || disp_u8->rows != f->rows || disp_u8->cols != f->cols ||
        disp_u8->step < (size_t)f->cols)
        return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)f->rows * f->cols;
    DevBuf& u8 = ctx->buf("prep_disp_u8");
    DevBuf& mm = ctx->buf("prep_mm");
    ASW_TRY(u8.ensure(n));
    ASW_TRY(mm.ensure(2 * sizeof(int)));
    ASW_TRY(launch_disp_to_u8(ctx->stream, f->disp.as<float>(), n, normalize, u8.as<uint8_t>(), mm.as<int>()));
    ASW_HIP_TRY(hipMemcpy2DAsync(disp_u8->data, disp_u8->step, u8.p, (size_t)f->cols, (size_t)f->cols, f->rows, hipMemcpyDeviceToHost,
                                 ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_download_disparity(asw_ctx* ctx, int slot, asw_image* disp)
{
    if (!ctx) return ASW_ERR_BAD_ARGUMENT;
    Frame* f = frame_slot(ctx, slot, false);
    if (!f || !f->valid || !f->disp.p) return ASW_ERR_NO_FRAME;
    ASW_TRY(check_disp_out(disp, f->rows, f->cols));
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    ASW_HIP_TRY(hipMemcpy2DAsync(disp->data, disp->step, f->disp.p, (size_t)f->cols * 4, (size_t)f->cols * 4, f->rows,
                                 hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_download_volume(asw_ctx* ctx, int slot, float* out, size_t n_floats)
{
    if (!ctx || !out) return ASW_ERR_BAD_ARGUMENT;
    Frame* f = frame_slot(ctx, slot, false);
    if (!f || !f->valid || !f->vol.p || f->vol_floats == 0) return ASW_ERR_NO_FRAME;
    if (n_floats != f->vol_floats) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    ASW_HIP_TRY(hipMemcpyAsync(out, f->vol.p, n_floats * 4, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

// ------------------------------------------------------------------------------------------
// classic bilateral ASW: host-side tables (tap list with the reference's two index conventions,
// weight LUT with the reference's expression) -- M.cpp:1044-1066, 1088-1102, SURVEY App. B-2
// ------------------------------------------------------------------------------------------
// mirror = 1: table for the x-mirrored problem (DISPARITY_RIGHT runs as DISPARITY_LEFT on mirrored, swapped images:
// M.cpp:1134-1138 is M.cpp:1104-1108 under x -> W-1-x), i.e. every x direction negated, tap ORDER unchanged.
// Tap table + weight LUT of the bilateral kernel.  kind 0: computeAdaptiveWeight (M.cpp:1041-1102);
// kind 1: computeAdaptiveWeight_direct8 (M.cpp:1195-1221, 1238-1259).
static int ensure_bilateral_tables(asw_ctx* ctx, int kind, int win, double gamma_c, double gamma_g, int mirror)
{
    BilateralTables& t = ctx->bil;
    if (t.kind == kind && t.win == win && t.gamma_c == gamma_c && t.gamma_g == gamma_g && t.mirror == mirror && t.taps.p)
        return ASW_OK;
    const int ks = win, h = ks / 2;
    std::vector<int> dxw, dyw, dxs, dys;  // weight direction (build order) / sample offset (consume order)
    if (kind == 0) {
        const int nt = ks * ks - 1;
        dxs.resize(nt); dys.resize(nt);
        for (int j = -h; j < h + 1; j++)          // build order of the weight maps, M.cpp:1044-1053
            for (int i = -h; i < h + 1; i++) {
                if (i == 0 && j == 0) continue;
                dxw.push_back(i); dyw.push_back(j);
            }
        for (int i = 0; i < nt; i++) {            // consume order of the samples, M.cpp:1088-1102
            int kx, ky;
            if (i > ks * ks / 2) { kx = (i + 1) / ks; ky = (i + 1) % ks; }
            else { kx = i / ks; ky = i % ks; }
            dxs[i] = -h + kx; dys[i] = -h + ky;
        }
    } else {
        for (int j = -h; j < h + 1; j++)          // M.cpp:1195-1201 == 1238-1245: same order, same test
            for (int i = -h; i < h + 1; i++) {
                if (i == 0 && j == 0) continue;
                if (i == j || i == 0 || j == 0 || (i + j) == ks - 1) { dxw.push_back(i); dyw.push_back(j); }
            }
        dxs = dxw; dys = dyw;                     // the sample is the neighbour the weight was built for
    }
    const int nt = (int)dxw.size();
    // distance classes: distinct values of i*i + j*j
    std::vector<int> cls_of_r2(2 * h * h + 1, -1);
    std::vector<int> r2s;
    for (int i = 0; i < nt; i++) {
        int r2 = dxw[i] * dxw[i] + dyw[i] * dyw[i];
        if (cls_of_r2[r2] < 0) { cls_of_r2[r2] = (int)r2s.size(); r2s.push_back(r2); }
    }
    // The kernel consumes taps in groups of 4: pad with taps of an all-zero weight class (0*w*c adds +0.0 to both sums).
    const int nt_pad = (nt + 3) / 4 * 4, zero_cls = (int)r2s.size();
    std::vector<float> lut((r2s.size() + 1) * 256, 0.0f);
    const double k = 3;  // M.cpp:1024, 1175
    for (size_t c = 0; c < r2s.size(); c++) {
        double delta_g = sqrt((double)r2s[c]);  // M.cpp:1054, 1205
        for (int dc = 0; dc < 256; dc++) {
            double delta_c = (double)dc;
            lut[c * 256 + dc] = (float)(k * exp(-(delta_c / gamma_c + delta_g / gamma_g)));  // M.cpp:1065, 1214
        }
    }
    std::vector<int4> taps(nt_pad);
    const int LW = bilateral_lds_row_stride(win);  // row stride of the kernel's LDS sample tile
    for (int i = 0; i < nt; i++) {
        const int sx = mirror ? -1 : 1;
        taps[i].x = dys[i] * LW + sx * dxs[i];  // sample cell, consume order (classic: transposed, App. B-2)
        taps[i].y = sx * dxw[i];                // weight direction, build order
        taps[i].z = dyw[i];
        taps[i].w = cls_of_r2[dxw[i] * dxw[i] + dyw[i] * dyw[i]] * 256;
    }
    for (int i = nt; i < nt_pad; i++) taps[i] = make_int4(0, 0, 0, zero_cls * 256);
    ASW_TRY(t.taps.ensure(taps.size() * sizeof(int4)));
    ASW_TRY(t.lut.ensure(lut.size() * sizeof(float)));
    if (!taps.empty())  // win = 1 has no taps at all (every E is 0/0)
        ASW_HIP_TRY(hipMemcpyAsync(t.taps.p, taps.data(), taps.size() * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
    ASW_HIP_TRY(hipMemcpyAsync(t.lut.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));  // host vectors die at return
    t.kind = kind; t.win = win; t.gamma_c = gamma_c; t.gamma_g = gamma_g; t.mirror = mirror; t.ntaps = nt_pad;
    t.ncls = (int)r2s.size() + 1;
    return ASW_OK;
}

struct MatchParams {
    int disparity_type, win, minD, numD;
    double gamma_c = 30, gamma_g = 20;  // M.cpp:58
    double eps = 1e-6;                  // M.cpp:73,76
    double rate_s = 10, rate_r = 10;    // M.cpp:82
    double blo_rate_r = 0.015;          // M.cpp:70
};

// computeAdaptiveWeight (direct8 = false) and computeAdaptiveWeight_direct8 (direct8 = true: sparse support, its own
// gamma_g, DISPARITY_LEFT only -- the RIGHT branch of the reference indexes its weight vectors with a negative tap
// coordinate, M.cpp:1291-1295)
static int run_bilateral(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume, bool direct8 = false)
{
    if (mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // build decision: the reference has no guard
    if (mp.win < 1) return ASW_ERR_BAD_ARGUMENT;
    if (f->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;  // cvtColor(BGR2GRAY) asserts scn==3/4
    if (mp.disparity_type != ASW_DISPARITY_LEFT && mp.disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    if (direct8 && mp.disparity_type != ASW_DISPARITY_LEFT) return ASW_ERR_UNSUPPORTED_LAYOUT;
    const int flip = mp.disparity_type == ASW_DISPARITY_RIGHT ? 1 : 0;
    if (mp.win > 127) return ASW_ERR_BAD_ARGUMENT;
    const int H = f->rows, W = f->cols, nD = mp.numD + 1;  // inclusive range, M.cpp:1021,1074
    if (direct8)
        ASW_TRY(ensure_bilateral_tables(ctx, 1, mp.win, 30.0, (double)(mp.win * 2 / 3), 0));  // M.cpp:1175: integer division
    else
        ASW_TRY(ensure_bilateral_tables(ctx, 0, mp.win, mp.gamma_c, mp.gamma_g, flip));
    DevBuf& gl = ctx->buf("grayL");
    DevBuf& gr = ctx->buf("grayR");
    ASW_TRY(gl.ensure((size_t)H * W));
    ASW_TRY(gr.ensure((size_t)H * W));
    ASW_TRY(f->disp.ensure((size_t)H * W * 4));
    f->vol_floats = 0;
    if (keep_volume) {
        ASW_TRY(f->vol.ensure((size_t)nD * H * W * 4));
        f->vol_floats = (size_t)nD * H * W;
    }
    ASW_TRY(launch_bgr2gray(ctx->stream, f->L.as<uint8_t>(), H, W, gl.as<uint8_t>()));
    ASW_TRY(launch_bgr2gray(ctx->stream, f->R.as<uint8_t>(), H, W, gr.as<uint8_t>()));
    BilateralLaunch a;
    a.gL = flip ? gr.as<uint8_t>() : gl.as<uint8_t>();  // RIGHT: reference image = right, read mirrored in the kernel
    a.gR = flip ? gl.as<uint8_t>() : gr.as<uint8_t>();
    a.flip = flip;
    a.H = H; a.W = W; a.win = mp.win; a.minD = mp.minD; a.nD = nD;
    a.taps = ctx->bil.taps.as<int4>(); a.lut = ctx->bil.lut.as<float>(); a.ntaps = ctx->bil.ntaps;
    a.vol = keep_volume ? f->vol.as<float>() : nullptr;
    a.disp = f->disp.as<float>();
    a.partE = nullptr; a.partD = nullptr; a.max_slices = 0;
    if ((size_t)H * W <= (size_t)1 << 20) {  // small frames only: scratch for the grid.z split of the disparity range
        const int max_slices = 8;
        DevBuf& pe = ctx->buf("bil_partE");
        DevBuf& pd = ctx->buf("bil_partD");
        ASW_TRY(pe.ensure((size_t)max_slices * H * W * sizeof(double)));
        ASW_TRY(pd.ensure((size_t)max_slices * H * W * sizeof(float)));
        a.partE = pe.as<double>(); a.partD = pd.as<float>(); a.max_slices = max_slices;
    }
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    ASW_TRY(launch_bilateral(ctx->stream, a));
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = 1;
    return ASW_OK;
}


// ------------------------------------------------------------------------------------------
// guided-filter ASW: computeAdaptiveWeight_GuidedF_2 (M.cpp:2976-3050, TAD C+G cost, guide = left image)
// and computeAdaptiveWeight_GuidedF (M.cpp:2867-2963, SAD cost, 6-channel guide [L, R shifted by d])
// ------------------------------------------------------------------------------------------
static int build_similarity_volume(asw_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int H, int W, int minD, int numD,
                                   double regularity, double thresC, double thresG, float* cost,
                                   uint32_t* ord_scratch = nullptr, float2* scales = nullptr)
{
    const int max_off = minD + numD - 1;
    DevBuf& gl = ctx->buf("scharrL");
    DevBuf& gr = ctx->buf("scharrR");
    ASW_TRY(gl.ensure((size_t)H * W * 3 * sizeof(short)));
    ASW_TRY(gr.ensure((size_t)H * (W + max_off) * 3 * sizeof(short)));
    ASW_TRY(launch_scharr_x(ctx->stream, dL, H, W, 0, gl.as<short>()));
    ASW_TRY(launch_scharr_x(ctx->stream, dR, H, W, max_off, gr.as<short>()));  // gradient of the PADDED right image
    return launch_similarity(ctx->stream, dL, dR, gl.as<short>(), gr.as<short>(), H, W, minD, numD, regularity, thresC, thresG,
                             cost, ord_scratch, scales);
}

// ------------------------------------------------------------------------------------------
// NCC cost (computeNCC / getInputImgNCC, M.cpp:767-1013): gray images, box means, window sums of squares, then k_ncc
// ------------------------------------------------------------------------------------------
static int run_ncc_cost(asw_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int H, int W, int disparity_type, int win, int minD,
                        int numD, float* vol /* optional, un-normalised */, float* disp /* optional */, int nwta,
                        int channels = 3)
{
    if (win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // M.cpp:828-831, 939-942
    if (win < 1 || win > 63) return ASW_ERR_BAD_ARGUMENT;
    const bool right = disparity_type == ASW_DISPARITY_RIGHT;
    const int max_off = minD + numD - 1, Wp = W + max_off;
    const size_t plane = (size_t)H * W, pplane = (size_t)H * Wp;
    DevBuf& g0 = ctx->buf("ncc_gray_ref");
    DevBuf& g1 = ctx->buf("ncc_gray_oth");
    DevBuf& gp = ctx->buf("ncc_gray_pad");
    DevBuf& m0 = ctx->buf("ncc_mean_ref");
    DevBuf& m1 = ctx->buf("ncc_mean_oth");
    DevBuf& s0 = ctx->buf("ncc_ss_ref");
    DevBuf& s1 = ctx->buf("ncc_ss_oth");
    ASW_TRY(g0.ensure(plane)); ASW_TRY(g1.ensure(plane)); ASW_TRY(gp.ensure(pplane));
    ASW_TRY(m0.ensure(plane * 4)); ASW_TRY(m1.ensure(pplane * 4));
    ASW_TRY(s0.ensure(plane * 8)); ASW_TRY(s1.ensure(pplane * 8));
    // COLOR_RGB2GRAY on BGR data (M.cpp:835,840); reference image = left (LEFT) or right (RIGHT)
    if (channels == 3) {
        ASW_TRY(launch_rgb2gray(ctx->stream, right ? dR : dL, H, W, g0.as<uint8_t>()));
        ASW_TRY(launch_rgb2gray(ctx->stream, right ? dL : dR, H, W, g1.as<uint8_t>()));
    } else {  // single-channel input is used as it is (M.cpp:833-841: cvtColor only for 3 channels)
        ASW_HIP_TRY(hipMemcpyAsync(g0.p, right ? dR : dL, plane, hipMemcpyDeviceToDevice, ctx->stream));
        ASW_HIP_TRY(hipMemcpyAsync(g1.p, right ? dL : dR, plane, hipMemcpyDeviceToDevice, ctx->stream));
    }
    // the other image is padded by max_offset REFLECT columns: on the left (LEFT, M.cpp:852) / on the right (RIGHT, M.cpp:882)
    ASW_TRY(launch_pad_gray(ctx->stream, g1.as<uint8_t>(), H, W, right ? 0 : max_off, right ? max_off : 0, gp.as<uint8_t>()));
    ASW_TRY(launch_box_mean_u8(ctx->stream, g0.as<uint8_t>(), H, W, win, m0.as<float>()));   // M.cpp:785-786
    ASW_TRY(launch_box_mean_u8(ctx->stream, gp.as<uint8_t>(), H, Wp, win, m1.as<float>()));
    ASW_TRY(launch_ncc_selfsum(ctx->stream, g0.as<uint8_t>(), m0.as<float>(), H, W, win, s0.as<double>()));
    ASW_TRY(launch_ncc_selfsum(ctx->stream, gp.as<uint8_t>(), m1.as<float>(), H, Wp, win, s1.as<double>()));
    NccLaunch a;
    a.gref = g0.as<uint8_t>(); a.mref = m0.as<float>(); a.sref = s0.as<double>();
    a.goth = gp.as<uint8_t>(); a.moth = m1.as<float>(); a.soth = s1.as<double>();
    a.H = H; a.W = W; a.Wp = Wp; a.win = win; a.minD = minD; a.numD = numD; a.right = right ? 1 : 0; a.nwta = nwta;
    a.vol = vol; a.disp = disp;
    return launch_ncc(ctx->stream, a);
}

// computeNCC -> disparity (M.cpp:812-913): candidates minD .. max_offset-1 only, the SMALLEST cost wins (LEFT);
// DISPARITY_RIGHT compares `cost > DBL_MAX`: nothing is ever written -> zeros here (reference: uninitialised Mat).
static int run_ncc(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume)
{
    if (f->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (mp.disparity_type != ASW_DISPARITY_LEFT && mp.disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    if (mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;
    const int H = f->rows, W = f->cols;
    const size_t plane = (size_t)H * W;
    ASW_TRY(f->disp.ensure(plane * 4));
    f->vol_floats = 0;
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    if (mp.disparity_type == ASW_DISPARITY_RIGHT) {
        ASW_HIP_TRY(hipMemsetAsync(f->disp.p, 0, plane * 4, ctx->stream));
        if (keep_volume) return ASW_ERR_UNSUPPORTED_LAYOUT;
    } else {
        float* vol = nullptr;
        if (keep_volume) {  // raw (un-normalised) costs of all numD offsets, for inspection
            ASW_TRY(f->vol.ensure(plane * mp.numD * 4));
            f->vol_floats = plane * mp.numD;
            vol = f->vol.as<float>();
        }
        ASW_TRY(run_ncc_cost(ctx, f->L.as<uint8_t>(), f->R.as<uint8_t>(), H, W, mp.disparity_type, mp.win, mp.minD, mp.numD, vol,
                             f->disp.as<float>(), mp.numD - 1));
    }
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = 1;
    return ASW_OK;
}

enum GuidedKind { GUIDED_SAD6 = 0 /* GuidedF */, GUIDED_SIM3 = 1 /* GuidedF_2 */, GUIDED_NCC = 2 /* GuidedF_3 */ };

static int run_guided(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume, int kind)
{
    const bool variant2 = kind == GUIDED_SIM3;
    // GuidedF_3 + DISPARITY_RIGHT: getGuidedFilter receives the plain right image (M.cpp:3110), a 3-channel guide
    const bool ncc = kind == GUIDED_NCC, plain3 = variant2 || (ncc && mp.disparity_type == ASW_DISPARITY_RIGHT);

    if (f->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;
    // GuidedF_2: RIGHT / gray branches of computeSimilarity throw in the reference (App. B-7).
    if (variant2 && mp.disparity_type != ASW_DISPARITY_LEFT) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (mp.disparity_type != ASW_DISPARITY_LEFT && mp.disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    const bool right = mp.disparity_type == ASW_DISPARITY_RIGHT;
    if (!variant2 && mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // getCostSAD_d, M.cpp:2458-2462; computeNCC, M.cpp:939-942
    if (mp.win < 1 || mp.win > 128) return ASW_ERR_BAD_ARGUMENT;
    const int H = f->rows, W = f->cols, n = mp.numD, C = plain3 ? 3 : 6;
    const size_t plane = (size_t)H * W;
    DevBuf& raw = ctx->buf("g_raw");
    DevBuf& ord = ctx->buf("g_ord");
    DevBuf& psc = ctx->buf("g_pscales");
    DevBuf& gsc = ctx->buf("g_gscales");
    DevBuf& stats = ctx->buf("g_stats");
    DevBuf& ab = ctx->buf("g_ab");
    DevBuf& pxa = ctx->buf("bgrxL");
    DevBuf& pxb = ctx->buf("bgrxR");
    const int nstat = plain3 ? 1 : n;
    ASW_TRY(raw.ensure(plane * n * 4));
    DevBuf& parts = ctx->buf("g_parts");
    ASW_TRY(parts.ensure(similarity_parts_words(H, W, n) * 4));
    ASW_TRY(ord.ensure((size_t)(2 * n + 2) * 4));
    ASW_TRY(psc.ensure((size_t)n * sizeof(float2)));
    ASW_TRY(gsc.ensure((size_t)n * sizeof(float2)));
    ASW_TRY(stats.ensure(guided_stats_floats(C, nstat, H, W) * 4));
    ASW_TRY(ab.ensure(guided_ab_floats(C, n, H, W) * 4));
    ASW_TRY(pxa.ensure(plane * 4));
    ASW_TRY(pxb.ensure(plane * 4));
    ASW_TRY(f->vol.ensure(plane * n * 4));  // q volume: always needed for the WTA pass
    ASW_TRY(f->disp.ensure(plane * 4));
    f->vol_floats = keep_volume ? plane * n : 0;
    const uint8_t* dL = f->L.as<uint8_t>();
    const uint8_t* dR = f->R.as<uint8_t>();

    GuidedLaunch a;
    // guide = [L, R shifted by -d] (LEFT, M.cpp:2907-2912) or [L shifted by +d, R] (RIGHT, M.cpp:2925-2929)
    a.shiftA = (!plain3 && right) ? 1 : 0; a.shiftB = (!plain3 && !right) ? -1 : 0; a.C = C; a.guide_per_slice = plain3 ? 0 : 1;
    const uint8_t* dGuide3 = variant2 ? dL : dR;  // the 3-channel guide: left image (GuidedF_2) / right image (GuidedF_3 RIGHT)
    ASW_TRY(launch_pack_words(ctx->stream, plain3 ? dGuide3 : dL, H, W, 3, 0, pxa.as<uint32_t>()));
    if (!plain3) ASW_TRY(launch_pack_words(ctx->stream, dR, H, W, 3, 0, pxb.as<uint32_t>()));
    a.guideA = pxa.as<uint32_t>(); a.guideB = plain3 ? nullptr : pxb.as<uint32_t>();
    if (ncc) {
        // costs_ds of computeNCC (M.cpp:3076): raw planes, then normalize(NORM_MINMAX) of every plane in place
        ASW_TRY(run_ncc_cost(ctx, dL, dR, H, W, mp.disparity_type, mp.win, mp.minD, n, raw.as<float>(), nullptr, 0));
        ASW_TRY(launch_slice_scales(ctx->stream, raw.as<float>(), n, plane, ord.as<uint32_t>(), psc.as<float2>()));
        ASW_TRY(launch_apply_scales(ctx->stream, raw.as<float>(), n, plane, psc.as<float2>()));
        if (plain3) {
            ASW_TRY(launch_u8_scale(ctx->stream, dGuide3, plane * 3, ord.as<uint32_t>() + 2 * n, gsc.as<float2>()));
        } else {
            DevBuf& colmm = ctx->buf("g_colmm");
            ASW_TRY(colmm.ensure((size_t)2 * W * sizeof(int)));
            ASW_TRY(launch_guide_scales_lr(ctx->stream, dL, dR, H, W, mp.minD, n, mp.disparity_type, ord.as<uint32_t>() + 2 * n,
                                           colmm.as<int>(), gsc.as<float2>()));
        }
    } else if (variant2) {
        ASW_TRY(build_similarity_volume(ctx, dL, dR, H, W, mp.minD, n, 0.4, 10, 50, raw.as<float>(), parts.as<uint32_t>(),
                                        psc.as<float2>()));  // M.cpp:2990 (+ the min/max of M.cpp:2775, fused)
        ASW_TRY(launch_u8_scale(ctx->stream, dL, plane * 3, ord.as<uint32_t>() + 2 * n, gsc.as<float2>()));
    } else {
        DevBuf& gl = ctx->buf("grayL");
        DevBuf& gr = ctx->buf("grayR");
        DevBuf& colmm = ctx->buf("g_colmm");
        ASW_TRY(gl.ensure(plane));
        ASW_TRY(gr.ensure(plane));
        ASW_TRY(colmm.ensure((size_t)2 * W * sizeof(int)));
        ASW_TRY(launch_bgr2gray(ctx->stream, dL, H, W, gl.as<uint8_t>()));
        ASW_TRY(launch_bgr2gray(ctx->stream, dR, H, W, gr.as<uint8_t>()));
        ASW_TRY(launch_cost_sad(ctx->stream, gl.as<uint8_t>(), gr.as<uint8_t>(), H, W, mp.disparity_type, mp.win, mp.minD, n,
                                raw.as<float>()));  // M.cpp:2884-2889
        ASW_TRY(launch_guide_scales_lr(ctx->stream, right ? dR : dL, right ? dL : dR, H, W, mp.minD, n, mp.disparity_type,
                                       ord.as<uint32_t>() + 2 * n, colmm.as<int>(), gsc.as<float2>()));
    }
    if (!variant2)
        ASW_TRY(launch_slice_scales(ctx->stream, raw.as<float>(), n, plane, ord.as<uint32_t>(), psc.as<float2>()));  // M.cpp:2775
    a.gscales = gsc.as<float2>(); a.P = raw.as<float>(); a.pscales = psc.as<float2>();
    a.H = H; a.W = W; a.n = n; a.r = mp.win; a.minD = mp.minD; a.eps = mp.eps;
    DevBuf& repb = ctx->buf("g_rep");
    ASW_TRY(repb.ensure((size_t)n * sizeof(int)));
    a.stats = stats.as<float>(); a.rep_scratch = repb.as<int>(); a.ab = ab.as<float>(); a.q = f->vol.as<float>();
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    ASW_TRY(launch_guided(ctx->stream, a));
    ASW_TRY(launch_wta(ctx->stream, f->vol.as<float>(), n, H, W, mp.minD, f->disp.as<float>()));  // M.cpp:3032-3048
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = plain3 ? 4 : 5;
    return ASW_OK;
}

// ------------------------------------------------------------------------------------------
// geodesic ASW: computeAdaptiveWeight_geodesic (M.cpp:1436-1534)
// ------------------------------------------------------------------------------------------
static int run_geodesic(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume)
{
    if (mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // M.cpp:1440-1443
    if (f->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;  // at<Vec3b>
    if (mp.disparity_type != ASW_DISPARITY_LEFT && mp.disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    const int flip = mp.disparity_type == ASW_DISPARITY_RIGHT ? 1 : 0;  // M.cpp:1498-1520 == LEFT on the mirrored problem
    if (mp.win < 1 || mp.win > 35) return ASW_ERR_BAD_ARGUMENT;
    const int H = f->rows, W = f->cols, nD = mp.numD + 1;  // inclusive range, M.cpp:1447,1467
    const size_t plane = (size_t)H * W, cells = (size_t)mp.win * mp.win;
    DevBuf& pl = ctx->buf("bgrxL");
    DevBuf& pr = ctx->buf("bgrxR");
    DevBuf& wl = ctx->buf("geoWL");
    DevBuf& wr = ctx->buf("geoWR");
    ASW_TRY(pl.ensure(plane * 4));
    ASW_TRY(pr.ensure(plane * 4));
    ASW_TRY(wl.ensure(plane * cells * 2));
    ASW_TRY(wr.ensure(plane * cells * 2));
    ASW_TRY(f->disp.ensure(plane * 4));
    f->vol_floats = 0;
    if (keep_volume) {
        ASW_TRY(f->vol.ensure(plane * nD * 4));
        f->vol_floats = plane * nD;
    }
    ASW_TRY(launch_pack_bgrx(ctx->stream, f->L.as<uint8_t>(), H, W, pl.as<uint32_t>()));
    ASW_TRY(launch_pack_bgrx(ctx->stream, f->R.as<uint8_t>(), H, W, pr.as<uint32_t>()));
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    ASW_TRY(launch_geodesic_weights_u16(ctx->stream, pl.as<uint32_t>(), H, W, mp.win, 3, wl.as<uint16_t>()));  // M.cpp:1464
    ASW_TRY(launch_geodesic_weights_u16(ctx->stream, pr.as<uint32_t>(), H, W, mp.win, 3, wr.as<uint16_t>()));  // M.cpp:1465
    double* partE = nullptr;
    float* partD = nullptr;
    if (plane <= (size_t)1 << 20) {  // small frames only: scratch for the grid.z split of the disparity range
        DevBuf& pe = ctx->buf("bil_partE");
        DevBuf& pd = ctx->buf("bil_partD");
        ASW_TRY(pe.ensure((size_t)8 * plane * sizeof(double)));
        ASW_TRY(pd.ensure((size_t)8 * plane * sizeof(float)));
        partE = pe.as<double>(); partD = pd.as<float>();
    }
    if (!flip)
        ASW_TRY(launch_asw_geodesic(ctx->stream, pl.as<uint32_t>(), pr.as<uint32_t>(), wl.as<uint16_t>(), wr.as<uint16_t>(), H, W,
                                    mp.win, mp.minD, nD, 0, keep_volume ? f->vol.as<float>() : nullptr, f->disp.as<float>(),
                                    partE, partD));
    else
        ASW_TRY(launch_asw_geodesic(ctx->stream, pr.as<uint32_t>(), pl.as<uint32_t>(), wr.as<uint16_t>(), wl.as<uint16_t>(), H, W,
                                    mp.win, mp.minD, nD, 1, keep_volume ? f->vol.as<float>() : nullptr, f->disp.as<float>(),
                                    partE, partD));
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = 3;
    return ASW_OK;
}

// ------------------------------------------------------------------------------------------
// weighted-median ASW: computeAdaptiveWeight_WeightedMedian (M.cpp:3228-3383)
// ------------------------------------------------------------------------------------------
static int ensure_wmedian_tables(asw_ctx* ctx, int win, double rate_s, double rate_r)
{
    if (ctx->wm_rate_r != rate_r || !ctx->wm_lut2.p) {
        // computeColorWeightGau: exp((d0+d1+d2)/rateR*(-1)) == exp(addWeighted(d0+d1, a, d2, a)) with
        // a = (float)(-1/rateR) (M.cpp:3177-3179); cv::exp restated as expf
        std::vector<float> lut((size_t)511 * 256);
        const float al = (float)((1.0 / rate_r) * (-1.0));
        for (int m = 0; m < 511; m++)
            for (int c = 0; c < 256; c++) {
                float arg = (float)m * al + (float)c * al;
                lut[(size_t)m * 256 + c] = expf(arg);
            }
        ASW_TRY(ctx->wm_lut2.ensure(lut.size() * 4));
        ASW_HIP_TRY(hipMemcpyAsync(ctx->wm_lut2.p, lut.data(), lut.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->wm_rate_r = rate_r;
    }
    if (ctx->wm_rate_s != rate_s || ctx->wm_win != win || !ctx->wm_wd.p) {
        // computeSpaceWeightGau (M.cpp:3207-3226)
        const int h = win / 2;
        std::vector<float> wd((size_t)win * win);
        const float al = (float)((1.0 / rate_s) * (-1.0));
        for (int y = 0; y < win; y++)
            for (int x = 0; x < win; x++) {
                float v = (float)((x - h) * (x - h)) + (float)((y - h) * (y - h));
                wd[(size_t)x * win + y] = expf(v * al);
            }
        ASW_TRY(ctx->wm_wd.ensure(wd.size() * 4));
        ASW_HIP_TRY(hipMemcpyAsync(ctx->wm_wd.p, wd.data(), wd.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->wm_rate_s = rate_s;
        ctx->wm_win = win;
    }
    return ASW_OK;
}

static int run_wmedian(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume)
{
    if (mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // M.cpp:3238-3241
    if (f->channels != 3) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (mp.disparity_type != ASW_DISPARITY_LEFT) return ASW_ERR_UNSUPPORTED_LAYOUT;  // App. B-7 / B-13
    if (mp.win < 1 || mp.win > 45) return ASW_ERR_BAD_ARGUMENT;  // 256-slot fast network up to 15x15, 64-bit general path up to 2048 slots
    const int H = f->rows, W = f->cols, n = mp.numD, cells = mp.win * mp.win;
    const int max_off = mp.minD + mp.numD - 1, Wb = W + max_off;
    const size_t plane = (size_t)H * W;
    ASW_TRY(ensure_wmedian_tables(ctx, mp.win, mp.rate_s, mp.rate_r));
    DevBuf& raw = ctx->buf("g_raw");
    DevBuf& wl = ctx->buf("wmWL");
    DevBuf& wr = ctx->buf("wmWR");
    ASW_TRY(raw.ensure(plane * n * 4));
    ASW_TRY(wl.ensure(plane * cells * 4));
    ASW_TRY(wr.ensure((size_t)H * Wb * cells * 4));
    ASW_TRY(f->vol.ensure(plane * n * 4));
    ASW_TRY(f->disp.ensure(plane * 4));
    f->vol_floats = keep_volume ? plane * n : 0;
    const uint8_t* dL = f->L.as<uint8_t>();
    const uint8_t* dR = f->R.as<uint8_t>();
    ASW_TRY(build_similarity_volume(ctx, dL, dR, H, W, mp.minD, n, 0.4, 10, 50, raw.as<float>()));  // M.cpp:3250
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    ASW_TRY(launch_wm_weights(ctx->stream, dL, H, W, 0, mp.win, ctx->wm_lut2.as<float>(), ctx->wm_wd.as<float>(), wl.as<float>()));
    ASW_TRY(launch_wm_weights(ctx->stream, dR, H, W, max_off, mp.win, ctx->wm_lut2.as<float>(), nullptr, wr.as<float>()));
    ASW_TRY(launch_wmedian(ctx->stream, raw.as<float>(), wl.as<float>(), wr.as<float>(), H, W, mp.win, n, max_off,
                           f->vol.as<float>()));
    ASW_TRY(launch_wta(ctx->stream, f->vol.as<float>(), n, H, W, mp.minD, f->disp.as<float>()));  // M.cpp:3365-3381
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = 4;
    return ASW_OK;
}

// ------------------------------------------------------------------------------------------
// O(1)-bilateral ASW: computeAdaptiveWeight_BLO1 (M.cpp:2505-2725)
// ------------------------------------------------------------------------------------------
static int run_blo1(asw_ctx* ctx, Frame* f, const MatchParams& mp, bool keep_volume)
{
    if (mp.win % 2 == 0) return ASW_ERR_EVEN_WINDOW;  // getCostSAD_d -> Mat(), M.cpp:2458-2462
    if (f->channels != 3 && f->channels != 1) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (mp.disparity_type != ASW_DISPARITY_LEFT && mp.disparity_type != ASW_DISPARITY_RIGHT) return ASW_ERR_BAD_ARGUMENT;
    // the reference indexes setsJB_ks_ds_x[key][offset] with the ABSOLUTE offset (M.cpp:2659): out of range unless 0
    if (mp.minD != 0) return ASW_ERR_BAD_ARGUMENT;
    if (mp.win < 1 || mp.win > 64) return ASW_ERR_BAD_ARGUMENT;
    const int step = (int)(256 * mp.blo_rate_r);  // M.cpp:2550
    if (step <= 0) return ASW_ERR_BAD_ARGUMENT;   // the reference's key loop would not terminate
    const int H = f->rows, W = f->cols, n = mp.numD;
    const size_t plane = (size_t)H * W;
    // keys 0, step, 2*step, ..., 255 (M.cpp:2551-2560) are implied by `step` in the kernel
    DevBuf& gl = ctx->buf("grayL");
    DevBuf& gr = ctx->buf("grayR");
    DevBuf& raw = ctx->buf("g_raw");
    ASW_TRY(gl.ensure(plane));
    ASW_TRY(gr.ensure(plane));
    ASW_TRY(raw.ensure(plane * n * 4));
    ASW_TRY(f->disp.ensure(plane * 4));
    f->vol_floats = 0;
    if (keep_volume) {
        ASW_TRY(f->vol.ensure(plane * n * 4));
        f->vol_floats = plane * n;
    }
    if (f->channels == 3) {  // M.cpp:2514-2521
        ASW_TRY(launch_bgr2gray(ctx->stream, f->L.as<uint8_t>(), H, W, gl.as<uint8_t>()));
        ASW_TRY(launch_bgr2gray(ctx->stream, f->R.as<uint8_t>(), H, W, gr.as<uint8_t>()));
    } else {
        ASW_HIP_TRY(hipMemcpyAsync(gl.p, f->L.p, plane, hipMemcpyDeviceToDevice, ctx->stream));
        ASW_HIP_TRY(hipMemcpyAsync(gr.p, f->R.p, plane, hipMemcpyDeviceToDevice, ctx->stream));
    }
    ASW_TRY(launch_cost_sad(ctx->stream, gl.as<uint8_t>(), gr.as<uint8_t>(), H, W, mp.disparity_type, mp.win, mp.minD, n,
                            raw.as<float>()));  // M.cpp:2529-2547
    ASW_HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    ASW_TRY(launch_blo1(ctx->stream, gl.as<uint8_t>(), gr.as<uint8_t>(), raw.as<float>(), step, H, W, mp.disparity_type, mp.win, n,
                        keep_volume ? f->vol.as<float>() : nullptr, f->disp.as<float>()));
    ASW_HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->timing.aggregate_launches = 1;
    return ASW_OK;
}

static int run_method(asw_ctx* ctx, Frame* f, int algorithm, const MatchParams& mp, bool keep_volume, bool sync = true)
{
    if (mp.numD <= 0 || mp.minD < 0) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    ASW_HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
    int rc;
    switch (algorithm) {  // M.cpp:49-87
    case ASW_ALG_ADAPTIVE_WEIGHT: rc = run_bilateral(ctx, f, mp, keep_volume); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_8DIRECT: rc = run_bilateral(ctx, f, mp, keep_volume, true); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_GEODESIC: rc = run_geodesic(ctx, f, mp, keep_volume); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_BLO1: rc = run_blo1(ctx, f, mp, keep_volume); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER: rc = run_guided(ctx, f, mp, keep_volume, GUIDED_SAD6); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_2: rc = run_guided(ctx, f, mp, keep_volume, GUIDED_SIM3); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_3: rc = run_guided(ctx, f, mp, keep_volume, GUIDED_NCC); break;
    case ASW_ALG_NCC: rc = run_ncc(ctx, f, mp, keep_volume); break;
    case ASW_ALG_ADAPTIVE_WEIGHT_MEDIAN: rc = run_wmedian(ctx, f, mp, keep_volume); break;
    default: rc = ASW_ERR_UNSUPPORTED_METHOD; break;
    }
    if (rc != ASW_OK) return rc;
    ASW_HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    if (!sync) return ASW_OK;  // pipelined callers (batch scheduler) order and wait on the stream themselves
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    float t = 0;
    ASW_HIP_TRY(hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[1]));
    ctx->timing.total_ms = t;
    ASW_HIP_TRY(hipEventElapsedTime(&t, ctx->ev[2], ctx->ev[3]));
    ctx->timing.aggregate_ms = t;
    ctx->timing.cost_ms = ctx->timing.total_ms - ctx->timing.aggregate_ms;
    return ASW_OK;
}

extern "C" int asw_match_resident(asw_ctx* ctx, int slot, int disparity_type, int algorithm, int win_size,
                                  int min_disparity, int num_disparity, int keep_volume)
{
    if (!ctx) return ASW_ERR_BAD_ARGUMENT;
    Frame* f = frame_slot(ctx, slot, false);
    if (!f || !f->valid) return ASW_ERR_NO_FRAME;
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    return run_method(ctx, f, algorithm, mp, keep_volume != 0);
}

// ---- internal hooks of the batch scheduler (batch.hip): device buffers of a slot, enqueue without waiting ----
int asw_internal_stage_slot(asw_ctx* ctx, int slot, int rows, int cols, int channels, Frame** out)
{
    Frame* f = frame_slot(ctx, slot, true);
    if (!f) return ASW_ERR_BAD_ARGUMENT;
    const size_t bytes = (size_t)rows * cols * channels;
    ASW_TRY(f->L.ensure(bytes));
    ASW_TRY(f->R.ensure(bytes));
    ASW_TRY(f->disp.ensure((size_t)rows * cols * 4));
    f->rows = rows; f->cols = cols; f->channels = channels; f->valid = true;
    *out = f;
    return ASW_OK;
}

int asw_internal_enqueue_match(asw_ctx* ctx, int slot, int disparity_type, int algorithm, int win_size, int min_disparity,
                               int num_disparity)
{
    Frame* f = frame_slot(ctx, slot, false);
    if (!f || !f->valid) return ASW_ERR_NO_FRAME;
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    return run_method(ctx, f, algorithm, mp, false, false);
}

int asw_internal_check_pair(const asw_image* l, const asw_image* r, const asw_image* d)
{
    ASW_TRY(check_pair(l, r));
    return check_disp_out(d, l->rows, l->cols);
}

static int match_host(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp, int algorithm,
                      const MatchParams& mp, float* cost_volume_out)
{
    if (!ctx) return ASW_ERR_BAD_ARGUMENT;
    ASW_TRY(check_pair(left, right));
    ASW_TRY(check_disp_out(disp, left->rows, left->cols));
    const int slot = 0;
    ASW_TRY(asw_upload_pair(ctx, slot, left, right));
    Frame* f = frame_slot(ctx, slot, false);
    ASW_TRY(run_method(ctx, f, algorithm, mp, cost_volume_out != nullptr));
    ASW_TRY(asw_download_disparity(ctx, slot, disp));
    if (cost_volume_out) ASW_TRY(asw_download_volume(ctx, slot, cost_volume_out, f->vol_floats));
    return ASW_OK;
}

extern "C" int asw_stereo_match(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                int disparity_type, int algorithm, int win_size, int min_disparity,
                                int num_disparity, float* cost_volume_out)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    return match_host(ctx, left, right, disp, algorithm, mp, cost_volume_out);
}

extern "C" int asw_aggregate_bilateral(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                       double gamma_c, double gamma_g, int disparity_type, int win_size,
                                       int min_disparity, int num_disparity, float* cost_volume_out)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    mp.gamma_c = gamma_c; mp.gamma_g = gamma_g;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT, mp, cost_volume_out);
}

extern "C" int asw_aggregate_direct8(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                     int disparity_type, int win_size, int min_disparity, int num_disparity,
                                     float* cost_volume_out)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_8DIRECT, mp, cost_volume_out);
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
    ASW_TRY(launch_bgr2gray(ctx->stream, d.as<uint8_t>(), bgr->rows, bgr->cols, g.as<uint8_t>()));
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


extern "C" int asw_aggregate_guided(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                    int disparity_type, double eps, int win_size, int min_disparity, int num_disparity,
                                    float* cost_volume_out)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity; mp.eps = eps;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER, mp, cost_volume_out);
}

extern "C" int asw_aggregate_guided2(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                     int disparity_type, double eps, int win_size, int min_disparity, int num_disparity,
                                     float* cost_volume_out)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity; mp.eps = eps;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_2, mp, cost_volume_out);
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
        ASW_TRY(launch_bgr2gray(ctx->stream, pl, H, W, gl.as<uint8_t>()));
        ASW_TRY(launch_bgr2gray(ctx->stream, pr, H, W, gr.as<uint8_t>()));
        pl = gl.as<uint8_t>(); pr = gr.as<uint8_t>();
    }
    ASW_TRY(launch_cost_sad(ctx->stream, pl, pr, H, W, disparity_type, win_size, min_disparity, n, raw.as<float>()));
    ASW_HIP_TRY(hipMemcpyAsync(cost, raw.p, (size_t)n * H * W * 4, hipMemcpyDeviceToHost, ctx->stream));
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
    return match_host(ctx, left, right, disp, ASW_ALG_NCC, mp, nullptr);
}

extern "C" int asw_aggregate_guided3(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                     int disparity_type, double eps, int win_size, int min_disparity, int num_disparity,
                                     float* cost_volume_out)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity; mp.eps = eps;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_3, mp, cost_volume_out);
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
    ASW_TRY(ab.ensure(guided_ab_floats(C, 1, H, W) * 4));
    ASW_TRY(qv.ensure(plane * 4));
    ASW_TRY(pxa.ensure(plane * 4));
    ASW_TRY(pxb.ensure(plane * 4));
    ASW_TRY(launch_pack_words(ctx->stream, dg.as<uint8_t>(), H, W, C, 0, pxa.as<uint32_t>()));
    if (C == 6) ASW_TRY(launch_pack_words(ctx->stream, dg.as<uint8_t>(), H, W, C, 1, pxb.as<uint32_t>()));
    ASW_HIP_TRY(hipMemcpyAsync(raw.p, p, plane * 4, hipMemcpyHostToDevice, ctx->stream));
    ASW_TRY(launch_u8_scale(ctx->stream, dg.as<uint8_t>(), plane * C, ord.as<uint32_t>() + 2, gsc.as<float2>()));  // M.cpp:2774
    ASW_TRY(launch_slice_scales(ctx->stream, raw.as<float>(), 1, plane, ord.as<uint32_t>(), psc.as<float2>()));     // M.cpp:2775
    GuidedLaunch a;
    a.shiftA = 0; a.shiftB = 0; a.C = C; a.guide_per_slice = 0;
    a.guideA = pxa.as<uint32_t>(); a.guideB = C == 6 ? pxb.as<uint32_t>() : nullptr;
    a.gscales = gsc.as<float2>(); a.P = raw.as<float>(); a.pscales = psc.as<float2>();
    a.H = H; a.W = W; a.n = 1; a.r = r; a.minD = 0; a.eps = eps;
    a.stats = stats.as<float>(); a.rep_scratch = nullptr; a.ab = ab.as<float>(); a.q = qv.as<float>();
    ASW_TRY(launch_guided(ctx->stream, a));
    ASW_HIP_TRY(hipMemcpyAsync(q, qv.p, plane * 4, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_aggregate_geodesic(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                      int disparity_type, int win_size, int min_disparity, int num_disparity,
                                      float* cost_volume_out)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_GEODESIC, mp, cost_volume_out);
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
                                  int num_disparity, float* cost_volume_out)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    mp.blo_rate_r = sample_rate_r;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_BLO1, mp, cost_volume_out);
}

extern "C" int asw_aggregate_wmedian(asw_ctx* ctx, const asw_image* left, const asw_image* right, asw_image* disp,
                                     int disparity_type, int win_size, double rate_s, double rate_r, int min_disparity,
                                     int num_disparity, float* cost_volume_out)
{
    MatchParams mp;
    mp.disparity_type = disparity_type; mp.win = win_size; mp.minD = min_disparity; mp.numD = num_disparity;
    mp.rate_s = rate_s; mp.rate_r = rate_r;
    return match_host(ctx, left, right, disp, ASW_ALG_ADAPTIVE_WEIGHT_MEDIAN, mp, cost_volume_out);
}
