// Context, frame slots and host <-> HBM plumbing of the C-ABI (include/asw_mi355x.h); the driver-side pre/post-processing
// entry points live here too because they only move and convert images.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "asw_internal.h"
#include "asw_host.h"

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

void AswTuning::read_environment()
{
    auto num = [](const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; };
    bilateral_xq = num("ASW_BILATERAL_XQ", bilateral_xq);
    geodesic_xq = num("ASW_GEODESIC_XQ", geodesic_xq);
    wmedian_tile = num("ASW_WMEDIAN_TILE", wmedian_tile);
    wmedian_tile_chunk = num("ASW_WMEDIAN_TILE_CHUNK", wmedian_tile_chunk);
    wmedian_tile_split = num("ASW_WMEDIAN_TILE_SPLIT", wmedian_tile_split);
    wmedian_gen_rows = num("ASW_WMEDIAN_GEN_ROWS", wmedian_gen_rows);
    band_ab = num("ASW_BAND_AB", band_ab);
    band_q = num("ASW_BAND_Q", band_q);
    ring_ab = num("ASW_RING_AB", ring_ab);
    ring_q = num("ASW_RING_Q", ring_q);
    q_wg_strips = num("ASW_Q_WG_STRIPS", q_wg_strips);
    guided_fused = num("ASW_GUIDED_FUSED", guided_fused);
    q6_pair = num("ASW_Q6_PAIR", q6_pair);
    ab6_pair = num("ASW_AB6_PAIR", ab6_pair);
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
    c->tune.read_environment();  // the only place the library reads its switches
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return ASW_ERR_HIP;
    }
    for (int i = 0; i < 4; i++)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) {
            delete c;
            return ASW_ERR_HIP;
        }
    bool ok = true;
    for (int i = 0; i < 2; i++) ok = ok && hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 3; i++) ok = ok && hipEventCreateWithFlags(&c->aux_ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        asw_destroy(c);
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
    ctx->host_frame.L.release(); ctx->host_frame.R.release(); ctx->host_frame.disp.release(); ctx->host_frame.vol.release();
    for (auto& kv : ctx->scratch) kv.second.release();
    ctx->bil.taps.release();
    ctx->bil.lut.release();
    ctx->bil.cells.release();
    ctx->wm_lut2.release();
    ctx->wm_wd.release();
    for (int i = 0; i < 4; i++)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < 3; i++)
        if (ctx->aux_ev[i]) (void)hipEventDestroy(ctx->aux_ev[i]);
    for (int i = 0; i < 2; i++)
        if (ctx->aux[i]) {
            (void)hipStreamSynchronize(ctx->aux[i]);
            (void)hipStreamDestroy(ctx->aux[i]);
        }
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

// cvtColor(COLOR_BGR2GRAY / RGB2GRAY) constant set of every method that converts to gray (M.cpp:1031-1033, 2448-2454, 835-840)
extern "C" int asw_set_gray_bits(asw_ctx* ctx, int bits)
{
    if (!ctx || (bits != 14 && bits != 15)) return ASW_ERR_BAD_ARGUMENT;
    ctx->gray_bits = bits;
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
int check_u8_image(const asw_image* im)
{
    if (!im || !im->data || im->rows <= 0 || im->cols <= 0) return ASW_ERR_BAD_ARGUMENT;
    if (im->depth != ASW_8U) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (im->channels != 1 && im->channels != 3 && im->channels != 6) return ASW_ERR_UNSUPPORTED_LAYOUT;
    if (im->step < (size_t)im->cols * im->channels) return ASW_ERR_BAD_ARGUMENT;
    return ASW_OK;
}

int check_pair(const asw_image* L, const asw_image* R)
{
    ASW_TRY(check_u8_image(L));
    ASW_TRY(check_u8_image(R));
    // leftImg.size != rightImg.size -> silent return in the reference (M.cpp:217-220)
    if (L->rows != R->rows || L->cols != R->cols) return ASW_ERR_SIZE_MISMATCH;
    if (L->channels != R->channels) return ASW_ERR_UNSUPPORTED_LAYOUT;
    return ASW_OK;
}

// Pitched host image <-> dense device plane.  Rows without padding are one 1-D copy.  hipMemcpy2DAsync on pageable memory takes
// milliseconds when the row length is not a multiple of the DMA granule (1242 x 3 bytes per row: 6.7 ms per call for the three
// copies of a KITTI-shape frame, against 0.9 ms at 1920 x 3), whatever the pitch -- so padded rows are packed / unpacked on the
// host through a dense staging buffer of the context and travel as a 1-D copy as well.
static hipError_t copy_rows(asw_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t rowbytes, size_t rows,
                            hipMemcpyKind kind)
{
    hipStream_t s = ctx->stream;
    if (dpitch == rowbytes && spitch == rowbytes) return hipMemcpyAsync(dst, src, rowbytes * rows, kind, s);
    ctx->host_pack.resize(rowbytes * rows);
    unsigned char* pack = ctx->host_pack.data();
    if (kind == hipMemcpyHostToDevice) {  // the device side is dense
        for (size_t y = 0; y < rows; y++) memcpy(pack + y * rowbytes, (const unsigned char*)src + y * spitch, rowbytes);
        hipError_t e = hipMemcpyAsync(dst, pack, rowbytes * rows, kind, s);
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(s);  // the staging buffer is reused by the next image
    }
    hipError_t e = hipMemcpyAsync(pack, src, rowbytes * rows, kind, s);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    for (size_t y = 0; y < rows; y++) memcpy((unsigned char*)dst + y * dpitch, pack + y * rowbytes, rowbytes);
    return hipSuccess;
}

int upload_image(asw_ctx* ctx, const asw_image* im, DevBuf& dst)
{
    size_t rowbytes = (size_t)im->cols * im->channels;
    ASW_TRY(dst.ensure(rowbytes * im->rows));
    ASW_HIP_TRY(copy_rows(ctx, dst.p, rowbytes, im->data, im->step, rowbytes, im->rows, hipMemcpyHostToDevice));
    return ASW_OK;
}

int check_disp_out(const asw_image* d, int rows, int cols)
{
    if (!d || !d->data) return ASW_ERR_BAD_ARGUMENT;
    if (d->depth != ASW_32F || d->channels != 1) return ASW_ERR_BAD_ARGUMENT;
    if (d->rows != rows || d->cols != cols || d->step < (size_t)cols * 4) return ASW_ERR_BAD_ARGUMENT;
    return ASW_OK;
}

Frame* frame_slot(asw_ctx* ctx, int slot, bool create)
{
    if (slot < 0 || slot >= 4096) return nullptr;
    if ((size_t)slot >= ctx->frames.size()) {
        if (!create) return nullptr;
        ctx->frames.resize(slot + 1);
    }
    return &ctx->frames[slot];
}

// uploads a checked pair into `f`; whatever `f` held before (pair, disparity, volume) is gone, also when the upload fails
int upload_pair_into(asw_ctx* ctx, Frame* f, const asw_image* left, const asw_image* right)
{
    f->valid = false;
    f->invalidate_results();
    ASW_TRY(upload_image(ctx, left, f->L));
    ASW_TRY(upload_image(ctx, right, f->R));
    f->rows = left->rows; f->cols = left->cols; f->channels = left->channels;
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));  // host buffers are caller-owned: done with them on return
    f->valid = true;
    return ASW_OK;
}

extern "C" int asw_upload_pair(asw_ctx* ctx, int slot, const asw_image* left, const asw_image* right)
{
    if (!ctx) return ASW_ERR_BAD_ARGUMENT;
    Frame* f = frame_slot(ctx, slot, true);
    if (!f) return ASW_ERR_BAD_ARGUMENT;
    // a rejected pair leaves no stale pair behind either: the slot is empty until a pair has been accepted
    f->valid = false;
    f->invalidate_results();
    ASW_TRY(check_pair(left, right));
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    return upload_pair_into(ctx, f, left, right);
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
    f->invalidate_results();
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
    ASW_HIP_TRY(copy_rows(ctx, left->data, left->step, f->L.p, rowbytes, rowbytes, f->rows, hipMemcpyDeviceToHost));
    ASW_HIP_TRY(copy_rows(ctx, right->data, right->step, f->R.p, rowbytes, rowbytes, f->rows, hipMemcpyDeviceToHost));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_download_disparity_u8(asw_ctx* ctx, int slot, asw_image* disp_u8, int normalize)
{
    if (!ctx || !disp_u8 || !disp_u8->data) return ASW_ERR_BAD_ARGUMENT;
    Frame* f = frame_slot(ctx, slot, false);
    if (!f || !f->valid || !f->has_disp || !f->disp.p || f->disp_rows != f->rows || f->disp_cols != f->cols) return ASW_ERR_NO_FRAME;
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
    ASW_HIP_TRY(copy_rows(ctx, disp_u8->data, disp_u8->step, u8.p, (size_t)f->cols, (size_t)f->cols, f->rows, hipMemcpyDeviceToHost));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

int download_disparity_from(asw_ctx* ctx, Frame* f, asw_image* disp)
{
    if (!f || !f->valid || !f->has_disp || !f->disp.p || f->disp_rows != f->rows || f->disp_cols != f->cols) return ASW_ERR_NO_FRAME;
    ASW_TRY(check_disp_out(disp, f->rows, f->cols));
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    ASW_HIP_TRY(copy_rows(ctx, disp->data, disp->step, f->disp.p, (size_t)f->cols * 4, (size_t)f->cols * 4, f->rows, hipMemcpyDeviceToHost));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_download_disparity(asw_ctx* ctx, int slot, asw_image* disp)
{
    if (!ctx) return ASW_ERR_BAD_ARGUMENT;
    return download_disparity_from(ctx, frame_slot(ctx, slot, false), disp);
}

int download_volume_from(asw_ctx* ctx, Frame* f, float* out, size_t n_floats)
{
    if (!f || !f->valid || !f->has_disp || !f->vol.p || f->vol_floats == 0) return ASW_ERR_NO_FRAME;
    if (n_floats != f->vol_floats) return ASW_ERR_BAD_ARGUMENT;
    ASW_HIP_TRY(hipSetDevice(ctx->device));
    ASW_HIP_TRY(hipMemcpyAsync(out, f->vol.p, n_floats * 4, hipMemcpyDeviceToHost, ctx->stream));
    ASW_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ASW_OK;
}

extern "C" int asw_download_volume(asw_ctx* ctx, int slot, float* out, size_t n_floats)
{
    if (!ctx || !out) return ASW_ERR_BAD_ARGUMENT;
    return download_volume_from(ctx, frame_slot(ctx, slot, false), out, n_floats);
}

