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
};

struct BilateralTables {  // cached per (win, gamma_c, gamma_g)
    int win = 0;
    double gamma_c = 0, gamma_g = 0;
    int ntaps = 0, ncls = 0;
    DevBuf taps;  // int4 per tap
    DevBuf lut;   // float [ncls][256]
};

struct asw_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<Frame> frames;
    std::map<std::string, DevBuf> scratch;  // named grow-only scratch buffers
    BilateralTables bil;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // total start/stop, aggregate start/stop
    asw_timing timing = {0, 0, 0, 0};
    DevBuf& buf(const char* name) { return scratch[name]; }
};

// ---- kernel launchers (each returns an asw_status; all work is enqueued on `s`) ----
int launch_bgr2gray(hipStream_t s, const uint8_t* bgr, int H, int W, uint8_t* gray);
int launch_cost_ad(hipStream_t s, const uint8_t* L, const uint8_t* R, int H, int W, int C, int disp_type, int minD,
                   int numD, int do_thresh /* 0: AD, 1: TAD mask */, int threshold, uint8_t* cost);
int launch_wta(hipStream_t s, const float* vol, int n, int H, int W, int minD, float* disp);

struct BilateralLaunch {
    const uint8_t* gL;
    const uint8_t* gR;
    int H, W, win, minD, nD;  // nD = number of candidates (numD + 1 for the reference's inclusive range)
    const int4* taps;         // {sample cell offset dys*LW+dxs, weight dx, weight dy, class*256}
    const float* lut;         // [ncls][256]
    int ntaps;
    float* vol;  // optional [nD][H][W]
    float* disp; // [H][W]
};
int launch_bilateral(hipStream_t s, const BilateralLaunch& a);
int bilateral_lds_row_stride(int win);  // LW of the kernel's sample tile (taps[].x is expressed in it)
