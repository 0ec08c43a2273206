// Streaming kernels: BGR->gray, AD / TAD cost volumes, WTA arg-min.  All HBM-bound, integer-exact.
#include "asw_device.h"
#include "asw_internal.h"

namespace {


// cvtColor(COLOR_BGR2GRAY) 8U, fixed point (M.cpp:1031-1033; App. A-1): gray = (c0*k0 + c1*k1 + c2*k2 + half) >> shift.
// OpenCV 4.1.0 (the reference's pin): 14 bits {1868, 9617, 4899}; later 4.x releases: 15 bits {3735, 19235, 9798}.
__global__ void k_bgr2gray(const uint8_t* __restrict__ bgr, int n, uint8_t* __restrict__ gray, uint32_t k0, uint32_t k1,
                           uint32_t k2, uint32_t shift)
{
    const uint32_t half = 1u << (shift - 1);
    // 4 pixels per thread: 12 input bytes (3 dwords), 1 output dword
    int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(bgr + (size_t)i * 3);
        uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        uint8_t b[12];
        b[0] = w0; b[1] = w0 >> 8; b[2] = w0 >> 16; b[3] = w0 >> 24;
        b[4] = w1; b[5] = w1 >> 8; b[6] = w1 >> 16; b[7] = w1 >> 24;
        b[8] = w2; b[9] = w2 >> 8; b[10] = w2 >> 16; b[11] = w2 >> 24;
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t g = (b[3 * k] * k0 + b[3 * k + 1] * k1 + b[3 * k + 2] * k2 + half) >> shift;
            out |= g << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(gray + i) = out;
    } else {
        for (; i < n; i++) {
            const uint8_t* q = bgr + (size_t)i * 3;
            gray[i] = (uint8_t)((q[0] * k0 + q[1] * k1 + q[2] * k2 + half) >> shift);
        }
    }
}

// u8 MatExpr (c0+c1+c2)/3 == round((min(255,c0+c1)+c2)/3): no exact .5 can occur, so the
// integer form floor((s+1)/3) is identical to OpenCV's float addWeighted + cvRound (App. A-4;
// tests/test_oracle_kat.py::test_ad_float_formula_equals_integer_rule).
__device__ __forceinline__ uint32_t mean3_u8(int c0, int c1, int c2)
{
    int t = min(255, c0 + c1);
    return (uint32_t)((t + c2 + 1) / 3);
}

// computeAD / computeTAD / computeSD (M.cpp:208-292, 304-401, 670-759).  One block per (row, d-slab); the two image
// rows are staged in LDS once and every disparity plane of the slab is produced from them.
// Each thread owns 4 consecutive pixels -> one dword store per plane (256 B per wave-instruction).
template <int C>
__global__ __launch_bounds__(256) void k_cost_ad(const uint8_t* __restrict__ L, const uint8_t* __restrict__ R, int H,
                                                 int W, int disp_type, int minD, int numD, int dPerBlock,
                                                 int do_thresh, int threshold, uint8_t* __restrict__ cost)
{
    extern __shared__ uint8_t srow[];  // [2][W*C]
    uint8_t* sa = srow;                // reference-side row (left for LEFT, right for RIGHT)
    uint8_t* sb = srow + (size_t)W * C;
    const int y = blockIdx.x;
    const uint8_t* ra = (disp_type == ASW_DISPARITY_LEFT ? L : R) + (size_t)y * W * C;
    const uint8_t* rb = (disp_type == ASW_DISPARITY_LEFT ? R : L) + (size_t)y * W * C;
    for (int i = threadIdx.x; i < W * C; i += blockDim.x) {
        sa[i] = ra[i];
        sb[i] = rb[i];
    }
    __syncthreads();
    const int k0 = blockIdx.y * dPerBlock, k1 = min(numD, k0 + dPerBlock);
    const int sgn = disp_type == ASW_DISPARITY_LEFT ? -1 : 1;  // LEFT reads R[x-d], RIGHT reads L[x+d]
    for (int k = k0; k < k1; k++) {
        const int off = minD + k;
        uint8_t* out = cost + ((size_t)k * H + y) * W;
        for (int x4 = threadIdx.x * 4; x4 < W; x4 += blockDim.x * 4) {
            uint32_t packed = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int x = x4 + j;
                if (x < W) {
                    int xb = reflect_idx(x + sgn * off, W);
                    uint32_t v;
                    if (C == 3) {
                        int c0 = abs((int)sa[x * 3] - (int)sb[xb * 3]);
                        int c1 = abs((int)sa[x * 3 + 1] - (int)sb[xb * 3 + 1]);
                        int c2 = abs((int)sa[x * 3 + 2] - (int)sb[xb * 3 + 2]);
                        v = mean3_u8(c0, c1, c2);
                    } else {
                        v = (uint32_t)abs((int)sa[x] - (int)sb[xb]);
                    }
                    if (do_thresh == 1) v = ((int)v > threshold) ? 255u : 0u;  // compare(CMP_GT) -> 0/255 (App. B-4)
                    if (do_thresh == 2) v = min(255u, v * v);                  // u8 Mat::mul saturates, M.cpp:701,718
                    packed |= v << (8 * j);
                }
            }
            if (x4 + 3 < W && ((W & 3) == 0)) {
                *reinterpret_cast<uint32_t*>(out + x4) = packed;
            } else {
                for (int j = 0; j < 4 && x4 + j < W; j++) out[x4 + j] = (uint8_t)(packed >> (8 * j));
            }
        }
    }
}

// WTA (M.cpp:1144-1150, 3032-3048): strict '<' in ascending d against DBL_MAX, NaN never wins,
// never-updated pixels are 0 (build-defined; the reference leaves them uninitialised, App. B-16).
// VEC pixels per thread: 4 (one dwordx4 per plane) for large planes, 1 for planes too small to fill the chip with a quarter
// of a thread per pixel (640x360: 225 workgroups at VEC = 4).
template <int VEC>
__global__ __launch_bounds__(256) void k_wta(const float* __restrict__ vol, int n, size_t plane, int minD,
                                             float* __restrict__ disp)
{
    size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (i4 >= plane) return;
    if (VEC == 4 && i4 + 3 < plane && (plane & 3) == 0) {
        float best[4] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
        bool any[4] = {false, false, false, false};
        float bd[4] = {0, 0, 0, 0};
        for (int k = 0; k < n; k++) {
            float4 v = *reinterpret_cast<const float4*>(vol + (size_t)k * plane + i4);
            float c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // (double)c < DBL_MAX start: every finite or -inf float wins the first time; +inf too
                // (inf < DBL_MAX is false) -> mirror exactly: first candidate must be < DBL_MAX.
                bool better = any[j] ? (c[j] < best[j]) : (c[j] <= 3.402823466e+38f);
                if (better) { best[j] = c[j]; bd[j] = (float)(k + minD); any[j] = true; }
            }
        }
        *reinterpret_cast<float4*>(disp + i4) = make_float4(bd[0], bd[1], bd[2], bd[3]);
    } else {
        for (size_t i = i4; i < plane && i < i4 + VEC; i++) {
            double best = 1.7976931348623157e308;
            float bd = 0.0f;
            for (int k = 0; k < n; k++) {
                double c = (double)vol[(size_t)k * plane + i];
                if (c < best) { best = c; bd = (float)(k + minD); }
            }
            disp[i] = bd;
        }
    }
}

}  // namespace

int launch_bgr2gray(hipStream_t s, const uint8_t* bgr, int H, int W, uint8_t* gray, int bits)
{
    int n = H * W;
    int threads = 256, blocks = (n / 4 + 1 + threads - 1) / threads;
    if (bits == 15)
        hipLaunchKernelGGL(k_bgr2gray, dim3(blocks), dim3(threads), 0, s, bgr, n, gray, 3735u, 19235u, 9798u, 15u);
    else
        hipLaunchKernelGGL(k_bgr2gray, dim3(blocks), dim3(threads), 0, s, bgr, n, gray, 1868u, 9617u, 4899u, 14u);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_rgb2gray(hipStream_t s, const uint8_t* bgr, int H, int W, uint8_t* gray, int bits)
{
    int n = H * W;
    int threads = 256, blocks = (n / 4 + 1 + threads - 1) / threads;
    if (bits == 15)
        hipLaunchKernelGGL(k_bgr2gray, dim3(blocks), dim3(threads), 0, s, bgr, n, gray, 9798u, 19235u, 3735u, 15u);
    else
        hipLaunchKernelGGL(k_bgr2gray, dim3(blocks), dim3(threads), 0, s, bgr, n, gray, 4899u, 9617u, 1868u, 14u);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_cost_ad(hipStream_t s, const uint8_t* L, const uint8_t* R, int H, int W, int C, int disp_type, int minD,
                   int numD, int do_thresh, int threshold, uint8_t* cost)
{
    const int dPerBlock = 16;
    dim3 grid(H, (numD + dPerBlock - 1) / dPerBlock);
    size_t lds = (size_t)2 * W * C;
    if (lds > 160 * 1024) return ASW_ERR_BAD_ARGUMENT;
    if (C == 3)
        hipLaunchKernelGGL(k_cost_ad<3>, grid, dim3(256), lds, s, L, R, H, W, disp_type, minD, numD, dPerBlock, do_thresh, threshold, cost);
    else
        hipLaunchKernelGGL(k_cost_ad<1>, grid, dim3(256), lds, s, L, R, H, W, disp_type, minD, numD, dPerBlock, do_thresh, threshold, cost);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

// Left-right consistency check (the consumer of the DISPARITY_RIGHT maps; the reference has none, so the rule
// is this build's: a left pixel x with disparity d points at right pixel x - d, whose own disparity must agree within max_diff).
__global__ __launch_bounds__(256) void k_lr_check(const float* __restrict__ dl, const float* __restrict__ dr, int H, int W,
                                                  float max_diff, float invalid, float* __restrict__ out, unsigned* __restrict__ n_invalid)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    bool bad = false;
    if (x < W && y < H) {
        const float d = dl[(size_t)y * W + x];
        const bool sane = fabsf(d) < 16777216.0f;  // NaN, inf and values no disparity can take: rejected before the int cast
        const int xr = x - (sane ? (int)d : 0);
        bad = !(sane && xr >= 0 && xr < W && fabsf(d - dr[(size_t)y * W + min(max(xr, 0), W - 1)]) <= max_diff);
        out[(size_t)y * W + x] = bad ? invalid : d;
    }
    const unsigned long long m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_invalid, (unsigned)__popcll(m));
}

int launch_lr_check(hipStream_t s, const float* dl, const float* dr, int H, int W, float max_diff, float invalid, float* out,
                    unsigned* n_invalid)
{
    ASW_HIP_TRY(hipMemsetAsync(n_invalid, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_lr_check, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, dl, dr, H, W, max_diff, invalid, out, n_invalid);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_wta(hipStream_t s, const float* vol, int n, int H, int W, int minD, float* disp)
{
    size_t plane = (size_t)H * W;
    if (plane < ((size_t)1 << 20)) {
        hipLaunchKernelGGL(k_wta<1>, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, s, vol, n, plane, minD, disp);
    } else {
        int blocks = (int)((plane / 4 + 1 + 255) / 256);
        hipLaunchKernelGGL(k_wta<4>, dim3(blocks), dim3(256), 0, s, vol, n, plane, minD, disp);
    }
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}
