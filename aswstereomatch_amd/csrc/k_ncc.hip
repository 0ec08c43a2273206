// NCC matching cost: getInputImgNCC + both computeNCC overloads (M.cpp:767-1013), SURVEY 8f row f4.
//
//   window(y,x)[r][c] = f32( pad_REFLECT(gray)(y-h+r, x-h+c) ) - boxmean_REFLECT101(gray)(y,x)        M.cpp:782-795
//   cost(y,x,off)     = sum(l*r) / ( sum(l*l) * sum(r*r) )      products in f32 (Mat::mul), sums in f64   M.cpp:867-868
//
// l = window of the reference image at (y,x); r = window of the OTHER image, which the reference pads with max_offset
// REFLECT columns first (on its left for DISPARITY_LEFT, on its right for DISPARITY_RIGHT, M.cpp:852,882), at column
// xo = x + max_offset - off (LEFT) or x + off (RIGHT).  Means and the sum(r*r) planes are taken on that padded image.
//
// k_ncc: a workgroup owns a 64 x 4 pixel tile and runs through the candidates in chunks of 16.  The reference tile
// and, per chunk, the other image's tile (64 + 2h + 15 columns) sit in LDS as f32; a thread keeps 16 f64 sums and
// walks the window row-major (the order the CPU restatement adds in), so per tap and candidate it spends one LDS read,
// one f32 subtract, one f32 multiply, one convert and one f64 add.  No MFMA: the subtraction of a per-pixel mean
// inside the product and the f32 rounding of every product keep this from being a contraction.
#include "asw_device.h"
#include "asw_internal.h"

namespace {


// copyMakeBorder(gray, 0, 0, padL, padR, BORDER_REFLECT)  (M.cpp:852, 882)
__global__ __launch_bounds__(256) void k_pad_gray(const uint8_t* __restrict__ g, int H, int W, int padL, int Wp,
                                                  uint8_t* __restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (c >= Wp) return;
    out[(size_t)y * Wp + c] = g[(size_t)y * W + reflect_idx(c - padL, W)];
}

// sum over the window of f32(l*l), l = f32(pixel) - mean(y,x), added row-major in f64   (M.cpp:868: sum(win.mul(win)))
__global__ __launch_bounds__(256) void k_ncc_selfsum(const uint8_t* __restrict__ g, const float* __restrict__ mean, int H, int W,
                                                     int win, double* __restrict__ ss)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int h = win / 2;
    const float m = mean[(size_t)y * W + x];
    double s = 0.0;
    for (int r = 0; r < win; r++) {
        const uint8_t* row = g + (size_t)reflect_idx(y - h + r, H) * W;
        for (int c = 0; c < win; c++) {
            const float v = (float)row[reflect_idx(x - h + c, W)] - m;
            const float p = v * v;
            s = s + (double)p;
        }
    }
    ss[(size_t)y * W + x] = s;
}

constexpr int NTW = 64, NTH = 4, NDC = 16;

struct NccParams {
    int H, W, Wp;          // reference image H x W; other (padded) image H x Wp
    int win, minD, numD;   // candidates: offsets minD .. minD + numD - 1
    int right;             // 0: xo = x + max_off - off (LEFT); 1: xo = x + off (RIGHT)
    int nwta;              // disparity overload: WTA over the first nwta candidates only (M.cpp:864: off < max_offset)
};

// vol (optional): f32 [numD][H][W], the planes of M.cpp:968-979 before normalize();
// disp (optional): the disparity overload's arg-min over the first nwta candidates (strict <, start DBL_MAX).
template <bool RIGHT>
__global__ __launch_bounds__(256) void k_ncc(NccParams p, const uint8_t* __restrict__ gref, const float* __restrict__ mref,
                                             const double* __restrict__ sref, const uint8_t* __restrict__ goth,
                                             const float* __restrict__ moth, const double* __restrict__ soth,
                                             float* __restrict__ vol, float* __restrict__ disp)
{
    extern __shared__ __align__(16) float ncc_smem[];
    const int h = p.win / 2, TR = NTH + 2 * h, LW = NTW + 2 * h, RW = LW + NDC - 1;
    float* sL = ncc_smem;            // [TR][LW]  reference tile
    float* sO = ncc_smem + TR * LW;  // [TR][RW]  other tile of the current chunk
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int x0 = blockIdx.x * NTW, y0 = blockIdx.y * NTH;
    const int H = p.H, W = p.W, Wp = p.Wp;
    const int max_off = p.minD + p.numD - 1;
    for (int i = tid; i < TR * LW; i += 256) {
        const int r = i / LW, c = i - r * LW;
        sL[i] = (float)gref[(size_t)reflect_idx(y0 - h + r, H) * W + reflect_idx(x0 - h + c, W)];
    }
    const int x = x0 + tx, y = y0 + ty;
    const int xc = min(x, W - 1), yc = min(y, H - 1);
    const float mL = mref[(size_t)yc * W + xc];
    const double ssL = sref[(size_t)yc * W + xc];
    double best = 1.7976931348623157e308;
    float bestD = 0.0f;

    for (int c0 = 0; c0 < p.numD; c0 += NDC) {
        const int off0 = p.minD + c0;
        // first padded column of the tile: column of window element 0 of the leftmost pixel's smallest xo in this chunk
        const int X0 = RIGHT ? x0 + off0 - h : x0 + max_off - off0 - (NDC - 1) - h;
        __syncthreads();  // previous chunk done with sO (first pass: nothing pending)
        for (int i = tid; i < TR * RW; i += 256) {
            const int r = i / RW, c = i - r * RW;
            sO[i] = (float)goth[(size_t)reflect_idx(y0 - h + r, H) * Wp + reflect_idx(X0 + c, Wp)];
        }
        __syncthreads();
        float mO[NDC];
        double ssO[NDC], acc[NDC];
#pragma unroll
        for (int dd = 0; dd < NDC; dd++) {
            const int xo = RIGHT ? x + off0 + dd : x + max_off - off0 - dd;
            const int xi = min(max(xo, 0), Wp - 1);  // only out of range for threads / candidates that are not stored
            mO[dd] = moth[(size_t)yc * Wp + xi];
            ssO[dd] = soth[(size_t)yc * Wp + xi];
            acc[dd] = 0.0;
        }
        for (int r = 0; r < p.win; r++) {
            const float* rowL = sL + (ty + r) * LW + tx;
            const float* rowO = sO + (ty + r) * RW + tx;
            for (int c = 0; c < p.win; c++) {
                const float l = rowL[c] - mL;
#pragma unroll
                for (int dd = 0; dd < NDC; dd++) {
                    const float rv = rowO[c + (RIGHT ? dd : NDC - 1 - dd)] - mO[dd];
                    const float pr = l * rv;                 // Mat::mul on CV_32F
                    acc[dd] = acc[dd] + (double)pr;          // cv::sum accumulates in f64
                }
            }
        }
        if (x < W && y < H) {
#pragma unroll
            for (int dd = 0; dd < NDC; dd++) {
                const int k = c0 + dd;
                if (k < p.numD) {
                    const double cost = acc[dd] / (ssL * ssO[dd]);  // M.cpp:867-868
                    if (vol) vol[((size_t)k * H + y) * W + x] = (float)cost;
                    if (k < p.nwta && cost < best) { best = cost; bestD = (float)(p.minD + k); }
                }
            }
        }
    }
    if (disp && x < W && y < H) disp[(size_t)y * W + x] = bestD;
}

// normalize(plane, 0, 1, NORM_MINMAX) applied in place with the per-slice scale/shift (M.cpp:981-983)
__global__ __launch_bounds__(256) void k_apply_scales(float* __restrict__ vol, size_t plane, const float2* __restrict__ scales)
{
    const int k = blockIdx.y;
    const float2 sc = scales[k];
    float* v = vol + (size_t)k * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x)
        v[i] = v[i] * sc.x + sc.y;
}

}  // namespace

int launch_pad_gray(hipStream_t s, const uint8_t* g, int H, int W, int padL, int padR, uint8_t* out)
{
    const int Wp = W + padL + padR;
    hipLaunchKernelGGL(k_pad_gray, dim3((Wp + 255) / 256, H), dim3(256), 0, s, g, H, W, padL, Wp, out);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_ncc_selfsum(hipStream_t s, const uint8_t* g, const float* mean, int H, int W, int win, double* ss)
{
    hipLaunchKernelGGL(k_ncc_selfsum, dim3((W + 255) / 256, H), dim3(256), 0, s, g, mean, H, W, win, ss);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_ncc(hipStream_t s, const NccLaunch& a)
{
    NccParams p{a.H, a.W, a.Wp, a.win, a.minD, a.numD, a.right, a.nwta};
    const int h = a.win / 2;
    const size_t lds = (size_t)(NTH + 2 * h) * ((NTW + 2 * h) + (NTW + 2 * h + NDC - 1)) * sizeof(float);
    if (lds > 64 * 1024) return ASW_ERR_BAD_ARGUMENT;
    dim3 grid((a.W + NTW - 1) / NTW, (a.H + NTH - 1) / NTH);
    if (a.right)
        hipLaunchKernelGGL(k_ncc<true>, grid, dim3(256), lds, s, p, a.gref, a.mref, a.sref, a.goth, a.moth, a.soth, a.vol, a.disp);
    else
        hipLaunchKernelGGL(k_ncc<false>, grid, dim3(256), lds, s, p, a.gref, a.mref, a.sref, a.goth, a.moth, a.soth, a.vol, a.disp);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_apply_scales(hipStream_t s, float* vol, int n, size_t plane, const float2* scales)
{
    int bx = (int)((plane + 256 * 8 - 1) / (256 * 8));
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_apply_scales, dim3(bx, n), dim3(256), 0, s, vol, plane, scales);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}
