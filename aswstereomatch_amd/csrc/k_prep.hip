// Driver-side pre/post-processing of the reference's main() on the device:
//   resize(640x360) (aswStereoMatch.cpp:30-31), the HSV-V bilateral detail boost (:67-89) and convertTo(CV_8U) +
//   normalize(0,255,NORM_MINMAX) of the disparity map (:97-98).
// The full-resolution pair is uploaded once; the resized / enhanced pair stays resident as the frame the matchers run on,
// and the disparity can be downloaded as one byte per pixel.  All kernels are streaming (HBM-bound); every arithmetic
// step is integer or f32 arithmetic in a fixed order (-ffp-contract=off).  The OpenCV 4.1.0 semantics followed here
// (portable C++ paths of resize, cvtColor, bilateralFilter, normalize) cannot be verified offline: DESIGN.md section 2.
#include "asw_device.h"
#include "asw_internal.h"

namespace {

__device__ __forceinline__ int sat_u8(int v) { return min(max(v, 0), 255); }
__device__ __forceinline__ int floor_f(float v)
{
    int i = (int)v;
    return i - (i > v);
}

// INTER_LINEAR coefficient pair of one destination coordinate: 11-bit fixed point, cvRound'ed (resize.cpp).
// CLAMP: the x direction resets the fraction where the source pair leaves the image; the y direction does not (the two
// source rows are clipped to the image instead, resizeGeneric_).
template <bool CLAMP>
__device__ __forceinline__ void linear_coef(int d, double scale, int slen, int& s0, int& a0, int& a1)
{
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = floor_f(f);
    f -= s;
    if (CLAMP) {
        if (s < 0) { f = 0; s = 0; }
        if (s >= slen - 1) { f = 0; s = slen - 1; }
    }
    s0 = s;
    a0 = __float2int_rn((1.f - f) * 2048.f);
    a1 = __float2int_rn(f * 2048.f);
}

// resize(src, dst, Size(dw,dh)), INTER_LINEAR, 8UC3: horizontal pass in int (x2048), vertical pass with the truncating
// 8u form of VResizeLinear.  One thread per destination pixel.
__global__ __launch_bounds__(256) void k_resize_linear(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst,
                                                       int dh, int dw, double scale_x, double scale_y, int xmax)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw) return;
    int sx, a0, a1, sy, b0, b1;
    linear_coef<true>(dx, scale_x, sw, sx, a0, a1);
    linear_coef<false>(dy, scale_y, sh, sy, b0, b1);
    const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
    const uint8_t* r0 = src + ((size_t)sy0 * sw + sx) * 3;
    const uint8_t* r1 = src + ((size_t)sy1 * sw + sx) * 3;
    const bool single = dx >= xmax;  // the row loop of resizeGeneric_ switches to S[sx]*ONE from xmax on
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int h0 = single ? r0[c] * 2048 : r0[c] * a0 + r0[c + 3] * a1;
        const int h1 = single ? r1[c] * 2048 : r1[c] * a0 + r1[c + 3] * a1;
        dst[((size_t)dy * dw + dx) * 3 + c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
}

// INTER_LINEAR with an exact 2x2 downscale is executed as INTER_AREA by cv::resize: rounded 2x2 average
__global__ __launch_bounds__(256) void k_resize_area2(const uint8_t* __restrict__ src, int sw, uint8_t* __restrict__ dst, int dh, int dw)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw) return;
    const uint8_t* p = src + ((size_t)(2 * dy) * sw + 2 * dx) * 3;
    const uint8_t* q = p + (size_t)sw * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) dst[((size_t)dy * dw + dx) * 3 + c] = (uint8_t)((p[c] + p[c + 3] + q[c] + q[c + 3] + 2) >> 2);
}

// cvtColor(COLOR_BGR2HSV), 8U, H in [0,180): 12-bit division tables (color_hsv: RGB2HSV_b)
__global__ __launch_bounds__(256) void k_bgr2hsv(const uint8_t* __restrict__ bgr, size_t n, const int* __restrict__ sdiv,
                                                 const int* __restrict__ hdiv, uint8_t* __restrict__ hsv)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
    const int v = max(b, max(g, r)), vmin = min(b, min(g, r)), diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int s = (diff * sdiv[v] + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * hdiv[diff] + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
    hsv[3 * i] = (uint8_t)sat_u8(h);
    hsv[3 * i + 1] = (uint8_t)s;
    hsv[3 * i + 2] = (uint8_t)v;
}

// HSV2RGB_f on one pixel (h in [0,180), s, v in [0,1]) -> b, g, r
__device__ __forceinline__ void hsv2bgr_f(float h, float s, float v, float& b, float& g, float& r)
{
    if (s == 0) { b = g = r = v; return; }
    h *= 6.f / 180.f;
    if (h < 0) do h += 6; while (h < 0);
    else if (h >= 6) do h -= 6; while (h >= 6);
    int sector = floor_f(h);
    h -= sector;
    if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
    const float t0 = v, t1 = v * (1.f - s), t2 = v * (1.f - s * h), t3 = v * (1.f - s * (1.f - h));
    // sector_data = {1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}
    switch (sector) {
    case 0: b = t1; g = t3; r = t0; break;
    case 1: b = t1; g = t0; r = t2; break;
    case 2: b = t3; g = t0; r = t1; break;
    case 3: b = t0; g = t2; r = t1; break;
    case 4: b = t0; g = t1; r = t3; break;
    default: b = t2; g = t1; r = t0; break;
    }
}

// main.cpp:72-80: blur = bilateralFilter(V, 7, 10, 3, BORDER_REFLECT); V += 2*(V - blur) (saturating u8); HSV2BGR.
// taps: {dy, dx, weight bits}; f32 accumulation in tap order, then cvRound(sum / wsum).
__global__ __launch_bounds__(256) void k_boost_hsv2bgr(const uint8_t* __restrict__ hsv, int H, int W, const int* __restrict__ taps,
                                                       int ntaps, const float* __restrict__ color_lut, uint8_t* __restrict__ bgr)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t i = (size_t)y * W + x;
    const int val0 = hsv[3 * i + 2];
    float sum = 0.f, wsum = 0.f;
    for (int k = 0; k < ntaps; k++) {
        const int yy = reflect_idx(y + taps[3 * k], H), xx = reflect_idx(x + taps[3 * k + 1], W);
        const int val = hsv[3 * ((size_t)yy * W + xx) + 2];
        const float w = __int_as_float(taps[3 * k + 2]) * color_lut[abs(val - val0)];
        sum = sum + (float)val * w;
        wsum = wsum + w;
    }
    const int blur = __float2int_rn(sum / wsum);
    const int detail = sat_u8(val0 - blur);
    const int v2 = sat_u8(val0 + 2 * detail);
    float b, g, r;
    hsv2bgr_f((float)hsv[3 * i], hsv[3 * i + 1] * (1.f / 255.f), v2 * (1.f / 255.f), b, g, r);
    bgr[3 * i] = (uint8_t)sat_u8(__float2int_rn(b * 255.f));
    bgr[3 * i + 1] = (uint8_t)sat_u8(__float2int_rn(g * 255.f));
    bgr[3 * i + 2] = (uint8_t)sat_u8(__float2int_rn(r * 255.f));
}

// disparityMap.convertTo(CV_8UC1) (+ min / max of the result for normalize)
__global__ __launch_bounds__(256) void k_disp_to_u8(const float* __restrict__ disp, size_t n, uint8_t* __restrict__ out,
                                                    int* __restrict__ mm /* {min, max}, preset to {255, 0} */)
{
    int lo = 255, hi = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int v = sat_u8(__float2int_rn(disp[i]));
        out[i] = (uint8_t)v;
        lo = min(lo, v);
        hi = max(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(mm, lo);
        atomicMax(mm + 1, hi);
    }
}

// normalize(src, dst, 0, 255, NORM_MINMAX) on 8U: dst = saturate(cvRound(src*(float)scale + (float)shift))
__global__ __launch_bounds__(256) void k_norm_u8(uint8_t* __restrict__ img, size_t n, const int* __restrict__ mm)
{
    const int mn = mm[0], mx = mm[1];
    const double scale = 255.0 * ((double)(mx - mn) > 2.220446049250313e-16 ? 1.0 / (double)(mx - mn) : 0.0);
    const double shift = 0.0 - (double)mn * scale;
    const float fs = (float)scale, fb = (float)shift;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        img[i] = (uint8_t)sat_u8(__float2int_rn((float)img[i] * fs + fb));
}

__global__ void k_set2(int* p, int a, int b)
{
    p[0] = a;
    p[1] = b;
}

}  // namespace

int launch_resize_linear(hipStream_t s, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw)
{
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double scale_x = 1.0 / inv_x, scale_y = 1.0 / inv_y;
    const int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
    dim3 grid((dw + 255) / 256, dh);
    if (fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16 && isx == 2 && isy == 2) {
        hipLaunchKernelGGL(k_resize_area2, grid, dim3(256), 0, s, src, sw, dst, dh, dw);
    } else {
        int xmax = dw;  // first destination column whose source pair would leave the image (resize.cpp: xmax)
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = (int)floorf(fx);
            if (sx < 0) sx = 0;
            if (sx + 1 >= sw) { xmax = dx; break; }
        }
        hipLaunchKernelGGL(k_resize_linear, grid, dim3(256), 0, s, src, sh, sw, dst, dh, dw, scale_x, scale_y, xmax);
    }
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_bgr2hsv(hipStream_t s, const uint8_t* bgr, size_t n, const int* sdiv, const int* hdiv, uint8_t* hsv)
{
    hipLaunchKernelGGL(k_bgr2hsv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bgr, n, sdiv, hdiv, hsv);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_boost_hsv2bgr(hipStream_t s, const uint8_t* hsv, int H, int W, const int* taps, int ntaps, const float* color_lut,
                         uint8_t* bgr)
{
    hipLaunchKernelGGL(k_boost_hsv2bgr, dim3((W + 255) / 256, H), dim3(256), 0, s, hsv, H, W, taps, ntaps, color_lut, bgr);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_disp_to_u8(hipStream_t s, const float* disp, size_t n, int normalize, uint8_t* out, int* mm_scratch)
{
    hipLaunchKernelGGL(k_set2, dim3(1), dim3(1), 0, s, mm_scratch, 255, 0);
    int bx = (int)((n + 256 * 8 - 1) / (256 * 8));
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_disp_to_u8, dim3(bx), dim3(256), 0, s, disp, n, out, mm_scratch);
    if (normalize) hipLaunchKernelGGL(k_norm_u8, dim3(bx), dim3(256), 0, s, out, n, mm_scratch);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}
