// Cost-volume kernels of the guided-filter / weighted-median paths:
//   Scharr-x gradients (filter2D, M.cpp:446-450), TAD C+G similarity (computeSimilarity, M.cpp:415-487),
//   REFLECT padding of the planes (M.cpp:651-668), per-slice min/max (normalize NORM_MINMAX, M.cpp:2774-2775).
// All arithmetic follows the MatExpr evaluation order of M.cpp:455-484 operation by operation; the
// library is compiled with -ffp-contract=off so no a*b+c is fused behind our back.
#include <math.h>

#include "asw_device.h"
#include "asw_internal.h"

namespace {


// filter2D(8UC3 -> CV_32F, [-3 0 3; -10 0 10; -3 0 3]), BORDER_REFLECT_101, on the image padded on
// the left by `pad` REFLECT columns (pad = 0 for the left image, max_offset for the right one:
// the reference filters right_border, M.cpp:450).  Values are exact integers, |v| <= 4080 -> int16.
__global__ __launch_bounds__(256) void k_scharr_x(const uint8_t* __restrict__ img, int H, int W, int pad,
                                                  short* __restrict__ grad /* [H][W+pad][3] */)
{
    const int Wb = W + pad;
    const int c = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (c >= Wb) return;
    const int ym = reflect101_idx(y - 1, H), yp = reflect101_idx(y + 1, H);
    const int cm = reflect_idx(reflect101_idx(c - 1, Wb) - pad, W), cp = reflect_idx(reflect101_idx(c + 1, Wb) - pad, W);
    const uint8_t *r0 = img + (size_t)ym * W * 3, *r1 = img + (size_t)y * W * 3, *r2 = img + (size_t)yp * W * 3;
    short* o = grad + ((size_t)y * Wb + c) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        int v = 3 * ((int)r0[cp * 3 + ch] - (int)r0[cm * 3 + ch]) + 10 * ((int)r1[cp * 3 + ch] - (int)r1[cm * 3 + ch]) +
                3 * ((int)r2[cp * 3 + ch] - (int)r2[cm * 3 + ch]);
        o[ch] = (short)v;
    }
}

// One pixel of the TAD C+G cost.  c01 = c0+c1, c2: absolute colour differences; g01 = g0+g1, g2: absolute gradient
// differences (exact integers, so the float sums/differences of the reference are exact too).
// tCi = floor(thresC): for an integer colour, (double)colour > thresC  <=>  colour > floor(thresC);
// tGdn = largest float <= thresG: for a float g, (double)g > thresG  <=>  g > tGdn.
// colour term times (1 - regularity), a function of s = min(255, c0+c1) + c2 in [0, 510] only: tabulated per workgroup
__device__ __forceinline__ float similarity_colour(int s, float rr, float thresCf, int tCi)
{
    // colour term (u8): (c0+c1+c2)/3 -> round((min(255,c0+c1)+c2)/3); >thresC ? min(255, v+thresC) : 0  (M.cpp:459-465)
    int color = (s + 1) / 3;
    int maskC = (color > tCi) ? 1 : 0;
    float tf = (float)(color * maskC) * 1.0f + 255.0f * (float)maskC * thresCf;  // addWeighted(m1,1,mask,thresC/255)
    int cc_i = __float2int_rn(tf);                                                 // cvRound: to nearest even
    cc_i = min(255, max(0, cc_i));
    return (float)cc_i * rr;                            // first product of addWeighted(cc, 1-reg, cg, reg), M.cpp:484
}
__device__ __forceinline__ float similarity_gradient(int g01i, int g2i, float rg, float tGdn, float thresGf)
{
    // gradient term (f32): (g0+g1+g2)/3 = addWeighted(g0+g1, 1/3, g2, 1/3)   (M.cpp:473)
    const float third = (float)(1.0 / 3.0);
    float g01 = (float)g01i, g2 = (float)g2i;
    float g = g01 * third + g2 * third;
    int maskG = (g > tGdn) ? 1 : 0;                     // compare(>thresG)/255
    float bit = (float)maskG, bit_not = (float)(255 - maskG);  // bitwise_not of a 0/1 mask: 255/254 (App. B-6)
    float gm = g * bit;
    float cg = bit_not * thresGf + gm;                  // scaleAdd(bit_not, thresG, gm)     (M.cpp:482)
    return cg * rg;                                     // second product of the addWeighted
}
// ---- min/max reductions ------------------------------------------------------------------
// order-preserving map float -> uint so that integer atomics give float min/max (any sign)
__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float ord2f(uint32_t o)
{
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

// normalize(src, dst, 0, 1, NORM_MINMAX, CV_32F) parameters (App. A-10):
// scale = (max-min > DBL_EPSILON) ? 1/(max-min) : 0; shift = -min*scale; both cast to float.
__device__ __forceinline__ float2 minmax_scale(double smin, double smax)
{
    double scale = (smax - smin > 2.220446049250313e-16) ? 1.0 / (smax - smin) : 0.0;
    double shift = 0.0 - smin * scale;
    return make_float2((float)scale, (float)shift);
}

__device__ __forceinline__ void block_minmax_commit(uint32_t lo, uint32_t hi, uint32_t* out2)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(out2, lo);
        atomicMax(out2 + 1, hi);
    }
}

// min / max over the 16 lanes of every DPP row, result in all lanes of the row
template <bool MAX>
__device__ __forceinline__ uint32_t row_reduce_u32(uint32_t v)
{
#define ASW_STEP(ctrl)                                                                                         \
    {                                                                                                          \
        uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, 0xf, 0xf, false);             \
        v = MAX ? max(v, o) : min(v, o);                                                                       \
    }
    ASW_STEP(0xB1)   // quad_perm [1,0,3,2]
    ASW_STEP(0x4E)   // quad_perm [2,3,0,1]
    ASW_STEP(0x141)  // row_half_mirror
    ASW_STEP(0x140)  // row_mirror
#undef ASW_STEP
    return v;
}
template <bool MAX>
__device__ __forceinline__ uint32_t wave_reduce_u32(uint32_t v)
{
    v = row_reduce_u32<MAX>(v);
    uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32),
             d = __builtin_amdgcn_readlane(v, 48);
    return MAX ? max(max(a, b), max(c, d)) : min(min(a, b), min(c, d));
}

// computeSimilarity, DISPARITY_LEFT + 3 channels (the only branch that can execute, App. B-7).
// A workgroup owns 256 columns x SIM_ROWS rows and runs through a chunk of up to SIM_DCH candidates itself: the left
// pixel and gradient stay in registers, the right-image row segment every candidate shifts over (256 + chunk - 1
// columns, colours packed BGRX, gradients as 3 x i16) is staged in LDS once, so the images are read from HBM ~1.5x
// instead of once per candidate.  grid: (ceil(W/256), ceil(H/SIM_ROWS), ceil(numD/dch)), dch <= SIM_DCH; plane k <-> offset minD+k.
constexpr int SIM_ROWS = 4;
constexpr int SIM_DCH = 512;
__global__ __launch_bounds__(256) void k_similarity(const uint8_t* __restrict__ L, const uint8_t* __restrict__ R,
                                                    const short* __restrict__ gL, const short* __restrict__ gR, int H, int W,
                                                    int minD, int numD, float rr, float rg, float thresCf, int tCi, float tGdn,
                                                    float thresGf, int dch, float* __restrict__ cost,
                                                    uint32_t* __restrict__ parts /* optional [numD][waves][2] min/max keys */)
{
    extern __shared__ __align__(16) uint32_t sim_smem[];
    const int tid = threadIdx.x, x0 = blockIdx.x * 256, y0 = blockIdx.y * SIM_ROWS;
    const int kb = blockIdx.z * dch, ke = min(numD, kb + dch);
    const int max_off = minD + numD - 1, Wb = W + max_off;
    const int WL = 256 + (ke - kb) - 1, WLp = (WL + 1) & ~1;
    const int u0 = x0 - (minD + ke - 1);  // u = x - offset: leftmost shifted column this chunk touches
    uint32_t* sC = sim_smem;                                            // [SIM_ROWS][WLp] B | G<<8 | R<<16
    uint2* sG = reinterpret_cast<uint2*>(sim_smem + SIM_ROWS * WLp);     // [SIM_ROWS][WLp] {g0 | g1<<16, g2}
    // the colour term depends on s = min(255, |a0-b0| + |a1-b1|) + |a2-b2| only: 511 values, evaluated once per workgroup with the
    // very expression a pixel would use (15 of the ~35 instructions per output become one LDS read)
    float* sLut = reinterpret_cast<float*>(sim_smem + SIM_ROWS * WLp * 3);
    for (int i = tid; i < 511; i += 256) sLut[i] = similarity_colour(i, rr, thresCf, tCi);
    for (int i = tid; i < SIM_ROWS * WL; i += 256) {
        const int r = i / WL, j = i - r * WL;
        const int y = min(y0 + r, H - 1), u = u0 + j;
        const uint8_t* b = R + ((size_t)y * W + reflect_idx(u, W)) * 3;          // M.cpp:455: REFLECT-padded right image
        const short* gb = gR + ((size_t)y * Wb + min(max(u + max_off, 0), Wb - 1)) * 3;
        sC[r * WLp + j] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
        sG[r * WLp + j] = make_uint2((uint32_t)(uint16_t)gb[0] | ((uint32_t)(uint16_t)gb[1] << 16), (uint32_t)(int)gb[2]);
    }
    const int x = x0 + tid, xc = min(x, W - 1);
    const int nrows = min(SIM_ROWS, H - y0);
    uint32_t a01[SIM_ROWS], a2[SIM_ROWS];
    int ga0[SIM_ROWS], ga1[SIM_ROWS], ga2[SIM_ROWS];
#pragma unroll
    for (int r = 0; r < SIM_ROWS; r++) {
        const int y = min(y0 + r, H - 1);
        const uint8_t* a = L + ((size_t)y * W + xc) * 3;
        const short* ga = gL + ((size_t)y * W + xc) * 3;
        a01[r] = (uint32_t)a[0] | ((uint32_t)a[1] << 8);
        a2[r] = (uint32_t)a[2];
        ga0[r] = ga[0]; ga1[r] = ga[1]; ga2[r] = ga[2];
    }
    __syncthreads();
    const int nb = gridDim.x * gridDim.y * 4, bw = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (tid >> 6);
    for (int k = kb; k < ke; k++) {
        const int j = tid + (ke - 1 - k);  // (x - (minD + k)) - u0
        uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int r = 0; r < SIM_ROWS; r++) {
            if (r < nrows) {
                const uint32_t b = sC[r * WLp + j];
                const uint2 gb = sG[r * WLp + j];
                // |a0-b0| + |a1-b1| and |a2-b2| with the byte SAD unit
                const int c01 = (int)__builtin_amdgcn_sad_u8(a01[r], b & 0xffffu, 0u);
                const int c2 = (int)__builtin_amdgcn_sad_u8(a2[r], b >> 16, 0u);
                const int g01 = abs(ga0[r] - (int)(short)(gb.x & 0xffffu)) + abs(ga1[r] - ((int)gb.x >> 16));
                const int g2 = abs(ga2[r] - (int)gb.y);
                const float v = sLut[min(255, c01) + c2] + similarity_gradient(g01, g2, rg, tGdn, thresGf);
                if (x < W) {
                    cost[((size_t)k * H + (y0 + r)) * W + x] = v;
                    const uint32_t o = f2ord(v);
                    lo = min(lo, o);
                    hi = max(hi, o);
                }
            }
        }
        // normalize(NORM_MINMAX) of this slice needs its global min/max (M.cpp:2775).  Per-wavefront partials, reduced
        // by k_scales_from_parts: same-address atomics from 10^5 workgroups serialise (measured: +3.3 ms per frame).
        if (parts) {
            lo = wave_reduce_u32<false>(lo);
            hi = wave_reduce_u32<true>(hi);
            if ((tid & 63) == 0) {
                parts[((size_t)k * nb + bw) * 2] = lo;
                parts[((size_t)k * nb + bw) * 2 + 1] = hi;
            }
        }
    }
}

// parts[n][nb][2] -> scales[n]
__global__ __launch_bounds__(256) void k_scales_from_parts(const uint32_t* __restrict__ parts, int nb, float2* __restrict__ scales)
{
    const int k = blockIdx.x;
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
        lo = min(lo, parts[((size_t)k * nb + i) * 2]);
        hi = max(hi, parts[((size_t)k * nb + i) * 2 + 1]);
    }
    __shared__ uint32_t s_lo[4], s_hi[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
        hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
        scales[k] = minmax_scale((double)ord2f(lo), (double)ord2f(hi));
    }
}

// copyMakeBorder(plane, h,h,h,h, BORDER_REFLECT) for every plane (M.cpp:662-667)
__global__ __launch_bounds__(256) void k_pad_reflect(const float* __restrict__ src, int H, int W, int h,
                                                     float* __restrict__ dst)
{
    const int Hp = H + 2 * h, Wp = W + 2 * h;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= Wp) return;
    dst[((size_t)k * Hp + y) * Wp + x] = src[((size_t)k * H + reflect_idx(y - h, H)) * W + reflect_idx(x - h, W)];
}

// per-slice min/max of a dense f32 volume [n][plane] -> ord[2*n] (init: {0xffffffff, 0})
__global__ __launch_bounds__(256) void k_slice_minmax(const float* __restrict__ vol, size_t plane, uint32_t* __restrict__ ord)
{
    const int k = blockIdx.y;
    const float* p = vol + (size_t)k * plane;
    uint32_t lo = 0xffffffffu, hi = 0u;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // NaN entries (0/0 NCC costs of flat windows) are skipped, like the ordered comparisons of minMaxIdx skip them
    auto take = [&](float v) {
        const uint32_t o = f2ord(v);
        const bool nan = v != v;
        lo = min(lo, nan ? 0xffffffffu : o);
        hi = max(hi, nan ? 0u : o);
    };
    if ((plane & 3) == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        const size_t n4 = plane / 4;
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {  // four loads in flight per thread (one: 3.4 TB/s of the ~6 a read stream reaches)
            const float4 v0 = p4[i], v1 = p4[i + stride], v2 = p4[i + 2 * stride], v3 = p4[i + 3 * stride];
            take(v0.x); take(v0.y); take(v0.z); take(v0.w);
            take(v1.x); take(v1.y); take(v1.z); take(v1.w);
            take(v2.x); take(v2.y); take(v2.z); take(v2.w);
            take(v3.x); take(v3.y); take(v3.z); take(v3.w);
        }
        for (; i < n4; i += stride) {
            float4 v = p4[i];
            take(v.x); take(v.y); take(v.z); take(v.w);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += stride) take(p[i]);
    }
    block_minmax_commit(lo, hi, ord + 2 * k);
}

// min/max over all bytes of a u8 buffer -> ord[2] (as float order keys)
__global__ __launch_bounds__(256) void k_minmax_u8(const uint8_t* __restrict__ img, size_t n, uint32_t* __restrict__ ord)
{
    uint32_t lo = 255, hi = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t v = img[i];
        lo = min(lo, v);
        hi = max(hi, v);
    }
    block_minmax_commit(f2ord((float)lo), f2ord((float)hi), ord);
}

// per-column min/max over rows and channels of an interleaved 3-channel u8 image -> colmm[2*W] (u8 values)
// (init: {255, 0} per column).  A workgroup covers 64 columns x 64 rows, four row lanes per column; one thread per column
// walking all H rows took 0.44 ms at 1080p -- pure load latency.
__global__ __launch_bounds__(256) void k_col_minmax_u8(const uint8_t* __restrict__ img, int H, int W, int* __restrict__ colmm)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y0 = blockIdx.y * 64 + (threadIdx.x >> 6), y1 = min(H, blockIdx.y * 64 + 64);
    if (x >= W) return;
    int lo = 255, hi = 0;
    for (int y = y0; y < y1; y += 4) {
        const uint8_t* p = img + ((size_t)y * W + x) * 3;
        lo = min(lo, min((int)p[0], min((int)p[1], (int)p[2])));
        hi = max(hi, max((int)p[0], max((int)p[1], (int)p[2])));
    }
    if (lo <= hi) {
        atomicMin(&colmm[2 * x], lo);
        atomicMax(&colmm[2 * x + 1], hi);
    }
}

// ord[2*n] -> scales[n] (float2 {a, b})
__global__ void k_scales_from_ord(const uint32_t* __restrict__ ord, int n, float2* __restrict__ scales)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    scales[k] = minmax_scale((double)ord2f(ord[2 * k]), (double)ord2f(ord[2 * k + 1]));
}

// Guide scales of computeAdaptiveWeight_GuidedF (M.cpp:2907-2915): the 6-channel guide at disparity d is
// [L, R shifted by d through the REFLECT pad]; its min/max run over L and over the right-image columns
// the shifted view actually contains: reflect(x-d), x in [0,W)  ==  columns [0, max(d-1, W-1-d)] when
// d < W (larger d: walk the reflection explicitly).
// One wavefront per slice, lanes striding over the columns.
__global__ __launch_bounds__(64) void k_guide_scales_lr(const uint32_t* __restrict__ ordL, const int* __restrict__ colmmR, int W,
                                                        int minD, int numD, int disp_type, float2* __restrict__ scales)
{
    const int k = blockIdx.x;
    const int d = minD + k;
    int lo = 255, hi = 0;
    for (int x = threadIdx.x; x < W; x += 64) {
        int c = disp_type == ASW_DISPARITY_LEFT ? reflect_idx(x - d, W) : reflect_idx(x + d, W);
        lo = min(lo, colmmR[2 * c]);
        hi = max(hi, colmmR[2 * c + 1]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    if (threadIdx.x == 0) {
        double mn = fmin((double)ord2f(ordL[0]), (double)lo), mx = fmax((double)ord2f(ordL[1]), (double)hi);
        scales[k] = minmax_scale(mn, mx);
    }
}

__global__ void k_fill_u32(uint32_t* p, int n, uint32_t a, uint32_t b)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (i & 1) ? b : a;
}

}  // namespace

int launch_scharr_x(hipStream_t s, const uint8_t* img, int H, int W, int pad, short* grad)
{
    dim3 grid((W + pad + 255) / 256, H);
    hipLaunchKernelGGL(k_scharr_x, grid, dim3(256), 0, s, img, H, W, pad, grad);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_similarity(hipStream_t s, const uint8_t* L, const uint8_t* R, const short* gL, const short* gR, int H, int W,
                      int minD, int numD, double regularity, double thresC, double thresG, float* cost, uint32_t* ord_scratch,
                      float2* scales)
{
    // candidates per workgroup: the range is split until ~4000 workgroups exist (1080p: 2160 tiles x 2; 640x360 D=64: 270 x 16),
    // at the price of staging the right-image row segment once per split
    int dch = SIM_DCH;
    // 1080p: 2160 workgroups are 1.2 rounds of the 1792 that fit the chip (20 KB of LDS each) -- the second round ran a fifth
    // full; two candidate slices per tile: 0.485 -> 0.44 ms (4 / 8 slices: 0.436 / 0.443, 20: 0.50)
    const long long sim_wg_target = 4096;
    const long long wg_xy = (long long)((W + 255) / 256) * ((H + SIM_ROWS - 1) / SIM_ROWS);
    while (dch > 8 && wg_xy * ((numD + dch - 1) / dch) < sim_wg_target) dch /= 2;
    const int nz = (numD + dch - 1) / dch, chunk = numD < dch ? numD : dch;
    dim3 grid((W + 255) / 256, (H + SIM_ROWS - 1) / SIM_ROWS, nz);
    const int WLp = (256 + chunk - 1 + 1) & ~1;
    const size_t lds = (size_t)SIM_ROWS * WLp * 12 + 512 * 4;  // tiles + the colour-term table
    float rr = (float)(1.0 - regularity), rg = (float)regularity;  // regularityR, M.cpp:435
    float thresCf = (float)(thresC * (1.0 / 255.0));
    // integer / float forms of the two double comparisons (see similarity_colour / similarity_gradient)
    double fc = floor(thresC);
    int tCi = fc < -1.0 ? -1 : (fc > 1e6 ? 1000000 : (int)fc);
    float tGdn = (float)thresG;
    if ((double)tGdn > thresG) tGdn = nextafterf(tGdn, -INFINITY);
    hipLaunchKernelGGL(k_similarity, grid, dim3(256), lds, s, L, R, gL, gR, H, W, minD, numD, rr, rg, thresCf, tCi, tGdn,
                       (float)thresG, dch, cost, ord_scratch);
    if (ord_scratch && scales)
        hipLaunchKernelGGL(k_scales_from_parts, dim3(numD), dim3(256), 0, s, ord_scratch, (int)(grid.x * grid.y * 4), scales);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

size_t similarity_parts_words(int H, int W, int numD)
{
    return (size_t)2 * numD * ((W + 255) / 256) * ((H + SIM_ROWS - 1) / SIM_ROWS) * 4;
}

int launch_pad_reflect(hipStream_t s, const float* src, int n, int H, int W, int h, float* dst)
{
    dim3 grid((W + 2 * h + 255) / 256, H + 2 * h, n);
    hipLaunchKernelGGL(k_pad_reflect, grid, dim3(256), 0, s, src, H, W, h, dst);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_slice_scales(hipStream_t s, const float* vol, int n, size_t plane, uint32_t* ord_scratch, float2* scales)
{
    hipLaunchKernelGGL(k_fill_u32, dim3((2 * n + 255) / 256), dim3(256), 0, s, ord_scratch, 2 * n, 0xffffffffu, 0u);
    int bx = (int)((plane + 256 * 16 - 1) / (256 * 16));
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_slice_minmax, dim3(bx, n), dim3(256), 0, s, vol, plane, ord_scratch);
    hipLaunchKernelGGL(k_scales_from_ord, dim3((n + 255) / 256), dim3(256), 0, s, ord_scratch, n, scales);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_u8_scale(hipStream_t s, const uint8_t* img, size_t nbytes, uint32_t* ord_scratch, float2* scale1)
{
    hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(256), 0, s, ord_scratch, 2, 0xffffffffu, 0u);
    int bx = (int)((nbytes + 256 * 64 - 1) / (256 * 64));
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_minmax_u8, dim3(bx), dim3(256), 0, s, img, nbytes, ord_scratch);
    if (scale1) hipLaunchKernelGGL(k_scales_from_ord, dim3(1), dim3(256), 0, s, ord_scratch, 1, scale1);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_guide_scales_lr(hipStream_t s, const uint8_t* ref_img, const uint8_t* shifted_img, int H, int W, int minD, int numD,
                           int disp_type, uint32_t* ord_scratch, int* colmm_scratch, float2* scales)
{
    int rc = launch_u8_scale(s, ref_img, (size_t)H * W * 3, ord_scratch, nullptr);
    if (rc != ASW_OK) return rc;
    hipLaunchKernelGGL(k_fill_u32, dim3((2 * W + 255) / 256), dim3(256), 0, s, reinterpret_cast<uint32_t*>(colmm_scratch), 2 * W, 255u, 0u);
    hipLaunchKernelGGL(k_col_minmax_u8, dim3((W + 63) / 64, (H + 63) / 64), dim3(256), 0, s, shifted_img, H, W, colmm_scratch);
    hipLaunchKernelGGL(k_guide_scales_lr, dim3(numD), dim3(64), 0, s, ord_scratch, colmm_scratch, W, minD, numD, disp_type, scales);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}
