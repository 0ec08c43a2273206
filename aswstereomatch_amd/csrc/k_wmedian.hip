// Weighted-median aggregation (computeColorWeightGau / computeSpaceWeightGau /
// computeAdaptiveWeight_WeightedMedian, M.cpp:3139-3383, DISPARITY_LEFT branch).
//
// Per (pixel, d) the reference inserts the win^2 (cost, weight) pairs of the support window into a
// std::multimap<float,float> (= stable sort by cost, insertion order = row-major window order), walks it
// accumulating weights in f64 until the partial sum exceeds half of the total weight, and returns the
// cost of the element BEFORE the crossing one (or the first one), M.cpp:3276-3304 (App. B-13).
//
// GPU mapping: one wavefront per pixel, looping over d two candidates at a time.  The 225 pairs of a 15x15 window sit
// 8 per lane in one half of the wavefront (256 slots, padded with max keys / zero weights); larger windows (up to 45x45) take the general path k_wmedian_big
// with 8 / 16 / 32 64-bit keys per lane.  A key made of (order-preserving cost bits, window
// index) makes a plain bitonic network a STABLE sort: 15 intra-lane and 21 cross-lane compare-exchange
// steps (DPP / ds_swizzle / bpermute), no payload is moved -- the weight of a sorted element is fetched
// from LDS by its window index afterwards.  Prefix sums are a per-lane chain plus a
// wavefront scan in f64.  Window weights are read in the reference's own per-pixel layout
// [y][x][cell] (900 contiguous bytes per pixel -> coalesced): the left one premultiplied by the space
// kernel once per frame, the right one for the REFLECT-padded right image (M.cpp:3246, 3263).
// exp() values come from host-built tables (same libm as the oracle), so weights are bit-identical.
#include "asw_device.h"
#include "asw_internal.h"
#include "wm_network.h"

namespace {


// colour weights of every window cell: out[(y*Wimg + c)*n + cell]
//   pad = 0:  left image, multiplied by the space kernel wd[cell]           (M.cpp:3262, 3274)
//   pad > 0:  right image seen through its REFLECT left pad of `pad` columns; the window itself is
//             taken from that padded image padded again by h (M.cpp:3156, 3263)
__global__ __launch_bounds__(256) void k_wm_weights(const uint8_t* __restrict__ img, int H, int W, int pad, int win,
                                                    const float* __restrict__ lut2 /* [511][256] */,
                                                    const float* __restrict__ wd /* [n] or null */, float* __restrict__ out)
{
    const int n = win * win, h = win / 2, Wb = W + pad;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)H * Wb * n) return;
    const int cell = (int)(i % n);
    const size_t pc = i / n;
    const int cb = (int)(pc % Wb), y = (int)(pc / Wb);
    const int j = cell / win, ii = cell - j * win;
    const uint8_t* c0 = img + ((size_t)y * W + reflect_idx(cb - pad, W)) * 3;
    const int cc = reflect_idx(cb - h + ii, Wb);
    const uint8_t* p = img + ((size_t)reflect_idx(y - h + j, H) * W + reflect_idx(cc - pad, W)) * 3;
    const int d0 = abs((int)p[0] - (int)c0[0]), d1 = abs((int)p[1] - (int)c0[1]), d2 = abs((int)p[2] - (int)c0[2]);
    float w = lut2[(d0 + d1) * 256 + d2];
    if (wd) w = w * wd[cell];
    out[i] = w;
}

// ---- 32-bit sort keys ---------------------------------------------------------------------------
// The TAD C+G cost with the method's literals (0.4, 10, 50; M.cpp:3250) is 0.6*cc + 0.4*cg with cc in [0,255]
// and cg in [12700, 12700+8160]: every cost lies in [4096, 16384), i.e. its f32 bit pattern minus that of
// 4096.0f fits in 24 bits and is order preserving.  key = (that << 8) | window index is a 32-bit key whose
// unsigned order is exactly the multimap's (cost, insertion order) -> v_min_u32 / v_max_u32 do the
// compare-exchange, no 64-bit compares.
constexpr uint32_t COST_BASE_BITS = 0x45800000u;  // 4096.0f

using namespace wmnet;

constexpr int WM_WAVES = 4;

// Fast path, win*win <= 256: one wavefront per pixel, TWO candidates at a time -- lanes 0-31 sort the window of candidate
// d, lanes 32-63 that of d+1, 8 keys per lane.  The network never exchanges across lane bit 5, so the two halves sort
// independently, 21 of the 36 steps are intra-lane min/max pairs and every cross-lane step is a DPP / ds_swizzle move
// (no ds_bpermute).
__global__ __launch_bounds__(256) void k_wmedian(const float* __restrict__ cost /* raw [numD][H][W] */,
                                                 const float* __restrict__ wLd /* [H][W][n] */,
                                                 const float* __restrict__ wRb /* [H][Wb][n] */, int H, int W, int win,
                                                 int numD, int max_off, float* __restrict__ out /* [numD][H][W] */)
{
    constexpr int KPL = 8;
    __shared__ float sW[WM_WAVES][2][256];
    __shared__ uint32_t sK[WM_WAVES][2][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half_id = lane >> 5, l32 = lane & 31;
    const size_t pix = (size_t)blockIdx.x * WM_WAVES + wv;
    if (pix >= (size_t)H * W) return;  // whole wave exits together
    const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
    const int n = win * win, h = win / 2, Wb = W + max_off;
    const size_t plane = (size_t)H * W;
    float* mW = sW[wv][half_id];
    uint32_t* mK = sK[wv][half_id];

    LaneMasks lm;
#pragma unroll
    for (int b = 0; b < 6; b++) lm.m[b] = (lane & (1 << b)) ? 0u : 0xffffffffu;

    float wl[KPL];
    int off[KPL];  // offset of the element's cost sample inside a cost plane (REFLECT-padded window, M.cpp:665,3273)
    int ee[KPL];   // window index, clamped for the padding slots (their loads are discarded)
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int e = l32 * KPL + r;
        const bool valid = e < n;
        ee[r] = valid ? e : 0;
        const int j = ee[r] / win, i = ee[r] - j * win;
        off[r] = reflect_idx(y + j - h, H) * W + reflect_idx(x + i - h, W);
        wl[r] = valid ? wLd[pix * n + ee[r]] : 0.0f;
    }

    for (int dbase = 0; dbase < numD; dbase += 2) {
        const int d = dbase + half_id;
        const bool dvalid = d < numD;
        const int dc = dvalid ? d : numD - 1;   // the idle half of an odd tail repeats the last candidate and stores nothing
        const int cb = x - dc + numD - 1;  // weightWinsR[y][x - offset + numDisparity - 1], M.cpp:3274
        const float* wr = wRb + ((size_t)y * Wb + cb) * n;
        const float* cp = cost + (size_t)dc * plane;
        uint32_t key[KPL];
        double s_loc = 0.0;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const int e = l32 * KPL + r;
            const float w = wl[r] * wr[ee[r]];  // (wL .mul wd) .mul wR, f32 (0 for padding slots: wl = 0)
            const uint32_t k = ((__float_as_uint(cp[off[r]]) - COST_BASE_BITS) << 8) | (uint32_t)e;
            key[r] = e < n ? k : (0xffffff00u | (uint32_t)e);  // padding slots sort last, weight 0
            s_loc += (double)w;
            mW[e] = w;
        }
        // cv::sum(weight_img_win)[0] / 2  (f64 accumulation, M.cpp:3284)
        double tot = s_loc;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        const double half = tot / 2;

        bitonic_sort<256>(key, lm);

        // weights in sorted order, inclusive prefix sums in f64
        double pre[KPL];
        double run = 0.0;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            run += (double)mW[key[r] & 0xffu];   // same-wave LDS: written above by this wave, in program order
            pre[r] = run;
            mK[l32 * KPL + r] = key[r];
        }
        double incl = run;  // inclusive scan of the lane totals inside each half
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            double t = __shfl_up(incl, o);
            if (l32 >= o) incl += t;
        }
        const double excl = incl - run;
        int first = KPL;
#pragma unroll
        for (int r = KPL - 1; r >= 0; r--)
            if (excl + pre[r] > half) first = r;
        const unsigned long long ball = __ballot(first < KPL);
        const uint32_t mine = (uint32_t)(ball >> (32 * half_id));
        float res = 0.0f;
        if (mine) {
            const int fl = __ffs((int)mine) - 1;
            const int fr = __shfl(first, fl + 32 * half_id);
            const int kpos = fl * KPL + fr;
            const int take = kpos == 0 ? 0 : kpos - 1;  // predecessor of the crossing element (M.cpp:3293-3301)
            res = __uint_as_float((mK[take] >> 8) + COST_BASE_BITS);
        }
        if (l32 == 0 && dvalid) out[(size_t)d * plane + pix] = res;
    }
}

// General path for windows above 15x15 (win*win <= 64*KPL slots): one wavefront per workgroup and pixel, KPL 64-bit keys
// per lane = (order-preserving cost bits << 32 | window index), the same stable network, the same f64 prefix walk.
// Nothing per element is kept across candidates (offsets are recomputed), so the register budget is the keys alone.
template <int KPL>
__global__ __launch_bounds__(64) void k_wmedian_big(const float* __restrict__ cost, const float* __restrict__ wLd,
                                                    const float* __restrict__ wRb, int H, int W, int win, int numD, int max_off,
                                                    float* __restrict__ out)
{
    constexpr int SLOTS = 64 * KPL;
    extern __shared__ __align__(16) unsigned char wm_smem[];
    unsigned long long* sK = reinterpret_cast<unsigned long long*>(wm_smem);   // [SLOTS] sorted keys
    float* sW = reinterpret_cast<float*>(wm_smem + (size_t)SLOTS * 8);          // [SLOTS] weights by window index
    const int lane = threadIdx.x;
    const size_t pix = blockIdx.x;
    const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
    const int n = win * win, h = win / 2, Wb = W + max_off;
    const size_t plane = (size_t)H * W;
    LaneMasks lm;
#pragma unroll
    for (int b = 0; b < 6; b++) lm.m[b] = (lane & (1 << b)) ? 0u : 0xffffffffu;

    for (int d = 0; d < numD; d++) {
        const int cb = x - d + numD - 1;  // weightWinsR[y][x - offset + numDisparity - 1], M.cpp:3274
        const float* wl = wLd + pix * n;
        const float* wr = wRb + ((size_t)y * Wb + cb) * n;
        const float* cp = cost + (size_t)d * plane;
        unsigned long long key[KPL];
        double s_loc = 0.0;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            const int e = lane * KPL + r;
            float w = 0.0f;
            key[r] = (0xffffffffull << 32) | (uint32_t)e;  // padding slots sort last and weigh nothing (sW[e] = 0)
            if (e < n) {
                const int j = e / win, i = e - j * win;
                const float c = cp[reflect_idx(y + j - h, H) * W + reflect_idx(x + i - h, W)];  // M.cpp:665,3273
                w = wl[e] * wr[e];  // (wL .mul wd) .mul wR, f32
                uint32_t u = __float_as_uint(c);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving for any sign
                key[r] = ((unsigned long long)u << 32) | (uint32_t)e;
            }
            s_loc += (double)w;
            sW[e] = w;
        }
        double tot = s_loc;  // cv::sum(weight_img_win)[0] / 2  (f64 accumulation, M.cpp:3284)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        const double half = tot / 2;

        bitonic_sort<SLOTS>(key, lm);

        double run = 0.0;  // lane total of the sorted weights (same-wave LDS: sW was written above in program order)
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            run += (double)sW[(uint32_t)key[r]];
            sK[lane * KPL + r] = key[r];
        }
        double incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            double t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        double acc = incl - run;  // exclusive prefix of this lane
        int first = KPL;
#pragma unroll
        for (int r = 0; r < KPL; r++) {
            acc += (double)sW[(uint32_t)key[r]];
            if (first == KPL && acc > half) first = r;
        }
        const unsigned long long ball = __ballot(first < KPL);
        float res = 0.0f;
        if (ball) {
            const int fl = __ffsll((long long)ball) - 1;
            const int fr = __shfl(first, fl);
            const int kpos = fl * KPL + fr;
            const int take = kpos == 0 ? 0 : kpos - 1;  // predecessor of the crossing element (M.cpp:3293-3301)
            uint32_t u = (uint32_t)(sK[take] >> 32);
            u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
            res = __uint_as_float(u);
        }
        if (lane == 0) out[(size_t)d * plane + pix] = res;
    }
}

}  // namespace

int launch_wm_weights(hipStream_t s, const uint8_t* img, int H, int W, int pad, int win, const float* lut2, const float* wd,
                      float* out)
{
    size_t n = (size_t)H * (W + pad) * win * win;
    hipLaunchKernelGGL(k_wm_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, H, W, pad, win, lut2, wd, out);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_wmedian(hipStream_t s, const float* cost, const float* wLd, const float* wRb, int H, int W, int win, int numD,
                   int max_off, float* out)
{
    size_t npix = (size_t)H * W;
    const int n = win * win;
    if (n > 256) {  // general path: 512 / 1024 / 2048 slots, 64-bit keys
        if (n > 2048) return ASW_ERR_BAD_ARGUMENT;
        const int kpl = n <= 512 ? 8 : (n <= 1024 ? 16 : 32);
        const size_t lds = (size_t)64 * kpl * 12;
        if (kpl == 8)
            hipLaunchKernelGGL(k_wmedian_big<8>, dim3((unsigned)npix), dim3(64), lds, s, cost, wLd, wRb, H, W, win, numD, max_off, out);
        else if (kpl == 16)
            hipLaunchKernelGGL(k_wmedian_big<16>, dim3((unsigned)npix), dim3(64), lds, s, cost, wLd, wRb, H, W, win, numD, max_off, out);
        else
            hipLaunchKernelGGL(k_wmedian_big<32>, dim3((unsigned)npix), dim3(64), lds, s, cost, wLd, wRb, H, W, win, numD, max_off, out);
        ASW_HIP_TRY(hipGetLastError());
        return ASW_OK;
    }
    hipLaunchKernelGGL(k_wmedian, dim3((unsigned)((npix + WM_WAVES - 1) / WM_WAVES)), dim3(256), 0, s, cost, wLd, wRb, H, W, win,
                       numD, max_off, out);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}
