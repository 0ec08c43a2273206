// Weighted-median aggregation (computeAdaptiveWeight_WeightedMedian, M.cpp:3228-3308), tile form for every window but 15x15
// (3x3 ... 13x13 and 17x17 ... 37x37; the header's default for this method is 35, M.h:179-182).  Same idea as k_wmedian_tile.hip -- the costs of
// slice d are one plane shared by all pixels, so the neighbourhood of an 8x8 pixel block is sorted ONCE per slice and every
// pixel walks the sorted list with its own weights -- with the window size a run-time parameter:
//   region  (8 + win - 1)^2 samples of the REFLECT-indexed cost plane: 256 slots (4 per lane) up to 9x9, 512 up to 13x13, 1024
//           (16 per lane) up to 25x25, 2048 (32 per lane) up to 37x37; larger windows keep the per-pixel sort (k_wmedian_big), their region does not fit 2048 slots;
//   key     (cost bits - bits(4096.0f)) * 4096 + (row * 64 + column) as an exact f64 integer: a compare-exchange is a
//           v_min_f64 / v_max_f64 pair and the order is the multimap's (cost, row-major insertion order, M.cpp:3276-3283);
//   walk    an entry's position relative to the pixel indexes a (win x 64)-entry table that yields the byte offset of its
//           window cell, or of a zero slot when the entry lies outside this pixel's window; prefix sums in sorted order,
//           first prefix above half of the total, the last MEMBER before it (M.cpp:3284-3301) -- as in k_wmedian_tile.hip;
//   parts   a workgroup holds (wL .mul wd) of one (big windows) or two block rows in LDS -- a window row of 35x35 weights is
//           4.9 KB per pixel -- and its wavefronts take the slices d = w, w + NW, ...; the parts of a block are neighbours in
//           an XCD's launch order, so the block's list comes from HBM once.
// The per-pixel sort needs a 66-step network over 2048 64-bit keys for EVERY (pixel, d): 3.9 s for a KITTI frame at 35x35.
#include <stdlib.h>

#include <algorithm>

#include "asw_device.h"
#include "asw_internal.h"
#include "wm_network.h"

namespace {
using namespace wmnet;

constexpr int BW = 8, BH = 8;
constexpr uint32_t COST_BASE_BITS = 0x45800000u;  // 4096.0f: every TAD C+G cost lies in [4096, 16384) (k_wmedian.hip)
constexpr uint32_t POS_PAD = 4095u;               // row 63: never inside a window
constexpr double KEY_PAD = 70368744177664.0;      // 2^46: above every real key (cost bits < 2^24, * 4096)

// ---- 1. sort the region of every (block, d): one wavefront per (block, slice) --------------------------------------------
template <int KPL>
__global__ __launch_bounds__(256) void k_wmg_sort(const float* __restrict__ cost /* [numD][H][W] */, int H, int W, int win, int nbx,
                                                  int d_begin, int d_count, uint32_t* __restrict__ listC,
                                                  uint16_t* __restrict__ listP)
{
    constexpr int SLOTS = 64 * KPL;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // every XCD takes a contiguous run of (slice group, block) pairs: neighbouring regions overlap and share lines of the cost plane
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, vid = xcd * (nwg >> 3) + min(xcd, nwg & 7) + (lin >> 3);
    const int dd = (vid / (int)gridDim.x) * 4 + wv;
    if (dd >= d_count) return;  // whole wavefront
    const int hw = win / 2, RW = BW + win - 1, nreg = RW * (BH + win - 1);
    const int blk = vid % (int)gridDim.x, by = blk / nbx, bx = blk - by * nbx;
    const int x0 = bx * BW, y0 = by * BH;
    const float* cp = cost + (size_t)(d_begin + dd) * H * W;
    LaneMasks lm;
#pragma unroll
    for (int b = 0; b < 6; b++) lm.m[b] = (lane & (1 << b)) ? 0u : 0xffffffffu;
    double key[KPL];
    int ey = (lane * KPL) / RW, ex = lane * KPL - ey * RW;  // element e = lane * KPL + r, walked without a division per key
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int e = lane * KPL + r;
        if (e < nreg) {
            // the window's sample at the REFLECT-padded position (M.cpp:665, 3273)
            const float c = cp[(size_t)reflect_idx(y0 - hw + ey, H) * W + reflect_idx(x0 - hw + ex, W)];
            key[r] = (double)(__float_as_uint(c) - COST_BASE_BITS) * 4096.0 + (double)(ey * 64 + ex);
        } else {
            key[r] = KEY_PAD + (double)e;
        }
        ex++;
        if (ex == RW) { ex = 0; ey++; }
    }
    bitonic_sort<SLOTS>(key, lm);
    const size_t base = ((size_t)blk * d_count + dd) * SLOTS + (size_t)lane * KPL;
    uint32_t oc[KPL], op[KPL];
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const bool pad = key[r] >= KEY_PAD;
        const uint32_t c24 = (uint32_t)(key[r] * (1.0 / 4096.0));  // exact: division by a power of two, truncation
        const uint32_t pos = (uint32_t)(key[r] - (double)c24 * 4096.0);
        oc[r] = pad ? 0u : c24 + COST_BASE_BITS;
        op[r] = pad ? POS_PAD : pos;
    }
#pragma unroll
    for (int r = 0; r < KPL; r += 4) *reinterpret_cast<uint4*>(listC + base + r) = make_uint4(oc[r], oc[r + 1], oc[r + 2], oc[r + 3]);
    if constexpr (KPL == 4) {
        *reinterpret_cast<uint2*>(listP + base) = make_uint2(op[0] | (op[1] << 16), op[2] | (op[3] << 16));
    } else {
#pragma unroll
        for (int r = 0; r < KPL; r += 8)
            *reinterpret_cast<uint4*>(listP + base + r) =
                make_uint4(op[r] | (op[r + 1] << 16), op[r + 2] | (op[r + 3] << 16), op[r + 4] | (op[r + 5] << 16), op[r + 6] | (op[r + 7] << 16));
    }
}

// ---- 2. every pixel of a part walks the sorted region ----------------------------------------------------------------------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double v)  // lanes without a source (row edge, masked rows) read 0
{
    const long long b = __double_as_longlong(v);
    int lo, hi;
    if constexpr (ROWMASK == 0xf) {
        lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, 0xf, 0xf, true);
        hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)((unsigned long long)b >> 32), CTRL, 0xf, 0xf, true);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROWMASK, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((unsigned long long)b >> 32), CTRL, ROWMASK, 0xf, false);
    }
    return __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo));
}

__device__ __forceinline__ double wave_inclusive_scan(double v)
{
    v += dpp_f64<0x111, 0xf>(v);  // row_shr:1
    v += dpp_f64<0x112, 0xf>(v);  // row_shr:2
    v += dpp_f64<0x114, 0xf>(v);  // row_shr:4
    v += dpp_f64<0x118, 0xf>(v);  // row_shr:8   -> inclusive inside each row of 16
    v += dpp_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_f64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// grid: nblocks * nparts workgroups of NW wavefronts (one dimension; the parts of a block are consecutive in an XCD's order).
// A part = rpp rows of the 8x8 block; wavefront w takes the slices d_begin + w, w + NW, ... of the chunk.
// LDS (dynamic): float sWL[8 * rpp][wls] | float sWR[NW][wls] | u16 sT[win * 64 + 1]   (wls = win^2 + 1 rounded up to 4, slot
// win^2 = 0: "not in this window").
template <int KPL, int NW, int WPE>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_wmg_pick(const float* __restrict__ wLd /* [H][W][n] */, const float* __restrict__ wRb /* [H][Wb][n] */,
                                                      const uint32_t* __restrict__ listC, const uint16_t* __restrict__ listP, int H, int W,
                                                      int win, int nbx, int nparts, int rpp, int numD, int max_off, int d_begin,
                                                      int d_count, float* __restrict__ out /* [numD][H][W] */)
{
    constexpr int SLOTS = 64 * KPL;
    constexpr int NCI = KPL == 32 ? 22 : (KPL == 16 ? 10 : (KPL == 8 ? 3 : 2));  // cells of a window in wavefront passes: 37^2 = 1369 <= 1408, 25^2 = 625 <= 640, 13^2 = 169 <= 192, 9^2 = 81 <= 128
    extern __shared__ __align__(16) unsigned char smem[];
    const int NC = win * win, hw = win / 2, wls = (NC + 1 + 3) & ~3, NT = win * 64 + 1, npart = BW * rpp;
    float* sWL = reinterpret_cast<float*>(smem);
    float* sWR = sWL + (size_t)npart * wls;
    uint16_t* sT = reinterpret_cast<uint16_t*>(sWR + (size_t)NW * wls);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // every XCD takes a contiguous run of (block, part) pairs: the parts of a block read the same lists at about the same time
    const int nwg = gridDim.x, lin = blockIdx.x;
    const int xcd = lin & 7, vid = xcd * (nwg >> 3) + min(xcd, nwg & 7) + (lin >> 3);
    const int blk = vid / nparts, part = vid - blk * nparts;
    const int by = blk / nbx, bx = blk - by * nbx;
    const int x0 = bx * BW, y0 = by * BH;
    const int Wb = W + max_off;
    const size_t plane = (size_t)H * W;
    const int p_begin = part * npart;
    if (y0 + (p_begin >> 3) >= H) return;  // this part lies below the image (whole workgroup)

    for (int i = tid; i < npart * wls; i += 64 * NW) {
        const int q = i / wls, c = i - q * wls;
        const int p = p_begin + q;
        const int x = x0 + (p & 7), y = y0 + (p >> 3);
        sWL[i] = (c < NC && x < W && y < H) ? wLd[((size_t)y * W + x) * NC + c] : 0.0f;
    }
    for (int i = tid; i < NT; i += 64 * NW) {  // t = dy * 64 + dx + 7 (dx, dy relative to the window's first cell)
        const int dy = i >> 6, dx = (i & 63) - (BW - 1);
        sT[i] = (uint16_t)(4 * ((i < NT - 1 && dx >= 0 && dx < win) ? dy * win + dx : NC));
    }
    float* wrp = sWR + (size_t)wv * wls;
    for (int c = NC + lane; c < wls; c += 64) wrp[c] = 0.0f;
    __syncthreads();

    unsigned long long vmask = 0;
    for (int q = 0; q < npart; q++)
        if (x0 + ((p_begin + q) & 7) < W && y0 + ((p_begin + q) >> 3) < H) vmask |= 1ull << q;
    for (int dd = wv; dd < d_count; dd += NW) {
        const int d = d_begin + dd;
        const size_t lbase0 = ((size_t)blk * d_count + dd) * SLOTS;
        const size_t lbase = lbase0 + (size_t)lane * KPL;
        int epos[KPL];  // twice the position: the byte offset into the u16 table needs no shift per pixel
        if constexpr (KPL == 4) {
            const uint2 pp = *reinterpret_cast<const uint2*>(listP + lbase);
            epos[0] = 2 * (int)(pp.x & 0xffffu); epos[1] = 2 * (int)(pp.x >> 16);
            epos[2] = 2 * (int)(pp.y & 0xffffu); epos[3] = 2 * (int)(pp.y >> 16);
        } else
#pragma unroll
        for (int r = 0; r + 7 < KPL; r += 8) {
            const uint4 pp = *reinterpret_cast<const uint4*>(listP + lbase + r);
            epos[r] = 2 * (int)(pp.x & 0xffffu); epos[r + 1] = 2 * (int)(pp.x >> 16);
            epos[r + 2] = 2 * (int)(pp.y & 0xffffu); epos[r + 3] = 2 * (int)(pp.y >> 16);
            epos[r + 4] = 2 * (int)(pp.z & 0xffffu); epos[r + 5] = 2 * (int)(pp.z >> 16);
            epos[r + 6] = 2 * (int)(pp.w & 0xffffu); epos[r + 7] = 2 * (int)(pp.w >> 16);
        }
        float* od = out + (size_t)d * plane;

        // right-image weight row of pixel p at this d: weightWinsR[y][x - offset + numDisparity - 1] (M.cpp:3274)
        float nxt[NCI];
        auto fetch = [&](int p) {
            const int x = x0 + (p & 7), y = y0 + (p >> 3);
            const float* row = wRb + ((size_t)y * Wb + (x - d + numD - 1)) * NC;
#pragma unroll
            for (int k = 0; k < NCI; k++) {
                const int c = lane + 64 * k;
                nxt[k] = c < NC ? row[c] : 0.0f;
            }
        };
        unsigned long long todo = vmask;  // pixels of the part that lie in the image (wave-uniform)
        int my_slot = -1;                 // lane q: list slot of the result of the part's q-th pixel
        fetch(p_begin);                   // the first pixel of a part always exists
        while (todo) {
            const int q = __builtin_ctzll(todo);
            const int p = p_begin + q;
            todo &= todo - 1;
            // the pixel's win^2 weights (wL .mul wd) .mul wR -- f32, in the reference's order, M.cpp:3274 -- one cell per lane and pass
            const float* wl_row = sWL + (size_t)q * wls;
#pragma unroll
            for (int k = 0; k < NCI; k++)
                if (lane + 64 * k < NC) wrp[lane + 64 * k] = wl_row[lane + 64 * k] * nxt[k];
            if (todo) fetch(p_begin + __builtin_ctzll(todo));  // the next pixel's row: in flight under this pixel's arithmetic
            // LDS operations of a wavefront execute in order; the compiler must know that other lanes read these words
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const char* wq = reinterpret_cast<const char*>(wrp);
            const int base2 = 2 * ((p >> 3) * 64 + (p & 7) - (BW - 1));
            double run = 0.0, pre[KPL];
            uint32_t mb = 0;  // member flags of this lane's entries, entry r at bit KPL - 1 - r
#pragma unroll
            for (int r = 0; r < KPL; r++) {
                const uint32_t t2 = min((uint32_t)(epos[r] - base2), (uint32_t)(2 * (NT - 1)));  // rows above the window wrap to huge values
                const uint32_t c4 = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(sT) + t2);
                const float w = *reinterpret_cast<const float*>(wq + c4);  // the cell's weight; 0 outside the window (zero slot)
                run = r == 0 ? (double)w : run + (double)w;
                pre[r] = run;
                mb = (mb << 1) | (c4 != 4u * (uint32_t)NC ? 1u : 0u);
            }
            const double incl = wave_inclusive_scan(run);
            const double half = readlane_f64(incl, 63) * 0.5;  // cv::sum(weight_img_win)[0] / 2, M.cpp:3284
            const double thr = half - (incl - run);            // pre[r] > half - (exclusive prefix of the lane)
            int first = KPL;
#pragma unroll
            for (int r = KPL - 1; r >= 0; r--)
                if (pre[r] > thr) first = r;
            const unsigned long long ball = __ballot(first < KPL);
            int slot = -1;
            if (ball) {  // wave-uniform
                const int fl = __ffsll((long long)ball) - 1;
                const int fr = __builtin_amdgcn_readlane(first, fl);
                const uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)mb, fl);
                const uint32_t before = fr ? bits >> (KPL - fr) : 0u;  // members among entries 0 .. fr-1 of that lane, entry fr-1 at bit 0
                int pl = fl, pr = fr;  // the crossing element itself if nothing precedes it (M.cpp:3293-3296)
                if (before) {
                    pr = fr - 1 - __builtin_ctz(before);
                } else {
                    const unsigned long long lower = __ballot(mb != 0u) & ((1ull << fl) - 1ull);
                    if (lower) {
                        pl = 63 - __builtin_clzll(lower);
                        pr = KPL - 1 - __builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)mb, pl));  // its last member
                    }
                }
                slot = pl * KPL + pr;
            }
            if (lane == q) my_slot = slot;
            // the next pixel's weights overwrite this wavefront's row: behind this pixel's gathers
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // results of the whole part at once: the cost of slot my_slot (no crossing at all -- an all-zero total -- gives 0)
        if (lane < npart && ((vmask >> lane) & 1ull)) {
            const int p = p_begin + lane;
            const float res = my_slot >= 0 ? __uint_as_float(listC[lbase0 + my_slot]) : 0.0f;
            od[(size_t)(y0 + (p >> 3)) * W + x0 + (p & 7)] = res;
        }
    }
}

template <int KPL, int NW, int WPE>
int launch_t(hipStream_t s, const float* cost, const float* wLd, const float* wRb, int H, int W, int win, int numD, int max_off,
             int d_begin, int d_count, uint32_t* listC, uint16_t* listP, float* out, int rpp)
{
    const int nbx = (W + BW - 1) / BW, nby = (H + BH - 1) / BH;
    const unsigned nb = (unsigned)(nbx * nby);
    hipLaunchKernelGGL((k_wmg_sort<KPL>), dim3(nb, (unsigned)((d_count + 3) / 4)), dim3(256), 0, s, cost, H, W, win, nbx, d_begin,
                       d_count, listC, listP);
    const int NC = win * win, wls = (NC + 1 + 3) & ~3, nparts = BH / rpp;
    const size_t lds = ((size_t)(BW * rpp + NW) * wls) * 4 + (((size_t)win * 64 + 1) * 2 + 15) / 16 * 16;
    auto kern = k_wmg_pick<KPL, NW, WPE>;
    if (lds > 160 * 1024) return ASW_ERR_BAD_ARGUMENT;
    if (lds > 64 * 1024)
        ASW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nb * (unsigned)nparts), dim3(64 * NW), lds, s, wLd, wRb, listC, listP, H, W, win, nbx, nparts, rpp, numD,
                       max_off, d_begin, d_count, out);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

}  // namespace

// Windows the general tile form serves (odd; 15x15 has its own kernels, larger regions do not fit 2048 slots, 1x1 has nothing to sort).
bool wmedian_tile_gen_supported(int win) { return (win & 1) && win >= 3 && win <= 37 && win != 15; }

// Sorted-list slots per (block, slice) of the general tile form: 256 up to 9x9, 512 up to 13x13, 1024 up to 25x25, else 2048.
int wmedian_tile_gen_slots(int win)
{
    const int nreg = (BW + win - 1) * (BH + win - 1);
    return nreg <= 256 ? 256 : (nreg <= 512 ? 512 : (nreg <= 1024 ? 1024 : 2048));
}

// Slices [d_begin, d_begin + d_count) of the weighted median at window `win`; listC: u32[blocks * d_count * slots], listP: u16[same].
// rows_per_part (0 = default): rows of the 8x8 block a workgroup keeps in LDS: 1 | 2 | 4 | 8.
int launch_wmedian_tile_gen(hipStream_t s, const float* cost, const float* wLd, const float* wRb, int H, int W, int win, int numD,
                            int max_off, int d_begin, int d_count, uint32_t* listC, uint16_t* listP, float* out, int rows_per_part)
{
    if (!wmedian_tile_gen_supported(win) || d_count <= 0 || d_begin < 0 || d_begin + d_count > numD) return ASW_ERR_BAD_ARGUMENT;
    const int slots = wmedian_tile_gen_slots(win);
    int rpp = rows_per_part;
    if (rpp != 1 && rpp != 2 && rpp != 4 && rpp != 8) rpp = slots <= 512 ? 4 : (slots == 1024 ? 2 : 1);
    if (slots == 256) return launch_t<4, 8, 4>(s, cost, wLd, wRb, H, W, win, numD, max_off, d_begin, d_count, listC, listP, out, rpp);
    if (slots == 512) return launch_t<8, 8, 4>(s, cost, wLd, wRb, H, W, win, numD, max_off, d_begin, d_count, listC, listP, out, rpp);
    if (slots == 1024) return launch_t<16, 8, 4>(s, cost, wLd, wRb, H, W, win, numD, max_off, d_begin, d_count, listC, listP, out, rpp);
    // 2048 slots: 180 registers, two workgroups of four wavefronts per CU (2 x 63 KB of LDS at 35x35).  Two workgroups of SIX
    // wavefronts at a register target of 168 (three per SIMD, 10 registers spilled): 201 against 158 ms at 35x35, 152 / 132 at 27x27.
    return launch_t<32, 4, 2>(s, cost, wLd, wRb, H, W, win, numD, max_off, d_begin, d_count, listC, listP, out, rpp);
}
