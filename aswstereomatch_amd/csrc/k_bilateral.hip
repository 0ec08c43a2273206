// Classic bilateral adaptive-support-weight aggregation + fused WTA
// (computeAdaptiveWeight, M.cpp:1016-1156), hand-written for gfx950.
//
// Per (pixel, d) the reference evaluates, in list order i = 0..win^2-2 (M.cpp:1088-1109):
//     ab   = wL_i(y,x) * wR_i(y, max(0,x-d))                     f32 * f32 -> f32
//     num += ab * |gL(ny,nx) - gR(ny, max(0,nx-d))|              promoted to f64
//     den += ab                                                   promoted to f64
// and E = num/den feeds a strict-< running minimum over d in ascending order.
//
// Design:
//  * the 2*(win^2-1) weight maps are never materialised: w = LUT[dist-class][|dgray|], the LUT
//    is built on the host with the reference's expression (M.cpp:1054,1065);
//  * a 256-thread workgroup owns a 64x4 pixel tile; a thread owns one pixel and DC consecutive
//    disparities whose 2*DC f64 accumulators live in VGPRs.  Loops are taps-outer / d-inner:
//    every d still sees its additions in ascending tap order, so E is bit-identical to the
//    reference order (only the order WITHIN one d matters for rounding);
//  * ab*cost is exact in f64 (24-bit x 8-bit significands), so fma(ab, cost, num) == num + ab*cost;
//  * per d-chunk the gray abs-diff cost C[ny][nx][d] is built once in LDS (DC bytes per cell, one
//    ds_read_b128 per tap at DC=16); per tap-group the right-image weights of the tile rows are
//    staged in LDS (each is shared by up to DC (pixel,d) pairs), the left-image weight belongs to
//    exactly one thread and is looked up inline;
//  * the candidate count (numD+1 = 129 at D=128) is covered by 16-wide chunks plus 8/4/2/1-wide
//    remainder chunks, so no lane computes a disparity that is thrown away;
//  * clamp borders are reproduced by building the tiles with clamped coordinates (M.cpp:1059-1060,
//    1101-1106); the right-hand weight is evaluated AT max(0,x-d) (its neighbour is clamped from
//    there), which is why the right staging clamps explicitly.
// This kernel is f64-VALU-bound, not HBM-bound ("Rooflines").
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "asw_internal.h"

namespace {

constexpr int TW = 64;     // tile width  = one wavefront
constexpr int TH = 4;      // tile height = waves per workgroup
constexpr int DCMAX = 16;  // widest d-chunk

struct BilParams {
    int H, W, h, minD, nD, ntaps;
    int flip;  // 1: the problem is mirrored in x (DISPARITY_RIGHT = DISPARITY_LEFT on mirrored, swapped images)
    int cand_per_z;  // candidates per grid.z slice (multiple of 16); small images split the d range over grid.z
    int c_begin;     // first candidate of this launch; > 0: the tail of a range whose head k_asw_bilateral_xq does
    int resume;      // 1: the running minimum is resumed from partE / partD ([H][W], grid.z == 1)
    int out_slice;   // >= 0 (grid.z == 1): winners go to slice out_slice of partE / partD (merged later) instead of disp
};

constexpr __host__ __device__ int round_up(int v, int a) { return (v + a - 1) / a * a; }

// LDS layout as a function of the half window h (compile-time when HH > 0)
struct Layout {
    int h, TR, LW, LWp, RW, RWp, offC, offWR, offWL, offL, offR, total;
    __host__ __device__ constexpr Layout(int h_, int G)
        : h(h_), TR(TH + 2 * h_), LW(TW + 2 * h_), LWp(round_up(TW + 2 * h_, 4)), RW(TW + 2 * h_ + DCMAX - 1),
          RWp(round_up(TW + 2 * h_ + DCMAX - 1, 4)), offC(0), offWR(round_up((TH + 2 * h_) * (TW + 2 * h_) * DCMAX, 16)),
          offWL(offWR + round_up(G * TH * (TW + DCMAX - 1) * 4, 16)),
          offL(offWR + round_up(G * TH * (TW + DCMAX - 1) * 4, 16) + G * TH * TW * 4),
          offR(offWR + round_up(G * TH * (TW + DCMAX - 1) * 4, 16) + G * TH * TW * 4 +
               round_up((TH + 2 * h_) * round_up(TW + 2 * h_, 4), 16)),
          total(offWR + round_up(G * TH * (TW + DCMAX - 1) * 4, 16) + G * TH * TW * 4 +
                round_up((TH + 2 * h_) * round_up(TW + 2 * h_, 4), 16) +
                round_up((TH + 2 * h_) * round_up(TW + 2 * h_ + DCMAX - 1, 4), 16))
    {
    }
};

// 32-bit byte offset from a uniform base: global_load with an SGPR base, no 64-bit address arithmetic per lane
__device__ __forceinline__ float lut_at(const float* __restrict__ lut, unsigned idx)
{
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lut) + (idx << 2));
}

template <int DC>
__device__ __forceinline__ void load_cost(const uint8_t* cell, uint32_t (&cw)[(DC + 3) / 4])
{
    if constexpr (DC == 16) {
        uint4 v = *reinterpret_cast<const uint4*>(cell);
        cw[0] = v.x; cw[1] = v.y; cw[2] = v.z; cw[3] = v.w;
    } else if constexpr (DC == 8) {
        uint2 v = *reinterpret_cast<const uint2*>(cell);
        cw[0] = v.x; cw[1] = v.y;
    } else if constexpr (DC == 4) {
        cw[0] = *reinterpret_cast<const uint32_t*>(cell);
    } else if constexpr (DC == 2) {
        cw[0] = *reinterpret_cast<const uint16_t*>(cell);
    } else {
        cw[0] = cell[0];
    }
}

// One chunk of DC consecutive disparities starting at d0 (candidate index c0) for the whole tile.
// G = 8 taps per group; win*win-1 = (win-1)(win+1) is a multiple of 8 for every odd win, so groups
// are always full.
template <int DC, int G>
__device__ __forceinline__ void process_chunk(const BilParams& p, const uint8_t* __restrict__ gR,
                                              const int4* __restrict__ taps, const float* __restrict__ lut,
                                              float* __restrict__ vol, const Layout& lay, unsigned char* smem, int c0,
                                              double& bestE, float& bestD)
{
    uint8_t* sL = smem + lay.offL;
    uint8_t* sR = smem + lay.offR;
    uint8_t* sC = smem + lay.offC;
    float* sWR = reinterpret_cast<float*>(smem + lay.offWR);
    float* sWL = reinterpret_cast<float*>(smem + lay.offWL);
    constexpr int SWR = TW + DC - 1;
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int H = p.H, W = p.W, h = lay.h, LW = lay.LW, LWp = lay.LWp, RWp = lay.RWp, TR = lay.TR;
    const int RW = TW + 2 * h + DC - 1;
    const int d0 = p.minD + c0;
    // first image column of the right tile.  Never left of -h: once max(0,x-d) clamps to column 0 the
    // tile must still hold columns 0..h (the clamped pixel's neighbours).
    const int sRx0 = max(x0 - h - (d0 + DC - 1), -h);

    __syncthreads();  // previous chunk finished reading sR / sC / sWR
    for (int i = tid; i < TR * RW; i += 256) {
        int r = i / RW, c = i - r * RW;
        int yy = min(max(y0 - h + r, 0), H - 1), xx = min(max(sRx0 + c, 0), W - 1);
        sR[r * RWp + c] = gR[(size_t)yy * W + (p.flip ? W - 1 - xx : xx)];
    }
    __syncthreads();
    // Tiles whose (shifted) windows lie inside the image need no clamping (most of them): `interior` for the right-image
    // weights (positions x - d and their neighbours), `interiorC` for the cost tile (left pixels x0-h .. x0+TW+h-1 too)
    const bool interior = x0 - h - (d0 + DC - 1) >= 0 && x0 + TW + h - d0 <= W;
    const bool interiorC = x0 - h - (d0 + DC - 1) >= 0 && x0 + TW + h <= W;
    // cost tile: C[r][c][dd] = |gL(ny,nx) - gR(ny, max(0, nx-d))|   (M.cpp:1106)
    for (int i = tid; i < TR * LW; i += 256) {
        int r = i / LW, c = i - r * LW;
        const uint32_t gl = sL[r * LWp + c];
        uint32_t pk[(DC + 3) / 4];
#pragma unroll
        for (int q = 0; q < (DC + 3) / 4; q++) pk[q] = 0;
        if (interiorC) {  // right pixel of candidate dd sits at tile column c + DC-1-dd: |a - b| with the byte SAD unit
            const uint8_t* q = sR + r * RWp + c + (DC - 1);
#pragma unroll
            for (int dd = 0; dd < DC; dd++)
                pk[dd >> 2] |= __builtin_amdgcn_sad_u8(gl, (uint32_t)q[-dd], 0u) << (8 * (dd & 3));
        } else {
            int nx = min(max(x0 - h + c, 0), W - 1);
#pragma unroll
            for (int dd = 0; dd < DC; dd++) {
                int xr = max(0, nx - (d0 + dd));
                int v = abs((int)gl - (int)sR[r * RWp + min(xr - sRx0, RW - 1)]);
                pk[dd >> 2] |= (uint32_t)v << (8 * (dd & 3));
            }
        }
        if constexpr (DC >= 4) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(sC + (size_t)i * DC);
#pragma unroll
            for (int q = 0; q < DC / 4; q++) dst[q] = pk[q];
        } else {
#pragma unroll
            for (int dd = 0; dd < DC; dd++) sC[(size_t)i * DC + dd] = (uint8_t)(pk[0] >> (8 * dd));
        }
    }

    double num[DC], den[DC];
#pragma unroll
    for (int dd = 0; dd < DC; dd++) { num[dd] = 0.0; den[dd] = 0.0; }

    const uint8_t* myL = sL + (ty + h) * LWp + (tx + h);     // this thread's own left pixel
    const int ctrL = *myL;
    const uint8_t* myC = sC + (size_t)((ty + h) * LW + (tx + h)) * DC;
    const float* myWR = sWR + ty * SWR + tx + (DC - 1);

    // right-weight staging positions of a tile row: pass A = column j = tx; pass B = the DC-1 extra columns.
    // xr = clamp(x0 - d0 - (DC-1) + j): the right pixel a weight is evaluated AT.
    // Neighbour columns are clamped in tile coordinates: image column X sits at tile column X - sRx0.
    const int colLo = max(-sRx0, 0), colHi = min(W - 1 - sRx0, RW - 1);
    const int colA = min(max(x0 - d0 - (DC - 1) + tx, 0), W - 1) - sRx0;
    const uint8_t* rowA = sR + (ty + h) * RWp;
    const int ctrA = rowA[min(colA, RW - 1)];
    // Every wavefront stages the right-image weights of ITS OWN tile row (TW + DC-1 positions: pass A = lane tx, pass B =
    // the DC-1 extra columns, lanes 0..DC-2) and parks its own left weights: nothing in the tap-group loop is shared
    // between wavefronts, so the loop needs no workgroup barrier -- LDS operations of one wavefront execute in order.
    // Pass B handles the extra columns of all G taps of a group in ONE round: lane -> (tap slot tB, extra column jj),
    // G*(DC-1) <= 64 lanes; the tap parameters of that lane's slot come from a per-lane load of the tap table.
    constexpr int NB = DC > 1 ? DC - 1 : 1;
    const int tB = min(tx / NB, G - 1);
    const int jB = TW + min(tx - tB * NB, NB - 1);
    const int colB = min(max(x0 - d0 - (DC - 1) + jB, 0), W - 1) - sRx0;
    const uint8_t* rowB = sR + (ty + h) * RWp;
    const int ctrB = rowB[min(colB, RW - 1)];
    float* dstA = sWR + ty * SWR + tx;
    float* dstB = sWR + tB * (TH * SWR) + ty * SWR + jB;

    // Software pipeline over the tap groups: the LUT gathers of group g+1 are issued (index pass + loads, nothing waits)
    // right before the taps of group g are accumulated and are written to LDS at the top of the next iteration, so the
    // memory round trip of the weight staging hides behind the 4 x 88 arithmetic instructions of a group.
    const bool doB = DC > 1 && tx < G * NB;
    float wa[G], wlv[G], wb = 0.0f;
    auto gather = [&](int g0) {
        unsigned ia[G], il[G], ib;
        const int4 tpb = taps[g0 + tB];  // this lane's tap slot of pass B (vector load, L1-resident table)
        auto index_pass = [&](auto is_interior) {
#pragma unroll
            for (int t = 0; t < G; t++) {
                const int4 tp = taps[g0 + t];  // uniform: scalar loads
                // right-image weights (M.cpp:1063,1066): |gR(neighbour of xr) - gR(xr)| + class*256 -> LUT
                const int nc = decltype(is_interior)::value ? colA + tp.y : min(max(colA + tp.y, colLo), colHi);
                ia[t] = __builtin_amdgcn_sad_u16((int)rowA[tp.z * RWp + nc], ctrA, tp.w);
                // this thread's own left-image weight (M.cpp:1062,1065), parked in LDS until its tap comes up
                il[t] = __builtin_amdgcn_sad_u16((int)myL[tp.z * LWp + tp.y], ctrL, tp.w);
            }
            const int ncb = decltype(is_interior)::value ? colB + tpb.y : min(max(colB + tpb.y, colLo), colHi);
            ib = __builtin_amdgcn_sad_u16((int)rowB[__mul24(tpb.z, RWp) + ncb], ctrB, tpb.w);
        };
        if (interior) index_pass(std::true_type());
        else index_pass(std::false_type());
#pragma unroll
        for (int t = 0; t < G; t++) {
            wa[t] = lut_at(lut, ia[t]);
            wlv[t] = lut_at(lut, il[t]);
        }
        if (doB) wb = lut_at(lut, ib);
    };
    __syncthreads();  // cost tile complete
    if (p.ntaps > 0) gather(0);  // win = 1 has no taps: nothing to stage, and the tap table may be empty
    for (int g0 = 0; g0 < p.ntaps; g0 += G) {
        // same-wavefront LDS traffic is ordered: these writes follow the previous group's reads and precede this group's
#pragma unroll
        for (int t = 0; t < G; t++) {
            dstA[t * (TH * SWR)] = wa[t];
            sWL[t * (TH * TW) + tid] = wlv[t];
        }
        if (doB) *dstB = wb;
        // The hardware orders them; the COMPILER must be told that other lanes read these words (to one thread its store and the
        // loads of its neighbours' slots never alias, so it may delay or sink the store: k_guided.hip ran into exactly that).
        // Wavefront-scope fences cost no instruction.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (g0 + G < p.ntaps) gather(g0 + G);  // in flight while this group is accumulated
        // rolled on purpose: one tap's operands (1 + 4 + DC registers) live at a time keeps the kernel at
        // <= 128 VGPRs, i.e. 4 waves per SIMD, which hides the LDS latency better than deeper unrolling did
#pragma unroll 1
        for (int t = 0; t < G; t++) {
            const int4 tp = taps[g0 + t];
            const float wl = sWL[t * (TH * TW) + tid];
            uint32_t cw[(DC + 3) / 4];
            load_cost<DC>(myC + tp.x * DC, cw);
            const float* w = myWR + t * (TH * SWR);
#pragma unroll
            for (int dd = 0; dd < DC; dd++) {
                float ab = wl * w[-dd];                             // f32 product, M.cpp:1104-1105
                double abd = (double)ab;
                double c = (double)(int)((cw[dd >> 2] >> (8 * (dd & 3))) & 0xffu);
                num[dd] = __builtin_fma(abd, c, num[dd]);           // exact product -> == num + ab*c
                den[dd] = den[dd] + abd;                            // M.cpp:1107-1108
            }
        }
        // the next group's stores stay behind this group's loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    const int x = x0 + tx, y = y0 + ty;
    if (x < W && y < H) {
#pragma unroll
        for (int dd = 0; dd < DC; dd++) {
            double E = num[dd] / den[dd];  // M.cpp:1111
            if (vol) vol[((size_t)(c0 + dd) * H + y) * W + (p.flip ? W - 1 - x : x)] = (float)E;
            if (E < bestE) {  // M.cpp:1145-1150, ascending d, strict <
                bestE = E;
                bestD = (float)(d0 + dd);
            }
        }
    }
}

template <int HH, int G, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_asw_bilateral(
    BilParams p, const uint8_t* __restrict__ gL, const uint8_t* __restrict__ gR, const int4* __restrict__ taps,
    const float* __restrict__ lut, float* __restrict__ vol, float* __restrict__ disp, double* __restrict__ partE,
    float* __restrict__ partD)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const Layout lay(HH > 0 ? HH : p.h, G);
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int H = p.H, W = p.W, h = lay.h;

    // left gray tile with replicate-clamped coordinates, staged once
    uint8_t* sL = smem + lay.offL;
    for (int i = tid; i < lay.TR * lay.LW; i += 256) {
        int r = i / lay.LW, c = i - r * lay.LW;
        int yy = min(max(y0 - h + r, 0), H - 1), xx = min(max(x0 - h + c, 0), W - 1);
        sL[r * lay.LWp + c] = gL[(size_t)yy * W + (p.flip ? W - 1 - xx : xx)];
    }

    double bestE = 1.7976931348623157e308;  // numeric_limits<double>::max(), M.cpp:1037
    float bestD = 0.0f;
    if (p.resume) {  // resume the strict-'<' scan where the head launch stopped (ascending d is preserved)
        const int xr = x0 + (tid & 63), yr = y0 + (tid >> 6);
        if (xr < W && yr < H) {
            const size_t o = (size_t)yr * W + (p.flip ? W - 1 - xr : xr);
            bestE = partE[o];
            bestD = partD[o];
        }
    }
    // this workgroup's candidate range [c0, cEnd): the whole range, or one grid.z slice of it for small images
    int c0 = p.c_begin + blockIdx.z * p.cand_per_z;
    const int cEnd = min(p.nD, c0 + p.cand_per_z);
    for (; c0 + 16 <= cEnd; c0 += 16) process_chunk<16, G>(p, gR, taps, lut, vol, lay, smem, c0, bestE, bestD);
    if (cEnd - c0 >= 8) { process_chunk<8, G>(p, gR, taps, lut, vol, lay, smem, c0, bestE, bestD); c0 += 8; }
    if (cEnd - c0 >= 4) { process_chunk<4, G>(p, gR, taps, lut, vol, lay, smem, c0, bestE, bestD); c0 += 4; }
    if (cEnd - c0 >= 2) { process_chunk<2, G>(p, gR, taps, lut, vol, lay, smem, c0, bestE, bestD); c0 += 2; }
    if (cEnd - c0 >= 1) { process_chunk<1, G>(p, gR, taps, lut, vol, lay, smem, c0, bestE, bestD); c0 += 1; }

    const int x = x0 + (tid & 63), y = y0 + (tid >> 6);
    if (x < W && y < H) {
        const size_t o = (size_t)y * W + (p.flip ? W - 1 - x : x);
        if (gridDim.z == 1 && p.out_slice < 0) {
            disp[o] = bestD;
        } else {  // per-slice winners; k_merge_slices picks the first strict minimum in ascending d
            const int z = gridDim.z == 1 ? p.out_slice : (int)blockIdx.z;
            partE[(size_t)z * H * W + o] = bestE;
            partD[(size_t)z * H * W + o] = bestD;
        }
    }
}

// winners of the grid.z slices -> WTA of the whole range (M.cpp:1145-1150: strict '<' while d ascends)
__global__ __launch_bounds__(256) void k_merge_slices(const double* __restrict__ partE, const float* __restrict__ partD, int nz,
                                                      size_t plane, float* __restrict__ disp)
{
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= plane) return;
    double best = 1.7976931348623157e308;
    float bd = 0.0f;
    for (int z = 0; z < nz; z++) {
        const double e = partE[(size_t)z * plane + o];
        if (e < best) { best = e; bd = partD[(size_t)z * plane + o]; }
    }
    disp[o] = bd;
}

template <int HH, int G, int WPE = 2>
int launch_t(hipStream_t s, const BilateralLaunch& a)
{
    BilParams p;
    p.H = a.H; p.W = a.W; p.h = a.win / 2; p.minD = a.minD; p.nD = a.nD; p.ntaps = a.ntaps; p.flip = a.flip;
    p.c_begin = a.c_begin; p.resume = a.resume; p.out_slice = a.out_slice;
    if (a.c_begin < 0 || a.c_begin >= a.nD || ((a.resume || a.out_slice >= 0) && !(a.partE && a.partD))) return ASW_ERR_BAD_ARGUMENT;
    const Layout lay(p.h, G);
    if (lay.total > 160 * 1024) return ASW_ERR_BAD_ARGUMENT;
    auto kern = k_asw_bilateral<HH, G, WPE>;
    if (lay.total > 64 * 1024)
        ASW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lay.total));
    dim3 grid((a.W + TW - 1) / TW, (a.H + TH - 1) / TH);
    if (a.ntaps % G != 0) return ASW_ERR_BAD_ARGUMENT;  // cannot happen for odd windows
    // Small images do not fill 256 CUs with tiles alone (C2: 450x375 -> 752 tiles): split the candidate range over
    // grid.z in multiples of 16 until there are ~2048 workgroups, and merge the per-slice winners afterwards.
    const int tiles = grid.x * grid.y, chunks16 = (a.nD - a.c_begin + 15) / 16;
    int nz = 1;
    if (a.c_begin == 0 && !a.resume && a.out_slice < 0 && a.partE && a.partD && tiles < 2048) nz = std::min(chunks16, std::min(a.max_slices, (2048 + tiles - 1) / tiles));
    const int chunks_per_z = (chunks16 + nz - 1) / nz;
    nz = (chunks16 + chunks_per_z - 1) / chunks_per_z;
    p.cand_per_z = chunks_per_z * 16;
    grid.z = nz;
    hipLaunchKernelGGL(kern, grid, dim3(256), lay.total, s, p, a.gL, a.gR, a.taps, a.lut, a.vol, a.disp, a.partE, a.partD);
    if (nz > 1) {
        const size_t plane = (size_t)a.H * a.W;
        hipLaunchKernelGGL(k_merge_slices, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, s, a.partE, a.partD, nz, plane, a.disp);
    }
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

}  // namespace

int bilateral_lds_row_stride(int win) { return TW + 2 * (win / 2); }

int launch_merge_slices(hipStream_t s, const double* partE, const float* partD, int nz, size_t plane, float* disp)
{
    hipLaunchKernelGGL(k_merge_slices, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, s, partE, partD, nz, plane, disp);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_bilateral(hipStream_t s, const BilateralLaunch& a)
{
    switch (a.win) {
    // <half window, taps per staging group, waves per SIMD the register allocator must allow>.
    // win 15, measured on MI355X (1080p, D=128): <7,8,2> 18.6 ms, <7,8,3> 15.7, <7,4,3> 16.3, <7,4,4> 14.3 ms.
    case 15: return launch_t<7, 4, 4>(s, a);   // the reference's call site (main.cpp:94) and configs C1/C5
    case 35: return launch_t<17, 4, 2>(s, a);  // config C2 (cost tile 60 KB: two workgroups per CU)
    default: return launch_t<0, 4, 3>(s, a);   // any other odd window: layout computed at run time
    }
}
