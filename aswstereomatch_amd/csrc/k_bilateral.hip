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
//  * per d-chunk the gray abs-diff cost C[ny][nx][d] is built once in LDS (16 B per cell, one
//    ds_read_b128 per tap), and per tap-group the left/right weights of the tile rows are staged in
//    LDS so that the inner loop is LDS reads + VALU only;
//  * clamp borders are reproduced by building the tiles with clamped coordinates (M.cpp:1059-1060,
//    1101-1106); the right-hand weight is evaluated AT max(0,x-d) (its neighbour is clamped from
//    there), which is why the right staging clamps explicitly.
// This kernel is f64-VALU-bound, not HBM-bound ("Rooflines").
#include "asw_internal.h"

namespace {

constexpr int TW = 64;  // tile width  = one wavefront
constexpr int TH = 4;   // tile height = waves per workgroup

struct BilParams {
    const uint8_t* gL;
    const uint8_t* gR;
    const int4* taps;
    const float* lut;
    float* vol;
    float* disp;
    int H, W, h, minD, nD, ntaps;
    // LDS layout (bytes from the start of dynamic LDS)
    int LW, LWp, RW, RWp, TR;
    int offL, offR, offC, offWR, offWL;
};

template <int DC, int G>
__global__ __launch_bounds__(256, 2) void k_asw_bilateral(BilParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint8_t* sL = smem + p.offL;
    uint8_t* sR = smem + p.offR;
    uint8_t* sC = smem + p.offC;
    float* sWR = reinterpret_cast<float*>(smem + p.offWR);
    float* sWL = reinterpret_cast<float*>(smem + p.offWL);
    constexpr int SWR = TW + DC - 1;

    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int H = p.H, W = p.W, h = p.h, LW = p.LW, LWp = p.LWp, RW = p.RW, RWp = p.RWp, TR = p.TR;

    // left gray tile with replicate-clamped coordinates, staged once
    for (int i = tid; i < TR * LW; i += 256) {
        int r = i / LW, c = i - r * LW;
        int yy = min(max(y0 - h + r, 0), H - 1), xx = min(max(x0 - h + c, 0), W - 1);
        sL[r * LWp + c] = p.gL[(size_t)yy * W + xx];
    }

    double bestE = 1.7976931348623157e308;  // numeric_limits<double>::max(), M.cpp:1037
    float bestD = 0.0f;
    const int x = x0 + tx, y = y0 + ty;

    for (int c0 = 0; c0 < p.nD; c0 += DC) {
        const int d0 = p.minD + c0;  // first disparity of the chunk
        // first image column of the right tile.  Never left of -h: once max(0,x-d) clamps to column 0 the
        // tile must still hold columns 0..h (the clamped pixel's neighbours).
        const int sRx0 = max(x0 - h - (d0 + DC - 1), -h);
        __syncthreads();  // previous chunk finished reading sR / sC
        for (int i = tid; i < TR * RW; i += 256) {
            int r = i / RW, c = i - r * RW;
            int yy = min(max(y0 - h + r, 0), H - 1), xx = min(max(sRx0 + c, 0), W - 1);
            sR[r * RWp + c] = p.gR[(size_t)yy * W + xx];
        }
        __syncthreads();
        // cost tile: C[r][c][dd] = |gL(ny,nx) - gR(ny, max(0, nx-d))|
        for (int i = tid; i < TR * LW; i += 256) {
            int r = i / LW, c = i - r * LW;
            int nx = min(max(x0 - h + c, 0), W - 1);
            int gl = sL[r * LWp + c];
            uint32_t pk[(DC + 3) / 4];
#pragma unroll
            for (int q = 0; q < (DC + 3) / 4; q++) pk[q] = 0;
#pragma unroll
            for (int dd = 0; dd < DC; dd++) {
                int xr = max(0, nx - (d0 + dd));
                int v = abs(gl - (int)sR[r * RWp + min(xr - sRx0, RW - 1)]);  // tile is clamp-replicated
                pk[dd >> 2] |= (uint32_t)v << (8 * (dd & 3));
            }
            uint32_t* dst = reinterpret_cast<uint32_t*>(sC + (size_t)i * DC);
            if (DC >= 4) {
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

        for (int g0 = 0; g0 < p.ntaps; g0 += G) {
            const int ng = min(G, p.ntaps - g0);
            __syncthreads();  // previous group's weights consumed (first pass: cost tile complete)
            // right-image weights for xr = clamp(x0 - d0 - (DC-1) + j), j in [0, SWR)
            for (int i = tid; i < ng * TH * SWR; i += 256) {
                int tt = i / (TH * SWR), rem = i - tt * (TH * SWR);
                int row = rem / SWR, j = rem - row * SWR;
                int4 tp = p.taps[g0 + tt];
                int dxw = (tp.z & 0xffff) - 128, dyw = (tp.z >> 16) - 128;
                int xr = min(max(x0 - d0 - (DC - 1) + j, 0), W - 1);
                int xn = min(max(xr + dxw, 0), W - 1);
                // the tile is clamp-replicated, so a column left/right of it holds the same value as its edge
                int ctr = sR[(row + h) * RWp + min(xr - sRx0, RW - 1)];
                int nb = sR[(row + h + dyw) * RWp + min(xn - sRx0, RW - 1)];
                sWR[i] = p.lut[tp.w * 256 + abs(nb - ctr)];
            }
            // left-image weights at the tile's own pixels
            for (int i = tid; i < ng * TH * TW; i += 256) {
                int tt = i / (TH * TW), rem = i - tt * (TH * TW);
                int row = rem / TW, c = rem - row * TW;
                int4 tp = p.taps[g0 + tt];
                int dxw = (tp.z & 0xffff) - 128, dyw = (tp.z >> 16) - 128;
                int ctr = sL[(row + h) * LWp + (c + h)];
                int nb = sL[(row + h + dyw) * LWp + (c + h + dxw)];
                sWL[i] = p.lut[tp.w * 256 + abs(nb - ctr)];
            }
            __syncthreads();
            for (int tt = 0; tt < ng; tt++) {
                const int4 tp = p.taps[g0 + tt];  // uniform -> scalar loads
                const float wl = sWL[(tt * TH + ty) * TW + tx];
                const uint8_t* cell = sC + (size_t)((ty + h + tp.y) * LW + (tx + h + tp.x)) * DC;
                const float* wr = sWR + (tt * TH + ty) * SWR + tx + (DC - 1);
                uint32_t cw[(DC + 3) / 4];
                if (DC == 16) {
                    uint4 v = *reinterpret_cast<const uint4*>(cell);
                    cw[0] = v.x; cw[1] = v.y; cw[2] = v.z; cw[3] = v.w;
                } else if (DC == 8) {
                    uint2 v = *reinterpret_cast<const uint2*>(cell);
                    cw[0] = v.x; cw[1] = v.y;
                } else if (DC == 4) {
                    cw[0] = *reinterpret_cast<const uint32_t*>(cell);
                } else {
                    cw[0] = 0;
#pragma unroll
                    for (int dd = 0; dd < DC; dd++) cw[0] |= (uint32_t)cell[dd] << (8 * dd);
                }
#pragma unroll
                for (int dd = 0; dd < DC; dd++) {
                    float ab = wl * wr[-dd];                        // f32 product, M.cpp:1104-1105
                    double abd = (double)ab;
                    double c = (double)(int)((cw[dd >> 2] >> (8 * (dd & 3))) & 0xffu);
                    num[dd] = __builtin_fma(abd, c, num[dd]);       // exact product -> == num + ab*c
                    den[dd] = den[dd] + abd;                        // M.cpp:1107-1108
                }
            }
        }

        if (x < W && y < H) {
#pragma unroll
            for (int dd = 0; dd < DC; dd++) {
                if (c0 + dd < p.nD) {
                    double E = num[dd] / den[dd];  // M.cpp:1111
                    if (p.vol) p.vol[((size_t)(c0 + dd) * H + y) * W + x] = (float)E;
                    if (E < bestE) {  // M.cpp:1145-1150, ascending d, strict <
                        bestE = E;
                        bestD = (float)(d0 + dd);
                    }
                }
            }
        }
    }
    if (x < W && y < H) p.disp[(size_t)y * W + x] = bestD;
}

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

template <int DC, int G>
int launch_t(hipStream_t s, const BilateralLaunch& a)
{
    BilParams p;
    p.gL = a.gL; p.gR = a.gR; p.taps = a.taps; p.lut = a.lut; p.vol = a.vol; p.disp = a.disp;
    p.H = a.H; p.W = a.W; p.h = a.win / 2; p.minD = a.minD; p.nD = a.nD; p.ntaps = a.ntaps;
    const int h = p.h;
    p.TR = TH + 2 * h;
    p.LW = TW + 2 * h;
    p.LWp = round_up(p.LW, 4);
    p.RW = TW + 2 * h + DC - 1;
    p.RWp = round_up(p.RW, 4);
    int off = 0;
    p.offC = off; off += round_up(p.TR * p.LW * DC, 16);
    p.offWR = off; off += round_up(G * TH * (TW + DC - 1) * 4, 16);
    p.offWL = off; off += G * TH * TW * 4;
    p.offL = off; off += round_up(p.TR * p.LWp, 16);
    p.offR = off; off += round_up(p.TR * p.RWp, 16);
    if (off > 160 * 1024) return ASW_ERR_BAD_ARGUMENT;
    auto kern = k_asw_bilateral<DC, G>;
    if (off > 64 * 1024) ASW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, off));
    dim3 grid((a.W + TW - 1) / TW, (a.H + TH - 1) / TH);
    hipLaunchKernelGGL(kern, grid, dim3(256), off, s, p);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

}  // namespace

int launch_bilateral(hipStream_t s, const BilateralLaunch& a)
{
    // large windows: the cost tile grows with (64+2h)*(4+2h)*DC, keep it inside LDS
    if (a.win <= 21) return launch_t<16, 8>(s, a);
    if (a.win <= 45) return launch_t<8, 8>(s, a);
    return launch_t<4, 4>(s, a);
}
