// O(1)-bilateral ASW, computeAdaptiveWeight_BLO1 (M.cpp:2505-2725) -- SURVEY 8f row f1.
//
// For every intensity key k (0, step, 2*step, ..., 255) and disparity index i the reference box-filters
//     J_{k,i} = |S_i - k| * |F - k| * cost_i          (F: the reference view, S_i: the other view shifted by i, f32 products)
// divides it by box(M_k) with M_k = |S_last - k| * |F - k| of the LAST disparity index only (M.cpp:2582 sits outside the
// i-loop), and a pixel of intensity c then blends the planes of the two keys around c (weights as written, i.e. swapped,
// M.cpp:2659-2660) or takes the plane of c itself when c is a key.
//
// The 86 x D planes (91 GB at 1080p x D=128) are never formed.  A pixel reads only TWO of the 86 keys, so this kernel
// GATHERS instead: one thread per pixel evaluates the 15x15 box sums of its own two keys directly -- 2*D*225 products
// per pixel instead of 86*D running-sum updates -- with the window rows summed first and the row sums second, in f64,
// exactly the association of the CPU restatement.  A workgroup owns a 64 x 4 tile and walks the disparity indices in
// chunks of DC: per chunk the shifted view (DC bytes per cell) and the SAD costs (DC floats per cell) of the tile + halo
// are staged in LDS with the REFLECT_101 box border and the REFLECT shift border already applied.
#include "asw_device.h"
#include "asw_internal.h"

namespace {


constexpr int BTW = 64, BTH = 4;

struct BloParams {
    int H, W, win, numD, step;
    int sgn;  // the shifted view is read at x + sgn*i: -1 (DISPARITY_LEFT: right image at x-i), +1 (RIGHT: left image at x+i)
};

template <int DC>
__global__ __launch_bounds__(256) void k_blo1(BloParams p, const uint8_t* __restrict__ gF /* reference view */,
                                              const uint8_t* __restrict__ gS /* shifted view */,
                                              const float* __restrict__ cost /* [numD][H][W] */, float* __restrict__ vol,
                                              float* __restrict__ disp)
{
    extern __shared__ __align__(16) unsigned char blo_smem[];
    const int h = p.win / 2, TR = BTH + p.win - 1, LW = BTW + p.win - 1, cells = TR * LW;
    float* sC = reinterpret_cast<float*>(blo_smem);                       // [cells][DC]  SAD cost of the chunk's indices
    uint8_t* sS = blo_smem + (size_t)cells * DC * 4;                      // [cells][DC]  shifted view
    uint8_t* sF = sS + (size_t)cells * DC;                                // [cells]      reference view
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int x0 = blockIdx.x * BTW, y0 = blockIdx.y * BTH;
    const int H = p.H, W = p.W, win = p.win;
    const size_t plane = (size_t)H * W;
    const double scale = 1.0 / ((double)win * (double)win);

    for (int i = tid; i < cells; i += 256) {
        const int r = i / LW, c = i - r * LW;
        sF[i] = gF[(size_t)reflect101_idx(y0 - h + r, H) * W + reflect101_idx(x0 - h + c, W)];
    }
    const int x = x0 + tx, y = y0 + ty;
    const int cur = gF[(size_t)min(y, H - 1) * W + min(x, W - 1)];
    // the one or two keys this pixel reads (M.cpp:2650-2666)
    const bool exact = (cur % p.step == 0) || cur == 255;
    const int lower = cur / p.step * p.step;
    const int upper = min(lower + p.step, 255);
    const int k0 = exact ? cur : lower, k1 = exact ? cur : upper;
    const float wl = (float)(cur - lower), wu = (float)(upper - cur);
    const uint8_t* myF = sF + ty * LW + tx;

    // stage the shifted view (and, for real chunks, the costs) of disparity indices i0 .. i0+DC-1
    auto stage = [&](int i0, bool with_cost) {
        __syncthreads();  // previous consumers done
        for (int i = tid; i < cells; i += 256) {
            const int r = i / LW, c = i - r * LW;
            const int yq = reflect101_idx(y0 - h + r, H), xq = reflect101_idx(x0 - h + c, W);
#pragma unroll
            for (int dd = 0; dd < DC; dd++) {
                const int ii = min(i0 + dd, p.numD - 1);
                sS[(size_t)i * DC + dd] = gS[(size_t)yq * W + reflect_idx(xq + p.sgn * ii, W)];
                if (with_cost) sC[(size_t)i * DC + dd] = cost[(size_t)ii * plane + (size_t)yq * W + xq];
            }
        }
        __syncthreads();
    };

    // normaliser box(M_k) of the LAST disparity index, for both keys (M.cpp:2582)
    float bm0, bm1;
    {
        stage(p.numD - 1, false);
        double t0 = 0.0, t1 = 0.0;
        for (int r = 0; r < win; r++) {
            double r0 = 0.0, r1 = 0.0;
            for (int c = 0; c < win; c++) {
                const int cell = r * LW + c;
                const int f = myF[cell], s = sS[(size_t)((ty + r) * LW + tx + c) * DC];
                const float m0 = (float)abs(s - k0) * (float)abs(f - k0);  // M_k_y_r.mul(M_k_y_l)
                const float m1 = (float)abs(s - k1) * (float)abs(f - k1);
                r0 = r0 + (double)m0;
                r1 = r1 + (double)m1;
            }
            t0 = t0 + r0;
            t1 = t1 + r1;
        }
        bm0 = (float)(t0 * scale);
        bm1 = (float)(t1 * scale);
    }

    double best = 1.7976931348623157e308;
    float bd = 0.0f;
    for (int i0 = 0; i0 < p.numD; i0 += DC) {
        stage(i0, true);
        double t0[DC], t1[DC];
#pragma unroll
        for (int dd = 0; dd < DC; dd++) { t0[dd] = 0.0; t1[dd] = 0.0; }
        for (int r = 0; r < win; r++) {
            double r0[DC], r1[DC];
#pragma unroll
            for (int dd = 0; dd < DC; dd++) { r0[dd] = 0.0; r1[dd] = 0.0; }
            const uint8_t* rowF = myF + r * LW;
            const uint8_t* rowS = sS + (size_t)((ty + r) * LW + tx) * DC;
            const float* rowC = sC + (size_t)((ty + r) * LW + tx) * DC;
            for (int c = 0; c < win; c++) {
                const int f = rowF[c];
                const float a0 = (float)abs(f - k0), a1 = (float)abs(f - k1);
#pragma unroll
                for (int dd = 0; dd < DC; dd++) {
                    const int s = rowS[c * DC + dd];
                    const float cs = rowC[c * DC + dd];
                    const float m0 = (float)abs(s - k0) * a0;  // M_k_y_r.mul(M_k_y_l)
                    const float m1 = (float)abs(s - k1) * a1;
                    const float j0 = m0 * cs;                  // .mul(costs_ds[i]), M.cpp:2577
                    const float j1 = m1 * cs;
                    r0[dd] = r0[dd] + (double)j0;
                    r1[dd] = r1[dd] + (double)j1;
                }
            }
#pragma unroll
            for (int dd = 0; dd < DC; dd++) { t0[dd] = t0[dd] + r0[dd]; t1[dd] = t1[dd] + r1[dd]; }
        }
        if (x < W && y < H) {
#pragma unroll
            for (int dd = 0; dd < DC; dd++) {
                const int i = i0 + dd;
                if (i < p.numD) {
                    const float jb0 = (float)(t0[dd] * scale) / bm0;  // setsJ_k_ds_y[i] / M_ki_kr_y, M.cpp:2588
                    const float jb1 = (float)(t1[dd] * scale) / bm1;
                    const float cst = exact ? jb0 : wl * jb0 + wu * jb1;  // M.cpp:2659-2660 (weights as written)
                    if (vol) vol[(size_t)i * plane + (size_t)y * W + x] = cst;
                    const double cd = (double)cst;
                    if (cd < best) { best = cd; bd = (float)i; }
                }
            }
        }
    }
    if (x < W && y < H) disp[(size_t)y * W + x] = bd;
}

template <int DC>
int launch_t(hipStream_t s, const BloParams& p, const uint8_t* gF, const uint8_t* gS, const float* cost, float* vol, float* disp)
{
    const size_t cells = (size_t)(BTH + p.win - 1) * (BTW + p.win - 1);
    const size_t lds = cells * (DC * 5 + 1) + 16;
    auto kern = k_blo1<DC>;
    if (lds > 64 * 1024)
        ASW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((p.W + BTW - 1) / BTW, (p.H + BTH - 1) / BTH);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p, gF, gS, cost, vol, disp);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

}  // namespace

int launch_blo1(hipStream_t s, const uint8_t* gl, const uint8_t* gr, const float* cost, int step, int H, int W, int disp_type,
                int win, int numD, float* vol, float* disp)
{
    const bool left = disp_type == ASW_DISPARITY_LEFT;
    BloParams p{H, W, win, numD, step, left ? -1 : 1};
    const uint8_t* gF = left ? gl : gr;
    const uint8_t* gS = left ? gr : gl;
    const size_t cells = (size_t)(BTH + win - 1) * (BTW + win - 1);
    // chunk width at win 15, 1080p D=128: 2 -> 29.6 ms, 4 -> 26.6 ms, 8 -> 30.0 ms
    if (cells * 21 + 16 <= 48 * 1024) return launch_t<4>(s, p, gF, gS, cost, vol, disp);   // win <= 21
    if (cells * 11 + 16 <= 64 * 1024) return launch_t<2>(s, p, gF, gS, cost, vol, disp);   // win <= 41
    if (cells * 6 + 16 <= 160 * 1024) return launch_t<1>(s, p, gF, gS, cost, vol, disp);
    return ASW_ERR_BAD_ARGUMENT;
}
