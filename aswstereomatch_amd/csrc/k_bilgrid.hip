// Bilateral-grid ASW: computeAdaptiveWeight_bilateralGrid (M.cpp:2253-2430, enum 5) with the grid builder createBilGrid
// (M.cpp:1831-2185) and quadrlinear_blGrid (M.cpp:2227-2251).
//
// Per candidate offset the reference rebuilds bilGrid[x][y][zL][zR] = (sum of |gL - gR_shifted|, pixel count) in nested
// std::maps, smooths it with four IN-PLACE passes (axes zR, zL, y, x: ascending along the axis, taps -1/-2 see already
// smoothed values, the count is an int member so every assignment truncates) and reads a pixel's cost as
// quadrilinear(sum) / quadrilinear(count) over the 16 keys cvCeil(coordinate / rate) +- 1; keys outside the grid are
// zeros (std::map::operator[] inserts them).  Here the grid is two dense arrays, f64 sums and i32 counts, laid out like
// the maps nest ([x][y][zL][zR], zR fastest), and one offset is four launches:
//   k_grid_cell   one wavefront per (x,y) cell: its pixels are binned with LDS atomics (integer sums: exact in any
//                 order), the zR and zL passes run on the cell's (zL,zR) block while it is still in LDS, one coalesced
//                 write of the block;
//   k_grid_pass   y pass, then x pass: one thread per line, adjacent lanes on adjacent (zL,zR) cells (coalesced), a
//                 5-element register window walks the axis (one load + one store per element and array);
//   k_grid_slice  one thread per pixel: 2 x 16 gathered reads, the two interpolations in the reference's association,
//                 the division, optional cost plane, running strict-< minimum over the offsets (f64 in HBM scratch).
// All arithmetic is the reference's f64 expression order (-ffp-contract=off): the cost volume is bit-identical to the
// CPU restatement.  HBM-bound: the grid (184 MB at 1080p, rates 10/10) is written once and read/written twice more per offset.
#include "asw_device.h"
#include "asw_internal.h"

namespace {

struct GridDims {
    int nx, ny, nl, nr;  // LAST valid index per axis: gridSize_width / height / rangeL / rangeR (M.cpp:1868-1871)
    int H, W;
    double rate_s, rate_r;
};

__device__ __forceinline__ int cv_round_dev(double v) { return (int)rint(v); }  // cvRound: ties to even
__device__ __forceinline__ int cv_ceil_dev(double v)
{
    int i = (int)v;
    return i + ((double)i < v);
}

// One smoothing pass over one line, in place, ascending (M.cpp:1936-1993 and its three repetitions).  LOADF / LOADS
// return the not-yet-smoothed element (0 beyond the last index), STORE writes the pair.
template <class LoadF, class LoadS, class Store>
__device__ __forceinline__ void smooth_line(int n, LoadF loadf, LoadS loads, Store store)
{
    double pf2 = 0.0, pf1 = 0.0, ps2 = 0.0, ps1 = 0.0;  // smoothed values at w-2, w-1
    double cf = loadf(0), cs = loads(0);                // old values at w, w+1, w+2
    double nf1 = n >= 1 ? loadf(1) : 0.0, ns1 = n >= 1 ? loads(1) : 0.0;
    for (int w = 0; w <= n; w++) {
        const double nf2 = w + 2 <= n ? loadf(w + 2) : 0.0;
        const double ns2 = w + 2 <= n ? loads(w + 2) : 0.0;
        double rf, rs;
        if (w == 0) {
            rf = 0.6 * cf + 0.3 * nf1 + 0.1 * nf2;
            rs = 0.6 * cs + 0.3 * ns1 + 0.1 * ns2;
        } else if (w == 1) {
            rf = 0.2 * pf1 + 0.5 * cf + 0.2 * nf1 + 0.1 * nf2;
            rs = 0.2 * ps1 + 0.5 * cs + 0.2 * ns1 + 0.1 * ns2;
        } else if (w == n - 1) {
            rf = 0.1 * pf2 + 0.2 * pf1 + 0.5 * cf + 0.2 * nf1;
            rs = 0.1 * ps2 + 0.2 * ps1 + 0.5 * cs + 0.2 * ns1;
        } else if (w == n) {
            rf = 0.1 * pf2 + 0.3 * pf1 + 0.6 * cf;
            rs = 0.1 * ps2 + 0.3 * ps1 + 0.6 * cs;
        } else {
            rf = 0.0625 * pf2 + 0.25 * pf1 + 0.375 * cf + 0.25 * nf1 + 0.0625 * nf2;
            rs = 0.0625 * ps2 + 0.25 * ps1 + 0.375 * cs + 0.25 * ns1 + 0.0625 * ns2;
        }
        const int si = (int)rs;  // pair<double,double> -> pair<double,int>
        store(w, rf, si);
        pf2 = pf1; pf1 = rf;
        ps2 = ps1; ps1 = (double)si;
        cf = nf1; nf1 = nf2;
        cs = ns1; ns1 = ns2;
    }
}

// Fill + zR pass + zL pass of one (x,y) cell.  64 threads.
__global__ __launch_bounds__(64) void k_grid_cell(GridDims g, const uint8_t* __restrict__ gl, const uint8_t* __restrict__ gr,
                                                  int offset, double* __restrict__ F, int* __restrict__ S)
{
    extern __shared__ __align__(16) unsigned char grid_smem[];
    const int NL = g.nl + 1, NR = g.nr + 1, ZW = NL * NR;
    double* sF = reinterpret_cast<double*>(grid_smem);            // [ZW]
    int* sS = reinterpret_cast<int*>(grid_smem + (size_t)ZW * 8);  // [ZW] counts
    unsigned* sA = reinterpret_cast<unsigned*>(sS + ZW);          // [ZW] integer sums of |gL - gR|
    const int gx = blockIdx.x, gy = blockIdx.y, t = threadIdx.x;
    for (int i = t; i < ZW; i += 64) { sS[i] = 0; sA[i] = 0u; }
    __syncthreads();
    // candidate pixels of this cell: cvRound(i / rate) == gx, tested exactly on a slightly wider interval
    const int i0 = max(0, (int)floor(((double)gx - 0.5) * g.rate_s) - 1), i1 = min(g.W - 1, (int)ceil(((double)gx + 0.5) * g.rate_s) + 1);
    const int j0 = max(0, (int)floor(((double)gy - 0.5) * g.rate_s) - 1), j1 = min(g.H - 1, (int)ceil(((double)gy + 0.5) * g.rate_s) + 1);
    const int ni = i1 - i0 + 1, nj = j1 - j0 + 1;
    for (int q = t; q < ni * nj; q += 64) {
        const int j = j0 + q / ni, i = i0 + q % ni;
        if (cv_round_dev(i / g.rate_s) != gx || cv_round_dev(j / g.rate_s) != gy) continue;
        const float vl = (float)gl[(size_t)j * g.W + i], vr = (float)gr[(size_t)j * g.W + max(0, i - offset)];  // M.cpp:1906-1907
        const int c = cv_round_dev(vl / g.rate_r) * NR + cv_round_dev(vr / g.rate_r);
        atomicAdd(&sA[c], (unsigned)fabsf(vl - vr));
        atomicAdd(&sS[c], 1);
    }
    __syncthreads();
    for (int i = t; i < ZW; i += 64) sF[i] = (double)sA[i];
    __syncthreads();
    for (int l = t; l < NL; l += 64) {  // zR pass, M.cpp:1936-1993
        double* f = sF + l * NR;
        int* s = sS + l * NR;
        smooth_line(g.nr, [&](int w) { return f[w]; }, [&](int w) { return (double)s[w]; },
                    [&](int w, double vf, int vs) { f[w] = vf; s[w] = vs; });
    }
    __syncthreads();
    for (int r = t; r < NR; r += 64) {  // zL pass, M.cpp:1996-2053
        double* f = sF + r;
        int* s = sS + r;
        smooth_line(g.nl, [&](int w) { return f[w * NR]; }, [&](int w) { return (double)s[w * NR]; },
                    [&](int w, double vf, int vs) { f[w * NR] = vf; s[w * NR] = vs; });
    }
    __syncthreads();
    const size_t base = ((size_t)gx * (g.ny + 1) + gy) * ZW;
    for (int i = t; i < ZW; i += 64) { F[base + i] = sF[i]; S[base + i] = sS[i]; }
}

// One in-place pass along y or x: line = (outer, inner) with element w at outer*outer_stride + w*stride + inner.
__global__ __launch_bounds__(256) void k_grid_pass(double* __restrict__ F, int* __restrict__ S, size_t nlines, size_t inner,
                                                   size_t outer_stride, size_t stride, int n)
{
    const size_t line = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (line >= nlines) return;
    const size_t base = (line / inner) * outer_stride + (line % inner);
    double* f = F + base;
    int* s = S + base;
    smooth_line(n, [&](int w) { return f[(size_t)w * stride]; }, [&](int w) { return (double)s[(size_t)w * stride]; },
                [&](int w, double vf, int vs) { f[(size_t)w * stride] = vf; s[(size_t)w * stride] = vs; });
}

__global__ __launch_bounds__(256) void k_grid_init(double* __restrict__ best, float* __restrict__ disp, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        best[i] = 1.7976931348623157e308;  // numeric_limits<double>::max(), M.cpp:2267
        disp[i] = 0.0f;                    // never-updated pixels (the reference leaves them uninitialised)
    }
}

// Slicing + running WTA of one offset, M.cpp:2284-2350.
__global__ __launch_bounds__(256) void k_grid_slice(GridDims g, const uint8_t* __restrict__ gl, const uint8_t* __restrict__ gr,
                                                    int offset, const double* __restrict__ F, const int* __restrict__ S,
                                                    double* __restrict__ best, float* __restrict__ disp, float* __restrict__ vol_plane)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.W || y >= g.H) return;
    const size_t o = (size_t)y * g.W + x;
    const double c[4] = {x / g.rate_s, y / g.rate_s, gl[o] / g.rate_r, gr[(size_t)y * g.W + max(0, x - offset)] / g.rate_r};
    int k[4];
    double d[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        k[a] = cv_ceil_dev(c[a]);
        d[a] = k[a] - c[a];
    }
    const int NY = g.ny + 1, NL = g.nl + 1, NR = g.nr + 1;
    double nf[16], ns[16];
#pragma unroll
    for (int n = 0; n < 16; n++) {  // x is the slowest bit, zR the fastest: the order of M.cpp:2308-2343
        const int qx = k[0] + ((n & 8) ? 1 : -1), qy = k[1] + ((n & 4) ? 1 : -1);
        const int ql = k[2] + ((n & 2) ? 1 : -1), qr = k[3] + ((n & 1) ? 1 : -1);
        const bool in = qx >= 0 && qx <= g.nx && qy >= 0 && qy <= g.ny && ql >= 0 && ql <= g.nl && qr >= 0 && qr <= g.nr;
        const size_t idx = in ? (((size_t)qx * NY + qy) * NL + ql) * NR + qr : 0;
        nf[n] = in ? F[idx] : 0.0;
        ns[n] = in ? (double)S[idx] : 0.0;
    }
    auto quad = [&](const double (&v)[16]) {  // quadrlinear_blGrid, M.cpp:2227-2251
        double a[8], b[4];
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = v[2 * i] * (1 - d[3]) + v[2 * i + 1] * d[3];
#pragma unroll
        for (int i = 0; i < 4; i++) b[i] = a[2 * i] * (1 - d[2]) + a[2 * i + 1] * d[2];
        const double c1 = b[0] * (1 - d[1]) + b[1] * d[1];
        const double c2 = b[2] * (1 - d[1]) + b[3] * d[1];
        return c1 * (1 - d[0]) + c2 * d[0];
    };
    const double cur = quad(nf) / quad(ns);
    if (vol_plane) vol_plane[o] = (float)cur;
    if (cur < best[o]) {  // M.cpp:2345-2350: ascending offsets, strict <, NaN / inf never win
        best[o] = cur;
        disp[o] = (float)offset;
    }
}

}  // namespace

int bilgrid_dims(int H, int W, double rate_s, double rate_r, int* nx, int* ny, int* nz)
{
    if (!(rate_s > 0) || !(rate_r > 0)) return ASW_ERR_BAD_ARGUMENT;  // the slicing divides by them
    const double fz = nearbyint(255.0 / rate_r), fx = nearbyint((W - 1) / rate_s), fy = nearbyint((H - 1) / rate_s);
    if (fz > 100 || fx > 1e6 || fy > 1e6) return ASW_ERR_BAD_ARGUMENT;  // the (zL,zR) block of a cell must fit in LDS
    *nz = (int)fz; *nx = (int)fx; *ny = (int)fy;
    return ASW_OK;
}

int launch_bilgrid(hipStream_t s, const uint8_t* gl, const uint8_t* gr, int H, int W, double rate_s, double rate_r, int minD,
                   int numD, double* F, int* S, double* best, float* vol, float* disp)
{
    GridDims g;
    const int rc = bilgrid_dims(H, W, rate_s, rate_r, &g.nx, &g.ny, &g.nl);
    if (rc != ASW_OK) return rc;
    g.nr = g.nl; g.H = H; g.W = W; g.rate_s = rate_s; g.rate_r = rate_r;
    const size_t plane = (size_t)H * W, ZW = (size_t)(g.nl + 1) * (g.nr + 1);
    const size_t lds = ZW * 16;
    if (lds > 64 * 1024)
        ASW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_cell), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_grid_init, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, s, best, disp, plane);
    const size_t sy = ZW, sx = ZW * (g.ny + 1);
    const size_t lines_y = (size_t)(g.nx + 1) * ZW, lines_x = sx;
    for (int k = 0; k <= numD; k++) {  // offsets minD .. minD+numD inclusive, M.cpp:2280
        const int offset = minD + k;
        hipLaunchKernelGGL(k_grid_cell, dim3(g.nx + 1, g.ny + 1), dim3(64), lds, s, g, gl, gr, offset, F, S);
        hipLaunchKernelGGL(k_grid_pass, dim3((unsigned)((lines_y + 255) / 256)), dim3(256), 0, s, F, S, lines_y, ZW, sx, sy, g.ny);  // M.cpp:2056-2118
        hipLaunchKernelGGL(k_grid_pass, dim3((unsigned)((lines_x + 255) / 256)), dim3(256), 0, s, F, S, lines_x, sx, (size_t)0, sx, g.nx);  // M.cpp:2120-2183
        hipLaunchKernelGGL(k_grid_slice, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, g, gl, gr, offset, F, S, best, disp,
                           vol ? vol + (size_t)k * plane : nullptr);
    }
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}
