// Geodesic ASW aggregation (computeAdaptiveWeight_geodesic, M.cpp:1467-1531) in the xq form of k_bilateral_xq.hip:
// thread = 4 pixels x 4 right-image positions, unit (a,b) runs b tap columns behind the step counter, so the units of one
// diagonal (one d, up to four pixels) share one cost sample per step.
//
//     num += fl(fl(wL(x)[j,i] * wR(max(0,x-d))[j,i]) * c),  den += fl(wL * wR)        f32 products, f64 sums
//     c = |B-B'| + |G-G'| + |R-R'| of L(ny,nx) and R(ny, max(0,nx-d)), nx = clamp(x - 7 + i), ny = clamp(y - 7 + j)
//
// Every addend is an integer below 2^40 and there are 225 of them, so the f64 sums are exact in ANY order (k_geodesic.hip):
// the tap order is free, the skew costs nothing in parity, and E = num/den is bit-identical to the reference.
//
// Differences to the bilateral form: the weights are not looked up from gray differences but streamed from the per-image
// weight planes W[cell][y][x] (u16 geodesic distances, k_geodesic_weights) -- the left column of a step for the 64 pixels,
// and per right position the tap column its units consume at that step; the cost is a v_sad_u8 of two BGRX words from
// u32 tiles (7 per step and thread instead of 16); and a launch covers 4*NJ candidates starting at any candidate base, so a
// long range is cut into passes of 128 / 64 candidates (8 / 4 wavefronts) whose winners are merged like grid.z slices.
// Against k_asw_geodesic (16 candidates per chunk) the weight planes are read once per 128 / 64 candidates instead of once
// per 16, and from LDS by 512 / 256 threads instead of per tile row.
//
// DISPARITY_RIGHT (M.cpp:1498-1520; RIGHT = true): fixed image = the right one (passed as imgL / wL), positions
// q = min(W-1, x + d) run to the right, d = d0 + 4j + b - a, dead / wrapped units are those with b < a, and the border tile is
// the first of a row (k_bilateral_xq.hip has the derivation).
#include <stdlib.h>

#include <algorithm>

#include "asw_internal.h"

namespace {

constexpr int HH = 7;
constexpr int KS = 2 * HH + 1;            // 15
constexpr int PXW = 64;                   // pixels per workgroup
constexpr int NSTEP = KS + 3;             // 18
constexpr int RING = 5;
constexpr int LWC = PXW + 2 * HH;         // 78

template <int NWAVE>
struct Geo {
    static constexpr int NJ = 4 * NWAVE;                 // position blocks per workgroup
    static constexpr int NFIN = 4 * NJ;                  // candidates finished by a launch
    static constexpr int NPOS = PXW + 4 * NJ;            // right positions [posmin, posmin + NPOS)
    static constexpr int RWC = NPOS + 2 * HH;            // right tile columns
    static constexpr int NT = 64 * NWAVE;                // threads
    static constexpr int OFF_L = 0;                                   // u32 [KS][LWC]
    static constexpr int OFF_R = OFF_L + KS * LWC * 4;                // u32 [KS][RWC]
    static constexpr int OFF_WL = (OFF_R + KS * RWC * 4 + 15) / 16 * 16;   // float [RING][KS][PXW]
    static constexpr int OFF_WR = OFF_WL + RING * KS * PXW * 4;       // float [2][KS][NPOS]
    static constexpr int TILES_END = OFF_WR + 2 * KS * NPOS * 4;
    static constexpr int OFF_PART = NFIN * PXW * 8;                   // epilogue: double E [NFIN][PXW] from 0, then partials
    static constexpr int EPI_END = OFF_PART + NWAVE * PXW * 12;
    static constexpr int LDS_TOTAL = (TILES_END > EPI_END ? TILES_END : EPI_END);
};

struct GeoXqParams {
    int H, W;
    int d0;      // disparity of candidate `cbase` = minD + cbase: the launch covers d0 .. d0 + NFIN - 1
    int cbase;   // first candidate (volume plane) of this launch
    int tile0;   // first 64-pixel tile of this launch (interior and border tiles are separate launches)
};

__device__ __forceinline__ uint32_t cdist(uint32_t a, uint32_t b) { return __builtin_amdgcn_sad_u8(a, b, 0u); }

// Stage the weights step `Kn` consumes (cf. k_bilateral_xq.hip): wave w handles window rows ky = w, w + NWAVE, ...
// plane of window cell (row ky = j, column kx = i): ky * 15 + kx (M.cpp:1481-1483: j outer, i inner; the order is free here).
// Split in two (cf. k_bilateral_xq.hip): stage_issue loads the weights of step Kn into registers at the top of step Kn - 1,
// stage_commit writes them to LDS in the middle of that step's row loop, so the plane loads fly under the accumulation.
template <int NWAVE>
struct Staged { float v[(KS + NWAVE - 1) / NWAVE][1 + Geo<NWAVE>::NPOS / 64]; };

template <int NWAVE>
__device__ __forceinline__ void stage_issue(int Kn, const uint16_t* __restrict__ wLrow, const uint16_t* __restrict__ wRrow,
                                            size_t plane, int wave, int lane, int xl, int posmin, int W, Staged<NWAVE>& st)
{
    using G = Geo<NWAVE>;
#pragma unroll
    for (int it = 0; it < (KS + NWAVE - 1) / NWAVE; it++) {
        const int ky = wave + NWAVE * it;
        if (ky >= KS) break;  // wave-uniform
        if (Kn < KS) st.v[it][0] = (float)wLrow[(size_t)(ky * KS + Kn) * plane + xl];
#pragma unroll
        for (int r3 = 0; r3 < G::NPOS / 64; r3++) {
            const int p = lane + 64 * r3;
            const int kx = Kn - (p & 3);  // the tap column the units of this position consume at step Kn
            float w = 0.0f;
            if (kx >= 0 && kx < KS) {
                const int xr = min(max(posmin + p, 0), W - 1);  // the weight window of max(0, x - d) (M.cpp:1489)
                w = (float)wRrow[(size_t)(ky * KS + kx) * plane + xr];
            }
            st.v[it][1 + r3] = w;
        }
    }
}

template <int NWAVE>
__device__ __forceinline__ void stage_commit(int Kn, unsigned char* smem, int wave, int lane, const Staged<NWAVE>& st)
{
    using G = Geo<NWAVE>;
    float* sWL = reinterpret_cast<float*>(smem + G::OFF_WL) + (Kn % RING) * (KS * PXW);
    float* sWR = reinterpret_cast<float*>(smem + G::OFF_WR) + (Kn & 1) * (KS * G::NPOS);
#pragma unroll
    for (int it = 0; it < (KS + NWAVE - 1) / NWAVE; it++) {
        const int ky = wave + NWAVE * it;
        if (ky >= KS) break;
        if (Kn < KS) sWL[ky * PXW + lane] = st.v[it][0];
#pragma unroll
        for (int r3 = 0; r3 < G::NPOS / 64; r3++) sWR[ky * G::NPOS + lane + 64 * r3] = st.v[it][1 + r3];
    }
}

template <int NWAVE>
__device__ __forceinline__ void stage_weights(int Kn, const uint16_t* __restrict__ wLrow, const uint16_t* __restrict__ wRrow,
                                              size_t plane, unsigned char* smem, int wave, int lane, int xl, int posmin, int W)
{
    Staged<NWAVE> st;
    stage_issue<NWAVE>(Kn, wLrow, wRrow, plane, wave, lane, xl, posmin, W, st);
    stage_commit<NWAVE>(Kn, smem, wave, lane, st);
}

template <int NWAVE, int K, bool EDGE, bool WRAPW, bool COMMIT, bool RIGHT>
__device__ __forceinline__ void run_step(unsigned char* smem, int g, int qrel, int qrel2, int xabs, int dbase, int dbase2,
                                         int W, int x0, int posmin, double (&num)[4][4], double (&den)[4][4], int wave, int lane,
                                         const Staged<NWAVE>& st)
{
    using G = Geo<NWAVE>;
    constexpr int BLO = K > KS - 1 ? K - (KS - 1) : 0;
    constexpr int BHI = K < 3 ? K : 3;
    constexpr int DLO = 0 - BHI, DHI = 3 - BLO;
    const uint32_t* sL = reinterpret_cast<const uint32_t*>(smem + G::OFF_L);
    const uint32_t* sR = reinterpret_cast<const uint32_t*>(smem + G::OFF_R);
    const float* sWL = reinterpret_cast<const float*>(smem + G::OFF_WL);
    const float* sWR = reinterpret_cast<const float*>(smem + G::OFF_WR) + (K & 1) * (KS * G::NPOS);

    int iL[7], iR[7];
    if constexpr (EDGE) {
#pragma unroll
        for (int dl = DLO; dl <= DHI; dl++) {
            // clamped sample column of the fixed image, the other image's column from there; its far clamp is the tile's
            const int lc = min(max(xabs + dl + K - HH, 0), W - 1);
            const int rc = RIGHT ? lc + ((dl > 0 ? dbase2 : dbase) - dl) : lc - ((dl < 0 ? dbase2 : dbase) + dl);
            iL[dl + 3] = min(max(lc - (x0 - HH), 0), LWC - 1);
            iR[dl + 3] = min(max(rc - (posmin - HH), 0), G::RWC - 1);
        }
    }
    const uint32_t* pl = sL + 4 * g + K;
    const uint32_t* pr = sR + qrel + K;
    const uint32_t* pr2 = sR + qrel2 + K;
    const float* pwl = sWL + 4 * g;
    const float* pwr = sWR + qrel;
    const float* pwr2 = sWR + qrel2;
#pragma unroll 1
    for (int ky = 0; ky < KS; ky++) {
        if constexpr (COMMIT) {
            if (ky == KS / 2) stage_commit<NWAVE>(K + 1, smem, wave, lane, st);  // the loads issued before this loop have landed
        }
        float c[7];
        if constexpr (!EDGE) {
            const uint32_t gr = pr[ky * G::RWC];
            uint32_t gr2 = gr;
            if constexpr (WRAPW && (RIGHT ? DHI > 0 : DLO < 0)) gr2 = pr2[ky * G::RWC];
#pragma unroll
            for (int dl = DLO; dl <= DHI; dl++)
                c[dl + 3] = (float)cdist(pl[ky * LWC + dl], (RIGHT ? dl > 0 : dl < 0) ? gr2 : gr);   // M.cpp:1490
        } else {
#pragma unroll
            for (int dl = DLO; dl <= DHI; dl++) c[dl + 3] = (float)cdist(sL[ky * LWC + iL[dl + 3]], sR[ky * G::RWC + iR[dl + 3]]);
        }
        const float4 wr4 = *reinterpret_cast<const float4*>(pwr + ky * G::NPOS);
        const float wr[4] = {wr4.x, wr4.y, wr4.z, wr4.w};
        float wr2[4] = {wr4.x, wr4.y, wr4.z, wr4.w};
        if constexpr ((WRAPW || EDGE) && (RIGHT ? BLO <= 2 : BHI >= 1)) {
            const float4 w2 = *reinterpret_cast<const float4*>(pwr2 + ky * G::NPOS);
            wr2[0] = w2.x; wr2[1] = w2.y; wr2[2] = w2.z; wr2[3] = w2.w;
        }
#pragma unroll
        for (int b = BLO; b <= BHI; b++) {
            const float4 wl4 = *reinterpret_cast<const float4*>(pwl + (((K - b) % RING) * KS + ky) * PXW);
            const float wl[4] = {wl4.x, wl4.y, wl4.z, wl4.w};
            float ab[4];
            if constexpr (!(WRAPW || EDGE)) {  // one multiplier per unit row: two packed products
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 lo = {wl[0], wl[1]}, hi = {wl[2], wl[3]}, w = {wr[b], wr[b]};
                const f32x2 p0 = lo * w, p1 = hi * w;
                ab[0] = p0.x; ab[1] = p0.y; ab[2] = p1.x; ab[3] = p1.y;
            } else {
#pragma unroll
                for (int a = 0; a < 4; a++) ab[a] = wl[a] * ((RIGHT ? b < a : a < b) ? wr2[b] : wr[b]);   // f32
            }
#pragma unroll
            for (int a = 0; a < 4; a++) {
                // f32 (M.cpp:1488-1490).  As one v_mul_f32: left to itself the compiler pairs these products into
                // v_pk_mul_f32 and pays for it with ~29 v_mov per step to line the (ab, c) operands up in even-aligned
                // register pairs (135 VALU instructions per step instead of ~110; every one costs an issue slot)
                float abc;
                asm("v_mul_f32 %0, %1, %2" : "=v"(abc) : "v"(ab[a]), "v"(c[a - b + 3]));
                num[a][b] = num[a][b] + (double)abc;
                den[a][b] = den[a][b] + (double)ab[a];               // M.cpp:1491-1492
            }
        }
    }
}

template <int NWAVE, bool EDGE, bool WRAPW, bool RIGHT>
__device__ __forceinline__ void run_all_steps(unsigned char* smem, const uint16_t* __restrict__ wLrow, const uint16_t* __restrict__ wRrow,
                                              size_t plane, int wave, int lane, int xl, int g, int qrel, int qrel2, int xabs, int dbase,
                                              int dbase2, int W, int x0, int posmin, double (&num)[4][4], double (&den)[4][4])
{
    Staged<NWAVE> st;
#define ASW_GXQ_STEP(KK)                                                                                             \
    if ((KK) + 1 < NSTEP) {                                                                                          \
        if constexpr (EDGE) stage_weights<NWAVE>((KK) + 1, wLrow, wRrow, plane, smem, wave, lane, xl, posmin, W);    \
        else stage_issue<NWAVE>((KK) + 1, wLrow, wRrow, plane, wave, lane, xl, posmin, W, st);                       \
    }                                                                                                                \
    run_step<NWAVE, (KK), EDGE, WRAPW, (!EDGE && (KK) + 1 < NSTEP), RIGHT>(smem, g, qrel, qrel2, xabs, dbase, dbase2, W, x0, posmin, num, \
                                                                    den, wave, lane, st);                            \
    __syncthreads();
    ASW_GXQ_STEP(0) ASW_GXQ_STEP(1) ASW_GXQ_STEP(2) ASW_GXQ_STEP(3) ASW_GXQ_STEP(4) ASW_GXQ_STEP(5)
    ASW_GXQ_STEP(6) ASW_GXQ_STEP(7) ASW_GXQ_STEP(8) ASW_GXQ_STEP(9) ASW_GXQ_STEP(10) ASW_GXQ_STEP(11)
    ASW_GXQ_STEP(12) ASW_GXQ_STEP(13) ASW_GXQ_STEP(14) ASW_GXQ_STEP(15) ASW_GXQ_STEP(16) ASW_GXQ_STEP(17)
#undef ASW_GXQ_STEP
}

// grid (tiles of this launch, H), 64 * NWAVE threads.  imgL / imgR: packed BGRX planes; wL / wR: u16 weight planes
// [225][H][W].  vol (optional): [nD][H][W]; outE / outD: one slice [H][W] of per-pass winners (strict '<' in ascending d).
template <int NWAVE, bool EDGE, bool RIGHT>
__global__ __launch_bounds__(64 * NWAVE) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_asw_geodesic_xq(
    GeoXqParams p, const uint32_t* __restrict__ imgL, const uint32_t* __restrict__ imgR, const uint16_t* __restrict__ wL,
    const uint16_t* __restrict__ wR, float* __restrict__ vol, double* __restrict__ outE, float* __restrict__ outD)
{
    using G = Geo<NWAVE>;
    __shared__ __align__(16) unsigned char smem[G::LDS_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;
    // Workgroups are dealt round-robin over the 8 XCDs in launch order.  Give every XCD a CONTIGUOUS run of (row, tile) pairs
    // instead: the 64 workgroups an XCD has in flight are then neighbouring tiles of a few rows, whose right-image positions
    // overlap (188 positions per 64 pixels) and whose weight-table lines are fetched from HBM once and hit in that XCD's L2
    // afterwards.  In launch order the tiles of an XCD were 512 pixels apart: every workgroup fetched its own 113 KB.
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, vid = xcd * (nwg >> 3) + min(xcd, nwg & 7) + (lin >> 3);
    const int x0 = (p.tile0 + vid % (int)gridDim.x) * PXW, y = vid / (int)gridDim.x;
    const int posmin = RIGHT ? x0 + p.d0 : x0 - p.d0 - 4 * G::NJ;
    const size_t plane = (size_t)H * W;

    {   // BGRX tiles, replicate-clamped (M.cpp:1485-1486, 1490)
        uint32_t* sL = reinterpret_cast<uint32_t*>(smem + G::OFF_L);
        uint32_t* sR = reinterpret_cast<uint32_t*>(smem + G::OFF_R);
        for (int i = tid; i < KS * LWC; i += G::NT) {
            const int r = i / LWC, c = i - r * LWC;
            const int yy = min(max(y - HH + r, 0), H - 1), xx = min(max(x0 - HH + c, 0), W - 1);
            sL[i] = imgL[(size_t)yy * W + xx];
        }
        for (int i = tid; i < KS * G::RWC; i += G::NT) {
            const int r = i / G::RWC, c = i - r * G::RWC;
            const int yy = min(max(y - HH + r, 0), H - 1), xx = min(max(posmin - HH + c, 0), W - 1);
            sR[i] = imgR[(size_t)yy * W + xx];
        }
    }
    // lane -> (pixel group g, position block jl), as in k_bilateral_xq.hip
    const int g = (lane & 7) | ((lane >> 2) & 8);
    const int jl = 4 * wave + ((lane >> 3) & 3);
    const int qrel = RIGHT ? 4 * g + 4 * jl : 4 * g + 4 * (G::NJ - jl);
    const int qrel2 = jl == 0 ? (RIGHT ? 4 * g + 4 * G::NJ : 4 * g) : qrel;  // block NJ's positions for the wrapped units of block 0
    const int xabs = x0 + 4 * g;
    const int dbase = p.d0 + 4 * jl;
    const int dbase2 = jl == 0 ? p.d0 + 4 * G::NJ : dbase;
    const int xl = min(x0 + lane, W - 1);                   // the pixel whose left weights this lane stages
    const uint16_t* wLrow = wL + (size_t)y * W;
    const uint16_t* wRrow = wR + (size_t)y * W;

    double num[4][4], den[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) { num[a][b] = 0.0; den[a][b] = 0.0; }

    stage_weights<NWAVE>(0, wLrow, wRrow, plane, smem, wave, lane, xl, posmin, W);
    __syncthreads();
    if (EDGE || wave == 0)
        run_all_steps<NWAVE, EDGE, true, RIGHT>(smem, wLrow, wRrow, plane, wave, lane, xl, g, qrel, qrel2, xabs, dbase, dbase2, W, x0, posmin, num, den);
    else
        run_all_steps<NWAVE, EDGE, false, RIGHT>(smem, wLrow, wRrow, plane, wave, lane, xl, g, qrel, qrel2, xabs, dbase, dbase2, W, x0, posmin, num, den);

    // ---- E = num / den (M.cpp:1496; 0/0 = NaN for windows flat in both images, App. B-9) -> LDS [candidate][pixel]
    double* sE = reinterpret_cast<double*>(smem);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int c = RIGHT ? (b < a ? dbase2 : dbase) - p.d0 + b - a : (a < b ? dbase2 : dbase) - p.d0 + a - b;  // in [0, NFIN)
            sE[c * PXW + 4 * g + a] = num[a][b] / den[a][b];
        }
    __syncthreads();
    const int px = lane, part = wave;
    const int x = x0 + px;
    double be = 1.7976931348623157e308;  // numeric_limits<double>::max(), M.cpp:1458
    float bd = 0.0f;
    if (x < W) {
        for (int c = 16 * part; c < 16 * part + 16; c++) {
            const double E = sE[c * PXW + px];
            if (vol) vol[((size_t)(p.cbase + c) * H + y) * W + x] = (float)E;
            if (E < be) { be = E; bd = (float)(p.d0 + c); }  // M.cpp:1523-1528
        }
    }
    double* sPE = reinterpret_cast<double*>(smem + G::OFF_PART);
    float* sPD = reinterpret_cast<float*>(smem + G::OFF_PART + NWAVE * PXW * 8);
    sPE[part * PXW + px] = be;
    sPD[part * PXW + px] = bd;
    __syncthreads();
    if (tid < PXW && x < W) {
        double e = 1.7976931348623157e308;
        float d = 0.0f;
#pragma unroll
        for (int q = 0; q < NWAVE; q++) {
            const double eq = sPE[q * PXW + tid];
            if (eq < e) { e = eq; d = sPD[q * PXW + tid]; }
        }
        outE[(size_t)y * W + x] = e;
        outD[(size_t)y * W + x] = d;
    }
}

template <int NWAVE>
int launch_pass(hipStream_t s, hipStream_t s_border, const GeoXqParams& base, const uint32_t* imgL, const uint32_t* imgR,
                const uint16_t* wL, const uint16_t* wR, float* vol, double* outE, float* outD, bool right)
{
    const int W = base.W, H = base.H;
    const int ntiles = (W + PXW - 1) / PXW;
    if (right) {  // border tile = the first of a row
        GeoXqParams p = base;
        p.tile0 = 0;
        hipLaunchKernelGGL((k_asw_geodesic_xq<NWAVE, true, true>), dim3(1, H), dim3(64 * NWAVE), 0, s_border, p, imgL, imgR, wL, wR, vol, outE, outD);
        if (ntiles > 1) {
            p.tile0 = 1;
            hipLaunchKernelGGL((k_asw_geodesic_xq<NWAVE, false, true>), dim3(ntiles - 1, H), dim3(64 * NWAVE), 0, s, p, imgL, imgR, wL, wR, vol, outE, outD);
        }
        ASW_HIP_TRY(hipGetLastError());
        return ASW_OK;
    }
    const int n_int = W >= PXW + HH ? std::min(ntiles, (W - PXW - HH) / PXW + 1) : 0;  // x0 + 63 + 7 <= W - 1
    if (n_int > 0) {
        GeoXqParams p = base;
        p.tile0 = 0;
        hipLaunchKernelGGL((k_asw_geodesic_xq<NWAVE, false, false>), dim3(n_int, H), dim3(64 * NWAVE), 0, s, p, imgL, imgR, wL, wR, vol, outE, outD);
    }
    if (ntiles > n_int) {
        GeoXqParams p = base;
        p.tile0 = n_int;
        hipLaunchKernelGGL((k_asw_geodesic_xq<NWAVE, true, false>), dim3(ntiles - n_int, H), dim3(64 * NWAVE), 0, s_border, p, imgL, imgR, wL, wR, vol, outE, outD);
    }
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

}  // namespace

// Candidates a pass of `nwave` wavefronts finishes (nwave = 8 or 4).
int geodesic_xq_pass_candidates(int nwave) { return 16 * nwave; }

// One pass: candidates [cbase, cbase + 16 * nwave) of a win = 15 problem; winners -> outE / outD ([H][W]).
// right: DISPARITY_RIGHT -- imgL / wL are then the fixed (right) image's planes, imgR / wR the left image's.
int launch_geodesic_xq(hipStream_t s, hipStream_t s_border, int nwave, const uint32_t* imgL, const uint32_t* imgR, const uint16_t* wL,
                       const uint16_t* wR, int H, int W, int minD, int cbase, float* vol, double* outE, float* outD, bool right)
{
    GeoXqParams p{H, W, minD + cbase, cbase, 0};
    if (nwave == 8) return launch_pass<8>(s, s_border, p, imgL, imgR, wL, wR, vol, outE, outD, right);
    if (nwave == 4) return launch_pass<4>(s, s_border, p, imgL, imgR, wL, wR, vol, outE, outD, right);
    return ASW_ERR_BAD_ARGUMENT;
}
