// Classic bilateral adaptive-support-weight aggregation (computeAdaptiveWeight, M.cpp:1016-1156), second kernel form:
// thread = 4 pixels x 4 right-image positions ("xq blocking").  Used for DISPARITY_LEFT, 15x15 windows and long candidate
// ranges (the reference's call site at 1080p, configs C5/C3-sized frames); everything else stays on k_asw_bilateral.
//
// Per (pixel x, candidate d) the reference adds, in tap order i (kernel_x = i / 15 outer, kernel_y = i % 15 inner):
//     ab   = wL_i(y,x) * wR_i(y, max(0,x-d))          f32 product
//     num += ab * |gL(ny,nx) - gR(ny, max(0,nx-d))|   f64        nx = clamp(x - 7 + kernel_x), ny = clamp(y - 7 + kernel_y)
//     den += ab                                       f64
// Put q = x - d (the right-image position of the pair).  Then
//     * the weight pair of a unit is (wL_i(x), wR_i(q))                       -> outer product over a block of x's and q's
//     * away from the right image border its cost sample is |gL(x+kx-7) - gR(q+kx-7)|: the SAME value that the unit
//       (x+1, q+1) -- same d, next pixel -- needs one tap column earlier.
// A thread owns pixels X..X+3 and positions Q..Q+3 (16 units, d = X+a - (Q+b)); unit (a,b) runs b tap columns behind the
// thread's step counter K (kx = K - b), so all units on a diagonal a-b (one d, up to four pixels) consume ONE cost value per
// step: 7 f64 subtractions per 16 units, and nothing is converted from bytes in the inner loop.  Every unit still adds its own
// taps in ascending i, in f64, with the exact product -- E is bit-identical to the reference order (fma(ab,c,num) == num + ab*c
// because ab*c is exact: 24-bit x 8-bit significands).
//
// Workgroup = one image row x 64 pixels x 32 position blocks (d = minD + 4j + a - b, j < 32): 8 wavefronts, lane -> (pixel
// group g, position block 4*wave + ((lane >> 3) & 3)); the lane -> g map is chosen for the LDS banks, see the kernel.  All 512
// threads share the staged weights of a step:
//     left : wL(x, tap column kx)   for the 64 pixels        -> ring of 5 tap columns in LDS (a column is used for 4 steps)
//     right: wR(q, tap column K-b)  for the 188 positions    -> b = (q - Q) mod 4 is a property of the position, so the buffer of
//            step K holds, per position, exactly the tap column that position's units consume at step K (double buffered)
// i.e. 7.5 weight evaluations per thread and step instead of 33 with wave-private staging, one workgroup barrier per step
// (18 steps of 15 rows), no cost tile, gray tiles as f64 in LDS.
// Block j = 0 holds six units with d < minD (a < b) and a block j = 32 would hold exactly six units with c = d - minD < 128
// in the same slots (c = 128 + a - b): the j = 0 threads run those instead ("wrapped" units: second position base qrel2,
// second right gray), so the 512 threads finish candidates [0, 128) with no idle unit.  The tail [128, nD) -- one candidate
// at the reference's numDisparity = 128, whose range is inclusive (M.cpp:1021,1074) -- is left to k_asw_bilateral, which
// resumes from this kernel's running minimum.
//
// DISPARITY_RIGHT (M.cpp:1113-1142; RIGHT = true): the fixed image is the right one (the host passes it as gL) and the pair of
// pixel x is the left-image position q = min(W-1, x + d).  Nothing in the blocking depends on the sign of d: unit (a,b) is
// still (pixel X+a, position Q+b) with tap column K - b, d = (Q+b) - (X+a) = minD + 4j + b - a, so the dead units of block 0 are
// those with b < a and the positions run to the right of the tile.  Sample columns clamp the other way round: the tile clamps
// coincide with the reference's at the right image border, not at the left one, so the border tiles are the first of a row.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "asw_internal.h"

namespace {

constexpr int HH = 7;                     // half window
constexpr int KS = 2 * HH + 1;            // 15
constexpr int PXW = 64;                   // pixels per workgroup
constexpr int LWC = PXW + 2 * HH;         // left tile columns  [x0 - 7, x0 + 70]
constexpr int NSTEP = KS + 3;             // 18 steps: unit row b runs b tap columns behind
constexpr int RING = 5;                   // tap columns of left weights kept (4 in use + the one being staged)
constexpr int NCELLCOL = NSTEP + 3;       // cell-table columns kx = -3 .. 17
constexpr int LW8 = 80;                   // u8 row stride of the left tile (multiple of 4)

// Everything that depends on the number of wavefronts NW of a workgroup (8: 128 candidates per pass -- the reference's
// configuration at 1080p; 4: 64 candidates -- its own call site, aswStereoMatch.cpp:94, numDisparity 64 -> 65 candidates):
//   NJ    position blocks per workgroup (+ the wrapped units of block NJ)
//   NPOS  right-image positions a workgroup touches: [posmin, posmin + NPOS - 1]
//   RWC   right tile columns [posmin - 7, posmin + NPOS + 6]
//   NFIN  candidates [0, NFIN) are finished by the kernel
// LDS layout (bytes): double [KS][LWC] left gray | double [KS][RWC] right gray | float [RING][KS][PXW] left weights |
// float [2][KS][NPOS] right weights | u8 [KS][LW8] | u8 [KS][RW8] | u16 [NCELLCOL * KS] packed cell table.
// Epilogue (over the dead tiles): double [NFIN][PXW] E | per-part WTA partials double E[NW][PXW], float d[NW][PXW].
// NW = 8: 81 280 B, two workgroups = 16 wavefronts per CU; NW = 4: 52 KB (f32 gray tiles), three workgroups = 12 wavefronts per CU.
template <int NW>
struct XqCfg {
    static constexpr int NWAVE = NW;
    static constexpr int NTHR = 64 * NW;
    static constexpr int NJ = 4 * NW;
    static constexpr int NPOS = PXW + 4 * NJ;
    static constexpr int RWC = NPOS + 2 * HH;
    static constexpr int NFIN = 4 * NJ;
    static constexpr int RW8 = (RWC + 3) / 4 * 4;
    static constexpr int NROWS = (KS + NW - 1) / NW;  // window rows a wavefront stages
    // Gray tiles for the cost samples: f64 (NW = 8: nothing is converted in the row loop), f32 for NW = 4 -- 13 KB less, 52 KB in
    // all, so that THREE workgroups (12 wavefronts) fit a CU instead of two, for 7 v_cvt_f64_f32 per row step (the differences of
    // two gray bytes are exact in f32)
    static constexpr int GB = NW == 8 ? 8 : 4;
    static constexpr int OFF_LD = 0;
    static constexpr int OFF_RD = (OFF_LD + KS * LWC * GB + 15) / 16 * 16;
    static constexpr int OFF_WL = (OFF_RD + KS * RWC * GB + 15) / 16 * 16;
    static constexpr int OFF_WR = OFF_WL + RING * KS * PXW * 4;
    static constexpr int OFF_L8 = OFF_WR + 2 * KS * NPOS * 4;
    static constexpr int OFF_R8 = OFF_L8 + KS * LW8;
    static constexpr int OFF_CELL = OFF_R8 + KS * RW8;
    static constexpr int LDS_TOTAL = (OFF_CELL + NCELLCOL * KS * 2 + 15) / 16 * 16;
    static constexpr int OFF_E64 = 0;
    static constexpr int OFF_PART = NFIN * PXW * 8;
    typedef typename std::conditional<NW == 8, double, float>::type GrayT;
    static_assert(LDS_TOTAL <= (NW == 8 ? 80 : 53) * 1024, "two (NW = 8) / three (NW = 4) workgroups per CU");
    static_assert(OFF_PART + NW * PXW * 12 <= LDS_TOTAL, "epilogue buffers fit in the dead tiles");
    static_assert(OFF_RD % 16 == 0 && OFF_WL % 16 == 0 && OFF_WR % 16 == 0 && (KS * NPOS * 4) % 16 == 0 && (NPOS * 4) % 16 == 0, "b128 alignment");
    static_assert(NPOS % 64 == 0, "positions are staged in whole wavefront passes");
};
#define XQ_CONSTS(NW)                                                                                                           \
    typedef XqCfg<NW> Cfg;                                                                                                       \
    constexpr int NWAVE = Cfg::NWAVE, NTHR = Cfg::NTHR, NJ = Cfg::NJ, NPOS = Cfg::NPOS, RWC = Cfg::RWC, NFIN = Cfg::NFIN, RW8 = Cfg::RW8,      \
                  NROWS = Cfg::NROWS, OFF_LD = Cfg::OFF_LD, OFF_RD = Cfg::OFF_RD, OFF_WL = Cfg::OFF_WL, OFF_WR = Cfg::OFF_WR,    \
                  OFF_L8 = Cfg::OFF_L8, OFF_R8 = Cfg::OFF_R8, OFF_CELL = Cfg::OFF_CELL, LDS_TOTAL = Cfg::LDS_TOTAL,              \
                  OFF_E64 = Cfg::OFF_E64, OFF_PART = Cfg::OFF_PART;                                                              \
    (void)NWAVE; (void)NTHR; (void)NJ; (void)NPOS; (void)RWC; (void)NFIN; (void)RW8; (void)NROWS; (void)OFF_LD; (void)OFF_RD; (void)OFF_WL; \
    (void)OFF_WR; (void)OFF_L8; (void)OFF_R8; (void)OFF_CELL; (void)LDS_TOTAL; (void)OFF_E64; (void)OFF_PART;

struct XqParams {
    int H, W, minD;
    int tile0;  // first 64-pixel tile of this launch (interior tiles and border tiles are separate launches)
};

__device__ __forceinline__ float lut_at(const float* __restrict__ lut, unsigned idx)
{
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lut) + (idx << 2));
}

// packed cell table in LDS: (dxw + 8) | (dyw + 8) << 4 | class << 8
template <int NW>
__device__ __forceinline__ void cell_at(const unsigned char* smem, int idx, int& dxw, int& dyw, int& cls256)
{
    XQ_CONSTS(NW)
    const uint32_t u = reinterpret_cast<const uint16_t*>(smem + OFF_CELL)[idx];
    dxw = (int)(u & 15u) - 8;
    dyw = (int)((u >> 4) & 15u) - 8;
    cls256 = (int)(u >> 8) << 8;
}

// Stage the weights step `Kn` consumes: left tap column Kn (ring slot Kn % RING) and the right buffer Kn & 1.
// Wave w evaluates window rows ky = w and w + 8.  Cell (kx, ky) -> {dxw, dyw, class}: the direction the reference BUILT
// weight map i for (transposed w.r.t. the sample it is applied to), all-zero class for the skipped cell and
// for kx outside the window.
// Split in two so that the LUT gathers of step Kn are in flight while step Kn - 1 is being accumulated: stage_issue (index
// pass + gathers, at the top of the step) and stage_commit (LDS writes, in the middle of the step's row loop).  Done in one
// piece in front of the step, the staging cost 10 % of the kernel for 5 % of its instructions: eight wavefronts waiting for
// the same two memory round trips (ablation: profiles/r02/ablation_*.csv).
template <int NW> struct Staged { float v[XqCfg<NW>::NROWS][1 + XqCfg<NW>::NPOS / 64]; };

template <int NW>
__device__ __forceinline__ void stage_issue(int Kn, const float* __restrict__ lut, const unsigned char* smem, int wave, int lane,
                                            int ctrL, int pclamp_lo, int pclamp_hi, Staged<NW>& st)
{
    XQ_CONSTS(NW)
    const uint8_t* sL8 = smem + OFF_L8;
    const uint8_t* sR8 = smem + OFF_R8;
#pragma unroll
    for (int rr = 0; rr < NROWS; rr++) {
        const int ky = wave + NWAVE * rr;
        if (ky >= KS) break;  // wave-uniform
        int dxw, dyw, cls;
        if (Kn < KS) {        // left column Kn exists
            cell_at<NW>(smem, (Kn + 3) * KS + ky, dxw, dyw, cls);
            const int nb = sL8[(HH + dyw) * LW8 + (lane + HH + dxw)];
            st.v[rr][0] = lut_at(lut, __builtin_amdgcn_sad_u16(nb, ctrL, cls));
        }
#pragma unroll
        for (int r3 = 0; r3 < NPOS / 64; r3++) {
            const int p = lane + 64 * r3;
            const int b = p & 3;  // posmin == Q (mod 4): the unit row a position belongs to is a property of the position
            cell_at<NW>(smem, (Kn - b + 3) * KS + ky, dxw, dyw, cls);
            // the weight is evaluated AT max(0, x - d) (M.cpp:1105); its neighbour is clamped from there (tile columns are
            // replicate-clamped, so adding the direction needs no further clamp)
            const int pc = min(max(p, pclamp_lo), pclamp_hi) + HH;  // tile column of the clamped position
            const int ctr = sR8[HH * RW8 + pc];
            const int nb = sR8[(HH + dyw) * RW8 + pc + dxw];
            st.v[rr][1 + r3] = lut_at(lut, __builtin_amdgcn_sad_u16(nb, ctr, cls));
        }
    }
}

template <int NW>
__device__ __forceinline__ void stage_commit(int Kn, unsigned char* smem, int wave, int lane, const Staged<NW>& st)
{
    XQ_CONSTS(NW)
    float* sWL = reinterpret_cast<float*>(smem + OFF_WL) + (Kn % RING) * (KS * PXW);
    float* sWR = reinterpret_cast<float*>(smem + OFF_WR) + (Kn & 1) * (KS * NPOS);
#pragma unroll
    for (int rr = 0; rr < NROWS; rr++) {
        const int ky = wave + NWAVE * rr;
        if (ky >= KS) break;
        if (Kn < KS) sWL[ky * PXW + lane] = st.v[rr][0];
#pragma unroll
        for (int r3 = 0; r3 < NPOS / 64; r3++) sWR[ky * NPOS + lane + 64 * r3] = st.v[rr][1 + r3];
    }
}

template <int NW>
__device__ __forceinline__ void stage_weights(int Kn, const float* __restrict__ lut, unsigned char* smem, int wave, int lane,
                                              int ctrL, int pclamp_lo, int pclamp_hi)
{
    Staged<NW> st;
    stage_issue<NW>(Kn, lut, smem, wave, lane, ctrL, pclamp_lo, pclamp_hi, st);
    stage_commit<NW>(Kn, smem, wave, lane, st);
}

// One step: window rows ky = 0..14 of tap column K - b for every active unit row b (BLO <= b <= BHI).
//   EDGE = false: the workgroup's windows never clamp at the right image border: one right gray per step (two with the
//                 wrapped units).
//   EDGE = true : per-diagonal sample columns (lc = min(x + kx - 7, W-1), rc = lc - d), computed once per step.
// Units with a < b (RIGHT: b < a) take their position from qrel2 / dbase2 (== qrel / dbase except in the threads of block
// j = 0, where they are the wrapped units of block 32).
// WRAPW: this wavefront holds the threads of block j = 0 (wave 0).  Elsewhere qrel2 == qrel and the second loads are skipped.
template <int NW, int K, bool EDGE, bool WRAPW, bool COMMIT, bool RIGHT>
__device__ __forceinline__ void run_step(unsigned char* smem, int g, int qrel, int qrel2, int xabs, int dbase, int dbase2,
                                         int W, int x0, int posmin, double (&num)[4][4], double (&den)[4][4], int wave, int lane,
                                         const Staged<NW>& st)
{
    XQ_CONSTS(NW)
    constexpr int BLO = K > KS - 1 ? K - (KS - 1) : 0;  // kx = K - b <= 14
    constexpr int BHI = K < 3 ? K : 3;                  // kx = K - b >= 0
    constexpr int DLO = 0 - BHI, DHI = 3 - BLO;         // diagonals a - b in use
    typedef typename Cfg::GrayT GrayT;
    const GrayT* sLd = reinterpret_cast<const GrayT*>(smem + OFF_LD);
    const GrayT* sRd = reinterpret_cast<const GrayT*>(smem + OFF_RD);
    const float* sWL = reinterpret_cast<const float*>(smem + OFF_WL);
    const float* sWR = reinterpret_cast<const float*>(smem + OFF_WR) + (K & 1) * (KS * NPOS);

    int iL[7], iR[7];
    if constexpr (EDGE) {
#pragma unroll
        for (int dl = DLO; dl <= DHI; dl++) {
            // clamped sample column of the fixed image, then the other image's column from THERE (M.cpp:1101-1106, 1129-1134);
            // the far clamp of that one (max(0, .) / min(W-1, .)) is the tile's
            const int lc = min(max(xabs + dl + K - HH, 0), W - 1);
            const int rc = RIGHT ? lc + ((dl > 0 ? dbase2 : dbase) - dl) : lc - ((dl < 0 ? dbase2 : dbase) + dl);
            iL[dl + 3] = min(max(lc - (x0 - HH), 0), LWC - 1);
            iR[dl + 3] = min(max(rc - (posmin - HH), 0), RWC - 1);
        }
    }
    const GrayT* pl = sLd + 4 * g + K;   // + dl: tile column of x + dl + K - 7
    const GrayT* pr = sRd + qrel + K;    // tile column of Q + K - 7
    const GrayT* pr2 = sRd + qrel2 + K;
    const float* pwl = sWL + 4 * g;
    const float* pwr = sWR + qrel;
    const float* pwr2 = sWR + qrel2;
#pragma unroll 1
    for (int ky = 0; ky < KS; ky++) {
        if constexpr (COMMIT) {
            if (ky == KS / 2) stage_commit<NW>(K + 1, smem, wave, lane, st);  // the gathers issued before this loop have landed
        }
        double c[7];
        if constexpr (!EDGE) {
            const GrayT gr = pr[ky * RWC];
            GrayT gr2 = gr;
            if constexpr (WRAPW && (RIGHT ? DHI > 0 : DLO < 0)) gr2 = pr2[ky * RWC];
            // the left grays of the step, read as 16-byte aligned pairs from an even tile column (4g, LWC and E0 are even): a
            // run that starts on an odd column is split by the compiler into ds_read2_b64, which cost four times the LDS cycles
            // of ds_read_b128 here (8 instead of 4, and 2-way conflicts: their banks are taken mod 32)
            constexpr int E0 = (K + DLO) & ~1, NP2 = (K + DHI - E0 + 2) / 2;
            typedef GrayT gx2 __attribute__((ext_vector_type(2)));
            const gx2* pl2 = reinterpret_cast<const gx2*>(sLd + 4 * g + E0 + ky * LWC);
            GrayT glv[2 * NP2];
#pragma unroll
            for (int i = 0; i < NP2; i++) { const gx2 t = pl2[i]; glv[2 * i] = t.x; glv[2 * i + 1] = t.y; }
#pragma unroll
            for (int dl = DLO; dl <= DHI; dl++) c[dl + 3] = (double)(glv[K + dl - E0] - ((RIGHT ? dl > 0 : dl < 0) ? gr2 : gr));  // exact in f32 too
        } else {
#pragma unroll
            for (int dl = DLO; dl <= DHI; dl++) c[dl + 3] = (double)(sLd[ky * LWC + iL[dl + 3]] - sRd[ky * RWC + iR[dl + 3]]);
        }
        const float4 wr4 = *reinterpret_cast<const float4*>(pwr + ky * NPOS);
        const float wr[4] = {wr4.x, wr4.y, wr4.z, wr4.w};
        float wr2[4] = {wr4.x, wr4.y, wr4.z, wr4.w};
        if constexpr ((WRAPW || EDGE) && (RIGHT ? BLO <= 2 : BHI >= 1)) {  // some active unit is a wrapped one
            const float4 w2 = *reinterpret_cast<const float4*>(pwr2 + ky * NPOS);
            wr2[0] = w2.x; wr2[1] = w2.y; wr2[2] = w2.z; wr2[3] = w2.w;
        }
#pragma unroll
        for (int b = BLO; b <= BHI; b++) {
            const float4 wl4 = *reinterpret_cast<const float4*>(pwl + (((K - b) % RING) * KS + ky) * PXW);
            const float wl[4] = {wl4.x, wl4.y, wl4.z, wl4.w};
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const float ab = wl[a] * ((RIGHT ? b < a : a < b) ? wr2[b] : wr[b]);  // f32 product, M.cpp:1104-1105
                const double abd = (double)ab;
                num[a][b] = __builtin_fma(abd, __builtin_fabs(c[a - b + 3]), num[a][b]);  // exact product: == num + ab*|c|
                den[a][b] = den[a][b] + abd;                                      // M.cpp:1107-1108
            }
        }
    }
}

// ABL (timing experiments only, results are wrong): bit 0 = no weight staging after step 0, bit 1 = no barriers between steps
template <int NW, bool EDGE, bool WRAPW, int ABL, bool RIGHT>
__device__ __forceinline__ void run_all_steps(unsigned char* smem, const float* __restrict__ lut, int wave, int lane, int ctrL,
                                              int pclamp_lo, int pclamp_hi, int g, int qrel, int qrel2, int xabs, int dbase, int dbase2,
                                              int W, int x0, int posmin, double (&num)[4][4], double (&den)[4][4])
{
    Staged<NW> st;
#define ASW_XQ_STEP(KK)                                                                                        \
    if (!(ABL & 1) && (KK) + 1 < NSTEP) {                                                                        \
        if constexpr (EDGE) stage_weights<NW>((KK) + 1, lut, smem, wave, lane, ctrL, pclamp_lo, pclamp_hi);  /* border tiles: registers are scarcer there */ \
        else stage_issue<NW>((KK) + 1, lut, smem, wave, lane, ctrL, pclamp_lo, pclamp_hi, st);                      \
    }                                                                                                           \
    run_step<NW, (KK), EDGE, WRAPW, (!EDGE && !(ABL & 1) && (KK) + 1 < NSTEP), RIGHT>(smem, g, qrel, qrel2, xabs, dbase, dbase2, W, x0, posmin, \
                                                                           num, den, wave, lane, st);          \
    if (!(ABL & 2) || (KK) == NSTEP - 1) __syncthreads();
    ASW_XQ_STEP(0) ASW_XQ_STEP(1) ASW_XQ_STEP(2) ASW_XQ_STEP(3) ASW_XQ_STEP(4) ASW_XQ_STEP(5)
    ASW_XQ_STEP(6) ASW_XQ_STEP(7) ASW_XQ_STEP(8) ASW_XQ_STEP(9) ASW_XQ_STEP(10) ASW_XQ_STEP(11)
    ASW_XQ_STEP(12) ASW_XQ_STEP(13) ASW_XQ_STEP(14) ASW_XQ_STEP(15) ASW_XQ_STEP(16) ASW_XQ_STEP(17)
#undef ASW_XQ_STEP
}

// grid (tiles of this launch, H), 512 threads.  gL / gR: gray planes [H][W].  vol (optional): [>= NFIN][H][W].
// bestE / bestD: [H][W] running minimum over candidates [0, NFIN) (strict '<' in ascending d, M.cpp:1145-1150) for the tail
// launch to resume from; disp (when there is no tail): the disparity itself.
template <int NW, bool EDGE, int ABL, bool RIGHT>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 3, NW == 8 ? 4 : 3))) void k_asw_bilateral_xq(
    XqParams p, const uint8_t* __restrict__ gL, const uint8_t* __restrict__ gR, const int4* __restrict__ cells,
    const float* __restrict__ lut, float* __restrict__ vol, double* __restrict__ bestE, float* __restrict__ bestD,
    float* __restrict__ disp)
{
    // static: the LDS base is then a compile-time 0 and folds into the instructions' offset fields (with the dynamic-LDS
    // symbol every address in the step loop carried its own "v_add_u32 v, <base>, v": 6 of 78 VALU instructions)
    XQ_CONSTS(NW)
    __shared__ __align__(16) unsigned char smem[LDS_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;
    const int x0 = (p.tile0 + blockIdx.x) * PXW, y = blockIdx.y;
    const int posmin = RIGHT ? x0 + p.minD : x0 - p.minD - 4 * NJ;  // == Q (mod 4) either way

    // ---- gray tiles, replicate-clamped (M.cpp:1059-1060, 1101-1106), as bytes (weight staging) and as f64 (cost samples)
    {
        uint8_t* sL8 = smem + OFF_L8;
        uint8_t* sR8 = smem + OFF_R8;
        typedef typename Cfg::GrayT GrayT;
        GrayT* sLd = reinterpret_cast<GrayT*>(smem + OFF_LD);
        GrayT* sRd = reinterpret_cast<GrayT*>(smem + OFF_RD);
        for (int i = tid; i < KS * LWC; i += NTHR) {
            const int r = i / LWC, c = i - r * LWC;
            const int yy = min(max(y - HH + r, 0), H - 1), xx = min(max(x0 - HH + c, 0), W - 1);
            const int v = gL[(size_t)yy * W + xx];
            sL8[r * LW8 + c] = (uint8_t)v;
            sLd[r * LWC + c] = (GrayT)v;
        }
        for (int i = tid; i < KS * RWC; i += NTHR) {
            const int r = i / RWC, c = i - r * RWC;
            const int yy = min(max(y - HH + r, 0), H - 1), xx = min(max(posmin - HH + c, 0), W - 1);
            const int v = gR[(size_t)yy * W + xx];
            sR8[r * RW8 + c] = (uint8_t)v;
            sRd[r * RWC + c] = (GrayT)v;
        }
        uint16_t* sCell = reinterpret_cast<uint16_t*>(smem + OFF_CELL);
        for (int i = tid; i < NCELLCOL * KS; i += NTHR) {
            const int4 ci = cells[i];
            sCell[i] = (uint16_t)((uint32_t)(ci.x + 8) | ((uint32_t)(ci.y + 8) << 4) | ((uint32_t)(ci.z >> 8) << 8));
        }
    }
    __syncthreads();

    // lane -> (pixel group g, position block jl): g's low three bits from the lane's low three bits and its bit 3 from lane bit
    // 5, so that the sixteen lanes one ds_read_b128 services together hold eight distinct g's (the f64 gray reads are 32 B
    // apart per g: g and g + 8 share a bank; measured: 2.9e9 -> 0.9e9 conflict cycles per frame)
    const int g = (lane & 7) | ((lane >> 2) & 8);
    const int jl = 4 * wave + ((lane >> 3) & 3);
    // Q - posmin; LEFT: Q = x0 + 4g - minD - 4 jl, RIGHT: Q = x0 + 4g + minD + 4 jl
    const int qrel = RIGHT ? 4 * g + 4 * jl : 4 * g + 4 * (NJ - jl);
    // block 32's positions for the wrapped units of block 0
    const int qrel2 = jl == 0 ? (RIGHT ? 4 * g + 4 * NJ : 4 * g) : qrel;
    const int xabs = x0 + 4 * g;                           // X
    const int dbase = p.minD + 4 * jl;                     // d of the diagonal a == b
    const int dbase2 = jl == 0 ? p.minD + 4 * NJ : dbase;
    const int ctrL = smem[OFF_L8 + HH * LW8 + lane + HH];  // this lane's pixel as the centre of left weights it stages
    // positions are clamped into the image before the weight is looked up: max(0, x - d) / min(W-1, x + d) (the other bound
    // only matters for the lanes of a partial tile, whose results are discarded)
    const int pclamp_lo = min(max(0 - posmin, 0), NPOS - 1), pclamp_hi = min(max(W - 1 - posmin, 0), NPOS - 1);

    double num[4][4], den[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) { num[a][b] = 0.0; den[a][b] = 0.0; }

    stage_weights<NW>(0, lut, smem, wave, lane, ctrL, pclamp_lo, pclamp_hi);
    __syncthreads();
    // wave 0 holds the threads of block j = 0 (the wrapped units): its steps load the second right weights / gray; ONE branch
    // around the whole step sequence (a branch per step made the register allocator spill 488 VGPRs)
    if (EDGE || wave == 0)
        run_all_steps<NW, EDGE, true, ABL, RIGHT>(smem, lut, wave, lane, ctrL, pclamp_lo, pclamp_hi, g, qrel, qrel2, xabs, dbase, dbase2, W,
                                                   x0, posmin, num, den);
    else
        run_all_steps<NW, EDGE, false, ABL, RIGHT>(smem, lut, wave, lane, ctrL, pclamp_lo, pclamp_hi, g, qrel, qrel2, xabs, dbase, dbase2, W,
                                                    x0, posmin, num, den);
    // (the last step ended with a barrier: the tiles are dead)

    // ---- E = num / den (M.cpp:1111) -> LDS [candidate][pixel]
    double* sE = reinterpret_cast<double*>(smem + OFF_E64);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int c = RIGHT ? (b < a ? dbase2 : dbase) - p.minD + b - a
                                : (a < b ? dbase2 : dbase) - p.minD + a - b;  // in [0, 128) for every unit
            sE[c * PXW + 4 * g + a] = num[a][b] / den[a][b];
        }
    __syncthreads();
    // ---- WTA (strict '<' while d ascends) and volume: thread -> (pixel, 16 consecutive candidates)
    const int px = lane, part = wave;
    const int x = x0 + px;
    double be = 1.7976931348623157e308;  // numeric_limits<double>::max(), M.cpp:1037
    float bd = 0.0f;
    if (x < W) {
        for (int c = 16 * part; c < 16 * part + 16; c++) {
            const double E = sE[c * PXW + px];
            if (vol) vol[((size_t)c * H + y) * W + x] = (float)E;
            if (E < be) { be = E; bd = (float)(p.minD + c); }
        }
    }
    double* sPE = reinterpret_cast<double*>(smem + OFF_PART);
    float* sPD = reinterpret_cast<float*>(smem + OFF_PART + NWAVE * PXW * 8);
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
        if (disp) {
            disp[(size_t)y * W + x] = d;
        } else {
            bestE[(size_t)y * W + x] = e;
            bestD[(size_t)y * W + x] = d;
        }
    }
}

}  // namespace

int bilateral_xq_candidates(int nwave) { return 16 * nwave; }

namespace {
template <int NW>
int launch_xq_t(hipStream_t s, hipStream_t s_border, const uint8_t* gL, const uint8_t* gR, int H, int W, int minD, const int4* cells,
                const float* lut, float* vol, double* bestE, float* bestD, float* disp, bool right)
{
    const int ntiles = (W + PXW - 1) / PXW;
    const dim3 blk(64 * NW);
    if (right) {
        // border tile = the first of a row (windows of x < 7 clamp at column 0); the clamped positions' weight neighbours
        // (columns W-8 .. W-1) must lie in the last tile's right-gray tile: x0_last + minD <= W - 1
        if ((ntiles - 1) * PXW + minD > W - 1) return ASW_ERR_BAD_ARGUMENT;
        XqParams pe{H, W, minD, 0};
        hipLaunchKernelGGL((k_asw_bilateral_xq<NW, true, 0, true>), dim3(1, H), blk, 0, s_border, pe, gL, gR, cells, lut, vol,
                           bestE, bestD, disp);
        if (ntiles > 1) {
            XqParams pi{H, W, minD, 1};
            hipLaunchKernelGGL((k_asw_bilateral_xq<NW, false, 0, true>), dim3(ntiles - 1, H), blk, 0, s, pi, gL, gR, cells, lut,
                               vol, bestE, bestD, disp);
        }
        ASW_HIP_TRY(hipGetLastError());
        return ASW_OK;
    }
    // tiles whose windows (of in-image pixels) stay left of the right border: x0 + 63 + 7 <= W - 1
    const int n_int = W >= PXW + HH ? std::min(ntiles, (W - PXW - HH) / PXW + 1) : 0;
#ifdef ASW_XQ_ABLATION  // measurement builds only (HIPCC_EXTRA=-DASW_XQ_ABLATION): the ablated kernels compute wrong results
    int abl = 0;
    if (const char* e = getenv("ASW_XQ_ABLATE")) abl = atoi(e) & 3;  // profiles/r02/ablation/*.csv
    auto ki = abl == 0 ? k_asw_bilateral_xq<NW, false, 0, false> : abl == 1 ? k_asw_bilateral_xq<NW, false, 1, false>
            : abl == 2 ? k_asw_bilateral_xq<NW, false, 2, false> : k_asw_bilateral_xq<NW, false, 3, false>;
#else
    auto ki = k_asw_bilateral_xq<NW, false, 0, false>;
#endif
    auto ke = k_asw_bilateral_xq<NW, true, 0, false>;
    if (n_int > 0) {
        XqParams p{H, W, minD, 0};
        hipLaunchKernelGGL(ki, dim3(n_int, H), blk, 0, s, p, gL, gR, cells, lut, vol, bestE, bestD, disp);
    }
    if (ntiles > n_int) {
        XqParams p{H, W, minD, n_int};
        hipLaunchKernelGGL(ke, dim3(ntiles - n_int, H), blk, 0, s_border, p, gL, gR, cells, lut, vol, bestE, bestD, disp);
    }
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}
}  // namespace

// nwave: 8 = candidates [0, 128), 4 = candidates [0, 64).
// cells: int4[21 * 15] {dxw, dyw, class * 256, -} per window cell, kx = -3..17; lut: float[ncls][256] with an all-zero class.
// disp != nullptr: the launch covers the whole candidate range: write the disparity; else bestE / bestD.
// s_border: stream of the border-tile launch (may equal s; a side stream lets the 1/30 of the tiles overlap the main launch)
// right: DISPARITY_RIGHT -- gL is then the fixed (right) image's gray plane and gR the left image's.
int launch_bilateral_xq(hipStream_t s, hipStream_t s_border, int nwave, const uint8_t* gL, const uint8_t* gR, int H, int W, int minD,
                        const int4* cells, const float* lut, float* vol, double* bestE, float* bestD, float* disp, bool right)
{
    if (nwave == 8) return launch_xq_t<8>(s, s_border, gL, gR, H, W, minD, cells, lut, vol, bestE, bestD, disp, right);
    if (nwave == 4) return launch_xq_t<4>(s, s_border, gL, gR, H, W, minD, cells, lut, vol, bestE, bestD, disp, right);
    return ASW_ERR_BAD_ARGUMENT;
}
