// Guided-filter aggregation (getGuidedFilter, M.cpp:2766-2854) and the box-mean machinery it is made of.
//
// boxFilter(CV_32F, Size(k,k), normalised, BORDER_REFLECT_101) == f64 window sum * 1/(k*k) -> f32
// (SURVEY App. A-9).  One generic "column walk" kernel evaluates NP box means at once:
//   * a wavefront owns 64 input columns (64-(k-1) output columns) of a band of rows and walks down the
//     band; every lane keeps NP vertical running sums in f64 (add the entering row, subtract the leaving
//     one -- the same sliding form as OpenCV's ColumnSum; the leaving row is recomputed from cached loads);
//   * per output row the vertical sums go through a wave-private LDS strip (no workgroup barrier) and each
//     output column adds its k neighbours in ascending order (conflict-free ds_read_b64);
//   * the producer (Src) and consumer (Dst) are functors, so normalisation, products, covariance,
//     a = cov/(var+eps), b and q are fused into the filters that need them and never hit HBM as
//     separate planes.
// The kernels are bound by f64 VALU + HBM streaming of the a/b planes, see DESIGN.md.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "asw_device.h"
#include "asw_internal.h"

namespace {


constexpr int BW = 256;  // threads per block = 4 independent wavefronts
struct NoRaw {};

// Every wavefront owns a strip of 128 input columns (two adjacent ones per lane, 128-(k-1) output columns)
// of a band of rows and walks down the band on its own:
//   * vertical running sums in f64 registers (add the entering row, subtract the leaving one -- ColumnSum's
//     sliding form; the leaving row is re-fetched from cache rather than kept: an LDS ring cost 61 KB per
//     workgroup and 2x the run time);
//   * horizontal sums through a wave-private LDS strip: LDS operations of one wavefront execute in order, so
//     no workgroup barrier is needed; a lane's second column reuses the first one's sum (- b[0] + b[k]).
//   * ND slices per wavefront: operands that do not depend on the slice (guide pixel, guide statistics) are
//     fetched once -- the ND fetches are issued back to back with identical addresses and merge (CSE).
//   * NANSAFE: the inputs may hold NaN (0/0 NCC costs of flat windows, M.cpp:867-868).  A sliding sum never loses a NaN once it
//     has entered (NaN - NaN = NaN), whereas a window sum is NaN only while the NaN is inside the window -- the form the CPU
//     restatement defines.  Whenever a running sum is not finite it is therefore rebuilt from the k rows of its window (and
//     the shared horizontal sum of the second column from its own k terms); finite data never takes these branches.
// KT: the window size when it is known at compile time (15: the reference's call site), 0 = run-time k.  With a run-time k
// the horizontal sum is a loop of dependent LDS round trips (8 terms, then one per iteration); with KT its 16 doubles per plane
// are read at once.
// RING (needs KT = 15): the fetched operands of the last KT rows stay in a register ring (Src::KEEP dwords per column and
// row, 16-entry vectors indexed with the wave-uniform slot s mod KT: s_set_gpr_idx_on / v_mov, the row loop is NOT unrolled),
// so the leaving row is never fetched again.  Without the ring the q pass of the guided filter fetched 11.5 GB to read 4.05 GB
// of a/b planes (between a row's first and second fetch the wavefronts of an XCD stream 15 MB through its 4 MB L2), and short
// bands were needed to keep that re-read in L2 at all; with it a band can be as tall as the launch geometry allows (k-1
// warm-up rows per band: 44 % of the work at 32 rows, 10 % at 135).
typedef uint32_t v16u __attribute__((ext_vector_type(16)));
template <int NP, int CPL, int ND, int WPE, bool NANSAFE, int KT, int RING, int PF, class Src, class Dst>
__global__ __launch_bounds__(BW) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_box_walk(Src src, Dst dst, int H, int W, int k_rt, int band, int nxw, int nslices, int ngx, int nby, int slice_par)
{
    static_assert(!RING || KT == 15, "the register ring is a 16-entry vector per kept dword");
    static_assert(PF >= 1 && PF <= 3, "depth of the software pipeline");
    const int k = KT ? KT : k_rt;
    constexpr int SW = 64 * CPL;  // strip width (input columns per wavefront)
    extern __shared__ __align__(16) unsigned char smem[];
    // the wavefront index is uniform within a wavefront: as an SGPR it makes strip, band, slice and every base address
    // derived from them scalar
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* hs = reinterpret_cast<double*>(smem) + (size_t)wv * ND * NP * (SW + 2);  // [ND][NP][SW+2] per wavefront
    const int hl = k / 2;  // OpenCV anchor = k/2 (also for even k)
    const int XO = SW - (k - 1);
    // Workgroup -> (region, slice group), XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so the ones with
    // equal blockIdx.x % 8 share an L2.  Each XCD takes every 8th region (a band of 4 strips) and runs through ALL
    // slices of it before the next region, so whatever does not depend on the slice (guide pixels, guide statistics)
    // and the halo rows/columns are fetched from HBM once per region and hit in that XCD's L2 afterwards.
    // The four wavefronts of a workgroup take four strips of one slice group, or (slice_par) one strip of four
    // consecutive slice groups: they then walk the same rows at the same time and the slice-independent loads of three
    // of them hit in the CU's L1.
    const int spw = slice_par ? 4 : 1;
    const int nzg = ((nslices + ND - 1) / ND + spw - 1) / spw;
    const int wj = blockIdx.x >> 3;
    // an XCD takes a contiguous run of regions: the regions it has in flight at any time are neighbouring strips of one band,
    // whose halo columns (and the 128-byte lines two strips share) are then fetched from HBM once
    const int rpx = (ngx * nby + 7) >> 3;
    const int reg = (blockIdx.x & 7) * rpx + wj / nzg;
    if (wj / nzg >= rpx || reg >= ngx * nby) return;
    const int gx = reg % ngx, by = reg / ngx;
    const int xw = slice_par ? gx : gx * 4 + wv;              // wavefront's strip index
    const int zg = slice_par ? (wj % nzg) * 4 + wv : wj % nzg;  // wavefront's slice group
    if (xw >= nxw || zg * ND >= nslices) return;              // whole wavefront exits
    const int xo0 = xw * XO;
    const int c0 = CPL * lane;           // first strip column of this lane
    int kz[ND];
    bool kvalid[ND];
#pragma unroll
    for (int n = 0; n < ND; n++) {
        kvalid[n] = zg * ND + n < nslices;
        kz[n] = min(zg * ND + n, nslices - 1);
    }
    int xin[CPL];
    bool out_col[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        xin[c] = reflect101_idx(xo0 - hl + c0 + c, W);
        out_col[c] = (c0 + c < XO) && (xo0 + c0 + c < W);
    }
    const bool any_out = out_col[0];  // columns are adjacent: column 1 is an output only if column 0 is
    if (!dst.active(kz[0], xo0, min(xo0 + XO, W) - 1)) return;  // consumers may skip whole slices / strips (wave-uniform)
    // Everything a producer / consumer needs that depends only on (column, slice) -- reflected or shifted column indices,
    // slice base pointers, normalisation parameters, key values -- is resolved ONCE here; the row walk adds the
    // wave-uniform row offset only (the per-row index arithmetic used to be a third of the instructions of a step).
    typename Src::Col scol[CPL][ND];
    typename Dst::Col dcol[CPL][ND];
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int n = 0; n < ND; n++) {
            scol[c][n] = src.col(xin[c], kz[n]);
            dcol[c][n] = dst.col(min(xo0 + c0 + c, W - 1), kz[n]);
        }
    const int y0 = by * band, y1 = min(H, y0 + band);
    const double scale = 1.0 / ((double)k * (double)k);
    double vs[CPL][ND][NP];
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int n = 0; n < ND; n++)
#pragma unroll
            for (int p = 0; p < NP; p++) vs[c][n][p] = 0.0;

    constexpr int NK = RING ? Src::KEEP : 1;
    v16u ring[CPL][ND][NK];
    if constexpr (RING) {
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (int n = 0; n < ND; n++)
#pragma unroll
                for (int j = 0; j < NK; j++) ring[c][n][j] = 0;
    }
    int slot = 0;  // s mod k, wave-uniform
    const int steps = (y1 - y0) + k - 1;

    // One step = one input row.  The walk is three straight-line loops (warm-up: accumulate only; the first output row: nothing
    // leaves yet; steady state) instead of one loop full of `s >= k` branches: around a conditional load the compiler puts the
    // wait for it right behind the load (a full memory round trip, twice per step), and nothing moves across the branches.
    // Software pipeline of depth PF: EVERY load of step s -- entering row, what the leaving row still needs from memory, the
    // consumer's operands of the output row -- is issued at the top of step s-PF into a register FIFO of PF+1 slots (the step
    // body exists once per slot: the slot index is a compile-time phase).  Two reasons: (i) memory-level parallelism -- a
    // wavefront with one row in flight at 2-3 wavefronts per SIMD keeps ~20 KB per CU in flight, short of what 5 TB/s need;
    // (ii) vmcnt counts loads and stores in issue order on gfx9, so a load issued after the previous step's stores cannot be
    // waited for without waiting for those stores as well; issued a step earlier, every wait names only older loads.
    constexpr int NPH = PF + 1;
    typename Src::Raw fN[NPH][CPL][ND];
    using LeaveT = typename std::conditional<RING != 0, typename Src::LRaw, typename Src::Raw>::type;  // what the leaving row loads
    LeaveT fL[NPH][CPL][ND];
    typename Dst::Raw fD[NPH][CPL][ND];
    auto issue = [&](int s, auto slot_c) {  // loads of step s into FIFO slot slot_c
        constexpr int SL = decltype(slot_c)::value;
        const int yn = reflect101_idx(y0 - hl + s, H);  // steps past the end: valid rows, never used
        const int yo = reflect101_idx(y0 - hl + s - k, H);
        const int yd = min(max(y0 + s - (k - 1), y0), y1 - 1);
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (int n = 0; n < ND; n++) {
                fN[SL][c][n] = src.fetch(yn, scol[c][n]);
                if constexpr (RING) fL[SL][c][n] = src.leave_fetch(yo, scol[c][n]);
                else fL[SL][c][n] = src.fetch(yo, scol[c][n]);
                fD[SL][c][n] = dst.fetch(yd, dcol[c][n]);
            }
    };
    if constexpr (PF >= 1) issue(0, std::integral_constant<int, 0>());
    if constexpr (PF >= 2) issue(1, std::integral_constant<int, 1>());
    if constexpr (PF >= 3) issue(2, std::integral_constant<int, 2>());

    auto step = [&](int s, auto ph_c, auto sub_c, auto out_c) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr bool SUB = decltype(sub_c)::value, OUT = decltype(out_c)::value;
        issue(s + PF, std::integral_constant<int, (PH + PF) % NPH>());
        typename Src::Raw (&rn)[CPL][ND] = fN[PH];
        typename Dst::Raw (&rd)[CPL][ND] = fD[PH];
        typename Src::Raw ro[CPL][ND];
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (int n = 0; n < ND; n++) {
                if constexpr (SUB) {
                    if constexpr (RING) {
                        uint32_t w[NK];
#pragma unroll
                        for (int j = 0; j < NK; j++) w[j] = ring[c][n][j][slot];
                        ro[c][n] = src.leave(w, fL[PH][c][n]);
                    } else {
                        ro[c][n] = fL[PH][c][n];
                    }
                }
            }
        if constexpr (RING) {
#pragma unroll
            for (int c = 0; c < CPL; c++)
#pragma unroll
                for (int n = 0; n < ND; n++) {
                    uint32_t w[NK];
                    src.keep(rn[c][n], w);
#pragma unroll
                    for (int j = 0; j < NK; j++) ring[c][n][j][slot] = w[j];
                }
            slot = slot + 1 == k ? 0 : slot + 1;
        }
        // ---- vertical running sums (ColumnSum: SUM -= leaving row, SUM += entering row) ----
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (int n = 0; n < ND; n++) {
                if constexpr (SUB) {
                    float o[NP];
                    src.eval(ro[c][n], scol[c][n], o);
#pragma unroll
                    for (int p = 0; p < NP; p++) vs[c][n][p] = vs[c][n][p] - (double)o[p];
                }
                float v[NP];
                src.eval(rn[c][n], scol[c][n], v);
#pragma unroll
                for (int p = 0; p < NP; p++) vs[c][n][p] = vs[c][n][p] + (double)v[p];
                if constexpr (NANSAFE && OUT) {
                    bool poisoned = false;
#pragma unroll
                    for (int p = 0; p < NP; p++) poisoned = poisoned || !__builtin_isfinite(vs[c][n][p]);
                    if (poisoned) {  // rebuild the window sum of rows s-k+1 .. s (ascending)
                        double acc[NP];
#pragma unroll
                        for (int p = 0; p < NP; p++) acc[p] = 0.0;
                        for (int i = k - 1; i >= 0; i--) {
                            const typename Src::Raw rr = src.fetch(reflect101_idx(y0 - hl + s - i, H), scol[c][n]);
                            float w[NP];
                            src.eval(rr, scol[c][n], w);
#pragma unroll
                            for (int p = 0; p < NP; p++) acc[p] = acc[p] + (double)w[p];
                        }
#pragma unroll
                        for (int p = 0; p < NP; p++) vs[c][n][p] = acc[p];
                    }
                }
            }
        if constexpr (OUT) {
#pragma unroll
            for (int c = 0; c < CPL; c++)
#pragma unroll
                for (int n = 0; n < ND; n++)
#pragma unroll
                    for (int p = 0; p < NP; p++) hs[(n * NP + p) * (SW + 2) + c0 + c] = vs[c][n][p];
            // Same-wavefront LDS traffic is ordered in hardware: the reads below see the writes above without a workgroup
            // barrier.  The COMPILER must be told that other lanes read these words: to a single thread its own store
            // (offset c0) and its loads (offsets c0+1 ..) never alias, and LLVM promotes the stored value to a register and
            // sinks the store out of the row loop (seen with the straight-line walk: every row after the first was wrong).
            // Wavefront-scope fences cost no instruction.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (any_out) {
                const int y = y0 + s - (k - 1);
#pragma unroll
                for (int n = 0; n < ND; n++) {
                    float m[CPL][NP];
#pragma unroll
                    for (int p = 0; p < NP; p++) {
                        const double* b = hs + (n * NP + p) * (SW + 2) + c0;
                        double sum = 0.0, b0, bk = 0.0;
                        if constexpr (KT > 0) {
                            double bb[KT + 1];
#pragma unroll
                            for (int i = 0; i < KT + (CPL > 1 ? 1 : 0); i++) bb[i] = b[i];
#pragma unroll
                            for (int i = 0; i < KT; i++) sum = sum + bb[i];
                            b0 = bb[0];
                            if constexpr (CPL > 1) bk = bb[KT];
                        } else {
                            for (int i = 0; i < k; i++) sum = sum + b[i];
                            b0 = b[0];
                            if constexpr (CPL > 1) bk = b[k];
                        }
                        m[0][p] = (float)(sum * scale);
                        if constexpr (CPL > 1) {
                            double sum1 = (sum - b0) + bk;  // window of the adjacent column
                            if constexpr (NANSAFE) {
                                if (!__builtin_isfinite(sum1)) {
                                    sum1 = 0.0;
                                    for (int i = 1; i <= k; i++) sum1 = sum1 + b[i];
                                }
                            }
                            m[CPL - 1][p] = (float)(sum1 * scale);
                        }
                    }
                    if (kvalid[n]) {
                        dst.emit(y, dcol[0][n], rd[0][n], m[0]);
                        if constexpr (CPL > 1) {
                            if (out_col[CPL - 1]) dst.emit(y, dcol[CPL - 1][n], rd[CPL - 1][n], m[CPL - 1]);
                        }
                    }
                }
            }
            // the next row's stores stay behind this row's loads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    int s = 0, ph = 0;
    auto run = [&](int s_end, auto sub_c, auto out_c) {
        for (; s < s_end; s++) {
            if (ph == 0) step(s, std::integral_constant<int, 0>(), sub_c, out_c);
            if constexpr (NPH > 1) { if (ph == 1) step(s, std::integral_constant<int, 1 % NPH>(), sub_c, out_c); }
            if constexpr (NPH > 2) { if (ph == 2) step(s, std::integral_constant<int, 2 % NPH>(), sub_c, out_c); }
            if constexpr (NPH > 3) { if (ph == 3) step(s, std::integral_constant<int, 3 % NPH>(), sub_c, out_c); }
            ph = ph + 1 == NPH ? 0 : ph + 1;
        }
    };
    run(min(k - 1, steps), F(), F());
    run(min(k, steps), F(), T());
    run(steps, T(), T());
}

// ---- guide access: normalised guide channels I_c(y,x) for slice k --------------------------------
// The guide lives in packed BGRX planes (one dword per pixel): channels 0-2 from A at x; channels 3-5 from B
// at x, or either plane at reflect(x + shift*d): the disparity-shifted view of computeAdaptiveWeight_GuidedF
// (LEFT: right image at x-d, M.cpp:2907-2912; RIGHT: left image at x+d, M.cpp:2925-2929).
template <bool SHIFT>
struct GuideAccT {
    const uint32_t* A;
    const uint32_t* B;
    const float2* scales;  // normalize() scale/shift per slice (index k * scale_stride)
    int scale_stride;
    int W, shiftA, shiftB, minD;
    struct Col { const uint32_t* a; const uint32_t* b; float2 sc; };
    __device__ __forceinline__ Col col(int x, int k) const
    {
        Col c;
        if constexpr (SHIFT) {
            c.a = A + (shiftA ? reflect_idx(x + shiftA * (minD + k), W) : x);
            c.b = B + (shiftB ? reflect_idx(x + shiftB * (minD + k), W) : x);  // never dereferenced when B is null (NW = 1)
            c.sc = scales[k * scale_stride];
        } else {  // slice-independent
            c.a = A + x;
            c.b = B + x;
            c.sc = scales[0];
        }
        return c;
    }
    template <int NW>
    __device__ __forceinline__ void fetch(int y, const Col& c, uint32_t (&u)[NW]) const
    {
        const size_t row = (size_t)y * W;
        u[0] = c.a[row];
        if constexpr (NW > 1) u[1] = c.b[row];
    }
    template <int NW>
    __device__ __forceinline__ void eval(const uint32_t (&u)[NW], const Col& c, float (&I)[3 * NW]) const
    {
        const float2 sc = c.sc;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            // convertTo 8u->32f with float scale/shift (App. A-10): v_cvt_f32_ubyteN, mul, add
            I[3 * w + 0] = (float)(u[w] & 0xffu) * sc.x + sc.y;
            I[3 * w + 1] = (float)((u[w] >> 8) & 0xffu) * sc.x + sc.y;
            I[3 * w + 2] = (float)((u[w] >> 16) & 0xffu) * sc.x + sc.y;
        }
    }
};


// Guide statistics, interleaved per pixel and per BGRX word: {meanI_0..2, den_0..2, pad, pad} = 8 floats, two dwordx4
// loads for the consumer.  A 3-channel guide has one such array [slot][H][W][8]; a 6-channel guide has one per word plus,
// when the guide is [fixed image, other image shifted by d] (GuidedF / GuidedF_3), a third one with the statistics of the
// other image WITHOUT the shift:
//   * the fixed word's statistics depend on the slice only through the normalisation scale, which is the same for whole
//     groups of slices (rep[k] = first slice with the same scale/shift): computed for the representatives only;
//   * away from the image border the box window of the shifted view holds exactly the pixels of the unshifted image at
//     column x + sgn*d, so its statistics are read from the third array at that column; only the columns whose window
//     touches the border or the reflected part of the view (d + r/2 columns on one side, r/2 on the other) are computed
//     per slice.
// GuidedF at 1080p D=128: two per-slice statistics passes of 5.9 ms each -> 0.3 + 0.3 + 1.0 ms.
constexpr int SS8 = 8;
struct StatsSplit {
    float* half[2];    // [slot][H][W][8] per BGRX word, as the guide shows the word (shifted view included)
    float* unshifted;  // [slot][H][W][8] of the shifted word's image without the shift, slot = group representative
    const int* rep;    // group representative of every slice; null: no sharing (every slice has its own statistics)
    int per_slice;     // 0: one slot (slice-independent guide)
    int shifted;       // which word is shifted (0 / 1); -1: none
    int sgn;           // the shifted word is read at x + sgn*d
    int lo, hi;        // box window of output column x = [x - lo, x + hi]
    int W, minD;
    __device__ __forceinline__ bool interior(int x, int k) const
    {
        const int d = minD + k;
        return sgn < 0 ? (x >= d + lo && x <= W - 1 - hi) : (x >= lo && x <= W - 1 - hi - d);
    }
};

// box(I_c), box(I_c*I_c) -> meanI_c, den_c = (corrI_c - meanI_c^2) + eps      (M.cpp:2778, 2796-2799, 2846)
template <int C, int W0, bool SHIFT>  // W0: which BGRX word (channels 3*W0 .. 3*W0+2) this launch covers
struct StatsSrc {
    static const char* band_env() { return "ASW_BAND_STATS"; }
    GuideAccT<SHIFT> g;
    typedef typename GuideAccT<SHIFT>::Col Col;
    struct Raw { uint32_t u[C / 3]; };
    __device__ __forceinline__ Col col(int x, int k) const { return g.col(x, k); }
    __device__ __forceinline__ Raw fetch(int y, const Col& c) const
    {
        Raw r;
        g.template fetch<C / 3>(y, c, r.u);
        return r;
    }
    static constexpr int KEEP = C / 3;
    __device__ __forceinline__ void keep(const Raw& r, uint32_t (&w)[KEEP]) const
    {
#pragma unroll
        for (int i = 0; i < KEEP; i++) w[i] = r.u[i];
    }
    typedef NoRaw LRaw;
    __device__ __forceinline__ LRaw leave_fetch(int, const Col&) const { return LRaw(); }
    __device__ __forceinline__ Raw leave(const uint32_t (&w)[KEEP], const LRaw&) const
    {
        Raw r;
#pragma unroll
        for (int i = 0; i < KEEP; i++) r.u[i] = w[i];
        return r;
    }
    __device__ __forceinline__ void eval(const Raw& r, const Col& c, float (&v)[6]) const
    {
        float I[C];
        g.template eval<C / 3>(r.u, c, I);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) { v[ch] = I[3 * W0 + ch]; v[3 + ch] = I[3 * W0 + ch] * I[3 * W0 + ch]; }
    }
};
// MODE 0: every slice; 1: group representatives only; 2: only the strips that contain border columns of the slice
template <int MODE>
struct StatsDst {
    float* out;  // [slot][H][W][8]
    StatsSplit sp;
    int H, W;
    float epsf;
    typedef NoRaw Raw;
    struct Col { float* o; };
    __device__ __forceinline__ bool active(int k, int xa, int xb) const
    {
        if constexpr (MODE == 1) return sp.rep[k] == k;
        if constexpr (MODE == 2) return !(sp.interior(xa, k) && sp.interior(xb, k));
        return true;
    }
    __device__ __forceinline__ Col col(int x, int k) const { return Col{out + ((size_t)k * H * W + x) * SS8}; }
    __device__ __forceinline__ Raw fetch(int, const Col&) const { return Raw(); }
    __device__ __forceinline__ void emit(int y, const Col& c, const Raw&, const float (&m)[6]) const
    {
        float* o = c.o + (size_t)y * W * SS8;
        float v[8];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float mm = m[ch] * m[ch];
            float var = m[3 + ch] - mm;
            v[ch] = m[ch];
            v[3 + ch] = 1.0f * epsf + var;  // scaleAdd(ones, eps, var)
        }
        v[6] = 0.0f; v[7] = 0.0f;
        reinterpret_cast<float4*>(o)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(o)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
};

// first slice with the same normalisation parameters (bitwise): the unshifted guide word looks the same in all of them
__global__ void k_scale_groups(const float2* __restrict__ scales, int n, int* __restrict__ rep)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float2 me = scales[k];
    int r = k;
    for (int j = k - 1; j >= 0; j--) {
        const float2 o = scales[j];
        if (__float_as_uint(o.x) == __float_as_uint(me.x) && __float_as_uint(o.y) == __float_as_uint(me.y)) r = j;
    }
    rep[k] = r;
}

// a/b planes interleaved per pixel: ab[k][y][x][AS], AS = 4 (C=3) or 8 (C=6): {a_0..a_C-1, b, pad}
template <int C> struct ABStride { static constexpr int value = (C == 3) ? 4 : 8; };

// box(P), box(I_c*P) -> a_c = cov_c / den_c, b = meanP - sum_c a_c*meanI_c      (M.cpp:2780-2847)
// KEEPG: the register ring keeps the guide word(s) next to the cost (C/3 + 1 dwords per column and row); otherwise the cost
// only, and the guide pixel of the leaving row is fetched again (slice-independent: an L1 / L2 hit)
template <int C, bool SHIFT, bool KEEPG = true>
struct ABSrc {
    static const char* band_env() { return "ASW_BAND_AB"; }
    static constexpr int KEEP = KEEPG ? C / 3 + 1 : 1;
    GuideAccT<SHIFT> g;
    const float* P;          // raw cost volume [n][H][W]
    const float2* pscales;   // per-slice normalize() parameters
    int H, W;
    struct Col { typename GuideAccT<SHIFT>::Col g; const float* p; float2 sc; };
    struct Raw { uint32_t u[C / 3]; float p; };
    __device__ __forceinline__ Col col(int x, int k) const { return Col{g.col(x, k), P + (size_t)k * H * W + x, pscales[k]}; }
    __device__ __forceinline__ Raw fetch(int y, const Col& c) const
    {
        Raw r;
        g.template fetch<C / 3>(y, c.g, r.u);
        r.p = c.p[(size_t)y * W];
        return r;
    }
    __device__ __forceinline__ void keep(const Raw& r, uint32_t (&w)[KEEP]) const
    {
        w[0] = __float_as_uint(r.p);
        if constexpr (KEEPG) {
#pragma unroll
            for (int i = 0; i < C / 3; i++) w[1 + i] = r.u[i];
        }
    }
    struct LRaw { uint32_t u[KEEPG ? 1 : C / 3]; };  // the guide word(s) of the leaving row when the ring holds the cost only
    __device__ __forceinline__ LRaw leave_fetch(int yo, const Col& c) const
    {
        LRaw l;
        if constexpr (KEEPG) l.u[0] = 0;
        else g.template fetch<C / 3>(yo, c.g, l.u);
        return l;
    }
    __device__ __forceinline__ Raw leave(const uint32_t (&w)[KEEP], const LRaw& l) const
    {
        Raw r;
#pragma unroll
        for (int i = 0; i < C / 3; i++) r.u[i] = KEEPG ? w[KEEPG ? 1 + i : 0] : l.u[KEEPG ? 0 : i];
        r.p = __uint_as_float(w[0]);
        return r;
    }
    __device__ __forceinline__ void eval(const Raw& r, const Col& c, float (&v)[C + 1]) const
    {
        float I[C];
        g.template eval<C / 3>(r.u, c.g, I);
        float p = r.p * c.sc.x + c.sc.y;
        v[0] = p;
#pragma unroll
        for (int ch = 0; ch < C; ch++) v[1 + ch] = I[ch] * p;
    }
};
template <int C>
struct ABDst {
    __device__ __forceinline__ bool active(int, int, int) const { return true; }
    StatsSplit sp;
    float* ab;
    int H, W;
    static constexpr int AS = ABStride<C>::value;
    struct Col { const float* st[C / 3]; float* ab; };  // statistics of every word at this column (slot and shift resolved)
    struct Raw { float4 s[C == 3 ? 2 : 4]; };  // per word: {mean_0..2, den_0}, {den_1, den_2, -, -}
    __device__ __forceinline__ Col col(int x, int k) const
    {
        Col c;
#pragma unroll
        for (int w = 0; w < C / 3; w++) {
            const float* base = sp.half[w];
            int slot = sp.per_slice ? k : 0, xs = x;
            if (sp.rep) {
                if (w != sp.shifted) {
                    slot = sp.rep[k];
                } else if (sp.interior(x, k)) {
                    base = sp.unshifted;
                    slot = sp.rep[k];
                    xs = x + sp.sgn * (sp.minD + k);
                }
            }
            c.st[w] = base + ((size_t)slot * H * W + xs) * SS8;
        }
        c.ab = ab + ((size_t)k * H * W + x) * AS;
        return c;
    }
    __device__ __forceinline__ Raw fetch(int y, const Col& c) const
    {
        Raw r;
        const size_t row = (size_t)y * W * SS8;
#pragma unroll
        for (int w = 0; w < C / 3; w++) {
            const float4* p = reinterpret_cast<const float4*>(c.st[w] + row);
            r.s[2 * w] = p[0];
            r.s[2 * w + 1] = p[1];
        }
        return r;
    }
    __device__ __forceinline__ void emit(int y, const Col& c, const Raw& r, const float (&m)[C + 1]) const
    {
        float mean[C], den[C];
#pragma unroll
        for (int w = 0; w < C / 3; w++) {
            mean[3 * w] = r.s[2 * w].x; mean[3 * w + 1] = r.s[2 * w].y; mean[3 * w + 2] = r.s[2 * w].z;
            den[3 * w] = r.s[2 * w].w; den[3 * w + 1] = r.s[2 * w + 1].x; den[3 * w + 2] = r.s[2 * w + 1].y;
        }
        const float meanP = m[0];
        float dot = 0.0f;
        float o[AS];
#pragma unroll
        for (int i = 0; i < AS; i++) o[i] = 0.0f;
#pragma unroll
        for (int ch = 0; ch < C; ch++) {
            float mI = mean[ch];
            float mp = mI * meanP;
            float cov = m[1 + ch] - mp;
            float ac = cov / den[ch];
            o[ch] = ac;
            float pr = ac * mI;
            dot = (ch == 0) ? pr : dot + pr;  // operator*(Vec,Vec): left to right (M.cpp:22-31)
        }
        o[C] = meanP - dot;
        float4* dstp = reinterpret_cast<float4*>(c.ab + (size_t)y * W * AS);
#pragma unroll
        for (int i = 0; i < AS / 4; i++) dstp[i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    }
};

// box(a_c), box(b) -> q = sum_c box(a_c)*I_c + box(b)                           (M.cpp:2849-2852)
template <int C>
struct QSrc {
    static const char* band_env() { return "ASW_BAND_Q"; }
    const float* ab;
    int H, W;
    static constexpr int AS = ABStride<C>::value;
    struct Col { const float* ab; };
    struct Raw { float4 v[AS / 4]; };
    __device__ __forceinline__ Col col(int x, int k) const { return Col{ab + ((size_t)k * H * W + x) * AS}; }
    __device__ __forceinline__ Raw fetch(int y, const Col& c) const
    {
        Raw r;
        const float4* p = reinterpret_cast<const float4*>(c.ab + (size_t)y * W * AS);
#pragma unroll
        for (int i = 0; i < AS / 4; i++) r.v[i] = p[i];
        return r;
    }
    static constexpr int KEEP = AS;
    __device__ __forceinline__ void keep(const Raw& r, uint32_t (&w)[KEEP]) const
    {
#pragma unroll
        for (int i = 0; i < AS / 4; i++) {
            w[4 * i] = __float_as_uint(r.v[i].x); w[4 * i + 1] = __float_as_uint(r.v[i].y);
            w[4 * i + 2] = __float_as_uint(r.v[i].z); w[4 * i + 3] = __float_as_uint(r.v[i].w);
        }
    }
    typedef NoRaw LRaw;
    __device__ __forceinline__ LRaw leave_fetch(int, const Col&) const { return LRaw(); }
    __device__ __forceinline__ Raw leave(const uint32_t (&w)[KEEP], const LRaw&) const
    {
        Raw r;
#pragma unroll
        for (int i = 0; i < AS / 4; i++)
            r.v[i] = make_float4(__uint_as_float(w[4 * i]), __uint_as_float(w[4 * i + 1]), __uint_as_float(w[4 * i + 2]), __uint_as_float(w[4 * i + 3]));
        return r;
    }
    __device__ __forceinline__ void eval(const Raw& r, const Col&, float (&v)[C + 1]) const
    {
        float t[AS];
#pragma unroll
        for (int i = 0; i < AS / 4; i++) { t[4 * i] = r.v[i].x; t[4 * i + 1] = r.v[i].y; t[4 * i + 2] = r.v[i].z; t[4 * i + 3] = r.v[i].w; }
#pragma unroll
        for (int ch = 0; ch < C + 1; ch++) v[ch] = t[ch];
    }
};
template <int C, bool SHIFT>
struct QDst {
    __device__ __forceinline__ bool active(int, int, int) const { return true; }
    GuideAccT<SHIFT> g;
    float* q;  // [n][H][W]
    int H, W;
    struct Col { typename GuideAccT<SHIFT>::Col g; float* q; };
    struct Raw { uint32_t u[C / 3]; };
    __device__ __forceinline__ Col col(int x, int k) const { return Col{g.col(x, k), q + (size_t)k * H * W + x}; }
    __device__ __forceinline__ Raw fetch(int y, const Col& c) const
    {
        Raw r;
        g.template fetch<C / 3>(y, c.g, r.u);
        return r;
    }
    __device__ __forceinline__ void emit(int y, const Col& c, const Raw& r, const float (&m)[C + 1]) const
    {
        float I[C];
        g.template eval<C / 3>(r.u, c.g, I);
        float dot = 0.0f;
#pragma unroll
        for (int ch = 0; ch < C; ch++) {
            float pr = m[ch] * I[ch];
            dot = (ch == 0) ? pr : dot + pr;
        }
        c.q[(size_t)y * W] = dot + m[C];
    }
};

// getCostSAD_d (M.cpp:2442-2503): |grayL - grayR shifted| as f32, box mean
struct SadSrc {
    static const char* band_env() { return "ASW_BAND_SAD"; }
    const uint8_t* gl;
    const uint8_t* gr;
    int W, minD, disp_type;
    struct Col { const uint8_t* a; const uint8_t* b; };
    struct Raw { int a, b; };
    __device__ __forceinline__ Col col(int x, int k) const
    {
        const int d = minD + k;
        if (disp_type == ASW_DISPARITY_LEFT) return Col{gl + x, gr + reflect_idx(x - d, W)};
        return Col{gl + reflect_idx(x + d, W), gr + x};
    }
    __device__ __forceinline__ Raw fetch(int y, const Col& c) const
    {
        const size_t row = (size_t)y * W;
        return Raw{(int)c.a[row], (int)c.b[row]};
    }
    static constexpr int KEEP = 1;
    __device__ __forceinline__ void keep(const Raw& r, uint32_t (&w)[1]) const { w[0] = (uint32_t)r.a | ((uint32_t)r.b << 8); }
    typedef NoRaw LRaw;
    __device__ __forceinline__ LRaw leave_fetch(int, const Col&) const { return LRaw(); }
    __device__ __forceinline__ Raw leave(const uint32_t (&w)[1], const LRaw&) const { return Raw{(int)(w[0] & 0xffu), (int)(w[0] >> 8)}; }
    __device__ __forceinline__ void eval(const Raw& r, const Col&, float (&v)[1]) const { v[0] = (float)abs(r.a - r.b); }
};
// plain 8U plane as f32 (boxFilter(8U -> CV_32F) of getInputImgNCC, M.cpp:785-786)
struct U8Src {
    static const char* band_env() { return "ASW_BAND_U8"; }
    const uint8_t* img;
    int W;
    struct Col { const uint8_t* p; };
    struct Raw { int v; };
    __device__ __forceinline__ Col col(int x, int) const { return Col{img + x}; }
    __device__ __forceinline__ Raw fetch(int y, const Col& c) const { return Raw{(int)c.p[(size_t)y * W]}; }
    static constexpr int KEEP = 1;
    __device__ __forceinline__ void keep(const Raw& r, uint32_t (&w)[1]) const { w[0] = (uint32_t)r.v; }
    typedef NoRaw LRaw;
    __device__ __forceinline__ LRaw leave_fetch(int, const Col&) const { return LRaw(); }
    __device__ __forceinline__ Raw leave(const uint32_t (&w)[1], const LRaw&) const { return Raw{(int)w[0]}; }
    __device__ __forceinline__ void eval(const Raw& r, const Col&, float (&v)[1]) const { v[0] = (float)r.v; }
};
struct PlaneDst {
    __device__ __forceinline__ bool active(int, int, int) const { return true; }
    float* out;
    int H, W;
    typedef NoRaw Raw;
    struct Col { float* o; };
    __device__ __forceinline__ Col col(int x, int k) const { return Col{out + (size_t)k * H * W + x}; }
    __device__ __forceinline__ Raw fetch(int, const Col&) const { return Raw(); }
    __device__ __forceinline__ void emit(int y, const Col& c, const Raw&, const float (&m)[1]) const { c.o[(size_t)y * W] = m[0]; }
};
// band: rows per band (0 = default); wg_strips: 1 = the four wavefronts of a workgroup take four neighbouring strips of one slice
// (they share halo columns in L1), 0 = four slices of one strip (they share the slice-independent operands), -1 = default
struct WalkOpts { int band = 0; int wg_strips = -1; };

template <int NP, int CPL, int ND, int WPE = 4, bool NANSAFE = false, int RING = 0, int PF = 1, class Src, class Dst>
int launch_walk_t(hipStream_t s, const Src& src, const Dst& dst, int H, int W, int k, int n, int n_active = -1, WalkOpts o = WalkOpts())
{
    constexpr int SW = 64 * CPL;
    if (k < 1 || k > SW / 2) return ASW_ERR_BAD_ARGUMENT;  // k-1 halo columns must leave outputs in the strip
    const int XO = SW - (k - 1);
    const int nxw = (W + XO - 1) / XO;
    const int nzs = (n + ND - 1) / ND;
    const bool ring = RING && k == 15;
    // rows per band.  Without the register ring the leaving row is fetched a second time, and with the XCD-aware order short
    // bands keep that re-read in L2 although every band pays k-1 warm-up rows (1080p D=128 GuidedF_2, a/b + q pass: 16 rows
    // 5.53 ms, 20..32 rows 5.43-5.50, 48 rows 5.79, 64 rows 6.09, 128 rows 6.24, 270 rows 8.15); the single-channel launches
    // (SAD cost, BLO1) are best at 64.  With the ring nothing is re-read: as few bands as still give the chip ~4 rounds of
    // wavefronts (4096 resident ones).
    int band = NP >= 4 ? 32 : 64;
    if (ring) {
        const long long per_band_row = (long long)nxw * nzs;
        const int nb = (int)std::max(1LL, std::min((long long)H, (16384 + per_band_row - 1) / per_band_row));
        band = (H + nb - 1) / nb;
    }
    if (o.band >= 2) band = o.band;  // measurement hook (AswTuning)
    if (band < 2 * k) band = 2 * k;  // keep the warm-up overhead (k-1 rows per band) below ~50 %
    // A launch with too few wavefronts to fill the chip (guide statistics of one slice: 578 at 1080p, 72 at 640x360) is bound by
    // the latency of its serial row walk, not by throughput: shorter bands mean shorter walks and more wavefronts, and the
    // redundant warm-up rows cost nothing there.
    // (n_active: slices that do not leave at once -- consumers that skip slices or strips say how many they expect)
    const int n_eff = n_active > 0 && n_active < n ? n_active : n;
    while (band > 2 && (long long)nxw * ((H + band - 1) / band) * ((n_eff + ND - 1) / ND) < 4096) band /= 2;
    size_t lds = (size_t)4 * ND * NP * (SW + 2) * sizeof(double);
    // register target: at least 4 waves/SIMD; asking for 6 or 8 makes the allocator serialise/spill (7.1 / 12.6 ms vs 6.2)
    auto kern = ring ? k_box_walk<NP, CPL, ND, WPE, NANSAFE, 15, RING, PF, Src, Dst>
              : k == 15 ? k_box_walk<NP, CPL, ND, (RING ? 4 : WPE), NANSAFE, 15, 0, 1, Src, Dst>
                        : k_box_walk<NP, CPL, ND, (RING ? 4 : WPE), NANSAFE, 0, 0, 1, Src, Dst>;
    // four slices of one strip per workgroup when there are enough slices (1080p D=128: GuidedF 24.1 -> 22.9 ms, BLO1 -7 %,
    // GuidedF_2 -1 %); four strips of the one slice otherwise
    int slice_par = nzs >= 4 ? 1 : 0;
    if (o.wg_strips == 1) slice_par = 0;
    const int spw = slice_par ? 4 : 1;
    const int ngx = slice_par ? nxw : (nxw + 3) / 4, nby = (H + band - 1) / band, nzg = (nzs + spw - 1) / spw;
    const long long nwg = (long long)((ngx * nby + 7) / 8) * 8 * nzg;  // regions rounded up to a multiple of the 8 XCDs
    if (nwg > 0x7fffffffLL) return ASW_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(BW), lds, s, src, dst, H, W, k, band, nxw, n, ngx, nby, slice_par);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

template <int NP, int ND = 1, bool NANSAFE = false, class Src, class Dst>
int launch_walk(hipStream_t s, const Src& src, const Dst& dst, int H, int W, int k, int n, int n_active = -1, WalkOpts o = WalkOpts())
{
    // Measured on MI355X (1080p D=128): <CPL,prefetch> = <1,0> 6.83 ms, <1,1> 7.11, <2,0> 6.31, <2,1> 6.40 for the NP=4
    // pair of launches; 24.3 / 24.9 / 24.3 / 24.1 ms for NP=7.  The kernels are bound by the memory system
    // (L2/MALL re-reads of statistics and a/b planes), not by issue: prefetching buys nothing, two adjacent
    // columns per lane save the shared horizontal sum, ND slices per wavefront share the slice-independent loads.
    // ND > 1 (several slices per wavefront sharing guide pixel and statistics) was measured and rejected: <CPL,ND> =
    // <2,1> 6.2 ms, <2,2> 7.4, <1,2> 7.5, <1,4> 9.7 -- the extra registers cost more occupancy than the traffic saves.
    return launch_walk_t<NP, 2, ND, 4, NANSAFE>(s, src, dst, H, W, k, n, n_active, o);
}

// the two passes of the 3-channel guided filter (a/b, q) in the forms AswTuning selects
template <bool SHIFT>
int launch_ab_q3(hipStream_t s, const GuidedLaunch& a, const GuideAccT<SHIFT>& g, const StatsSplit& sp)
{
    const AswTuning& t = *a.tune;
    int rc;
    ABDst<3> dst{sp, a.ab, a.H, a.W};
    WalkOpts oab; oab.band = t.band_ab;
    WalkOpts oq; oq.band = t.band_q; oq.wg_strips = t.q_wg_strips;
    const int ring_ab = a.nan_safe ? 0 : t.ring_ab, ring_q = a.nan_safe ? 0 : t.ring_q;
    if (ring_ab == 1) {  // ring = {cost, guide word}: 2 wavefronts per SIMD
        ABSrc<3, SHIFT, true> src{g, a.P, a.pscales, a.H, a.W};
        rc = launch_walk_t<4, 2, 1, 2, false, 1, 1>(s, src, dst, a.H, a.W, a.r, a.n, -1, oab);
    } else if (ring_ab == 2) {  // ring = {cost}, the leaving row's guide word fetched again: 3 wavefronts per SIMD
        ABSrc<3, SHIFT, false> src{g, a.P, a.pscales, a.H, a.W};
        rc = launch_walk_t<4, 2, 1, 3, false, 1, 1>(s, src, dst, a.H, a.W, a.r, a.n, -1, oab);
    } else {
        ABSrc<3, SHIFT, true> src{g, a.P, a.pscales, a.H, a.W};
        rc = a.nan_safe ? launch_walk<4, 1, true>(s, src, dst, a.H, a.W, a.r, a.n, -1, oab) : launch_walk<4>(s, src, dst, a.H, a.W, a.r, a.n, -1, oab);
    }
    if (rc != ASW_OK) return rc;
    QSrc<3> qs{a.ab, a.H, a.W};
    QDst<3, SHIFT> qd{g, a.q, a.H, a.W};
    if (ring_q == 1) return launch_walk_t<4, 1, 1, 4, false, 1, 1>(s, qs, qd, a.H, a.W, a.r, a.n, -1, oq);  // one column per lane
    if (ring_q == 2) return launch_walk_t<4, 2, 1, 2, false, 1, 1>(s, qs, qd, a.H, a.W, a.r, a.n, -1, oq);  // two: 2 wavefronts per SIMD
    return a.nan_safe ? launch_walk<4, 1, true>(s, qs, qd, a.H, a.W, a.r, a.n, -1, oq) : launch_walk<4>(s, qs, qd, a.H, a.W, a.r, a.n, -1, oq);
}

}  // namespace

int launch_cost_sad(hipStream_t s, const uint8_t* gl, const uint8_t* gr, int H, int W, int disp_type, int win, int minD,
                    int numD, float* cost)
{
    SadSrc src{gl, gr, W, minD, disp_type};
    PlaneDst dst{cost, H, W};
    return launch_walk<1>(s, src, dst, H, W, win, numD);
}

int launch_box_mean_u8(hipStream_t s, const uint8_t* img, int H, int W, int win, float* mean)
{
    U8Src src{img, W};
    PlaneDst dst{mean, H, W};
    return launch_walk<1>(s, src, dst, H, W, win, 1);
}

int launch_guided(hipStream_t s, const GuidedLaunch& a)
{
    const int nstat = a.guide_per_slice ? a.n : 1;
    const float epsf = (float)a.eps;
    const bool shifted = a.shiftA != 0 || a.shiftB != 0 || a.guide_per_slice;
    const size_t half_floats = (size_t)nstat * a.H * a.W * SS8;
    StatsSplit sp;
    sp.half[0] = a.stats; sp.half[1] = a.stats + half_floats; sp.unshifted = a.stats + 2 * half_floats;
    sp.rep = nullptr; sp.per_slice = a.guide_per_slice ? 1 : 0; sp.shifted = -1; sp.sgn = 0;
    sp.lo = a.r / 2; sp.hi = a.r - 1 - a.r / 2; sp.W = a.W; sp.minD = a.minD;
    int rc;
    if (a.C == 3 && !shifted) {
        // GuidedF_2 / 3-channel getGuidedFilter: the guide does not depend on the slice.
        // 1. guide statistics, once   2. a, b   3. q
        GuideAccT<false> g{a.guideA, a.guideB, a.gscales, 0, a.W, 0, 0, a.minD};
        StatsSrc<3, 0, false> ss{g};
        StatsDst<0> sd{sp.half[0], sp, a.H, a.W, epsf};
        rc = launch_walk<6>(s, ss, sd, a.H, a.W, a.r, 1);
        if (rc != ASW_OK) return rc;
        return launch_ab_q3<false>(s, a, g, sp);
    }
    GuideAccT<true> g{a.guideA, a.guideB, a.gscales, a.guide_per_slice ? 1 : 0, a.W, a.shiftA, a.shiftB, a.minD};
    if (a.C == 3) {
        StatsSrc<3, 0, true> ss{g};
        StatsDst<0> sd{sp.half[0], sp, a.H, a.W, epsf};
        rc = launch_walk<6>(s, ss, sd, a.H, a.W, a.r, nstat);
        if (rc != ASW_OK) return rc;
        return launch_ab_q3<true>(s, a, g, sp);
    }
    // 6-channel guide
    const bool share = a.guide_per_slice && a.rep_scratch && ((a.shiftA != 0) != (a.shiftB != 0));
    StatsSrc<6, 0, true> s0{g};
    StatsSrc<6, 1, true> s1{g};
    if (!share) {  // public getGuidedFilter (one slice) or both / no words shifted: plain per-slot statistics
        StatsDst<0> d0{sp.half[0], sp, a.H, a.W, epsf};
        rc = launch_walk<6>(s, s0, d0, a.H, a.W, a.r, nstat);
        if (rc != ASW_OK) return rc;
        StatsDst<0> d1{sp.half[1], sp, a.H, a.W, epsf};
        rc = launch_walk<6>(s, s1, d1, a.H, a.W, a.r, nstat);
        if (rc != ASW_OK) return rc;
    } else {
        // GuidedF / GuidedF_3: [fixed image, other image shifted by d].  See StatsSplit.
        hipLaunchKernelGGL(k_scale_groups, dim3((a.n + 255) / 256), dim3(256), 0, s, a.gscales, a.n, a.rep_scratch);
        sp.rep = a.rep_scratch;
        sp.shifted = a.shiftA != 0 ? 0 : 1;
        sp.sgn = a.shiftA != 0 ? a.shiftA : a.shiftB;
        GuideAccT<true> gu = g;  // the same guide without the shift
        gu.shiftA = 0; gu.shiftB = 0;
        StatsSrc<6, 0, true> u0{gu};
        StatsSrc<6, 1, true> u1{gu};
        // fixed word: representatives only
        StatsDst<1> df{sp.half[1 - sp.shifted], sp, a.H, a.W, epsf};
        const int n_rep = 4;  // scale groups are few (natural images: the global extrema are visible at nearly every d)
        rc = sp.shifted == 1 ? launch_walk<6>(s, s0, df, a.H, a.W, a.r, a.n, n_rep) : launch_walk<6>(s, s1, df, a.H, a.W, a.r, a.n, n_rep);
        if (rc != ASW_OK) return rc;
        // shifted word's image without the shift: representatives only
        StatsDst<1> du{sp.unshifted, sp, a.H, a.W, epsf};
        rc = sp.shifted == 1 ? launch_walk<6>(s, u1, du, a.H, a.W, a.r, a.n, n_rep) : launch_walk<6>(s, u0, du, a.H, a.W, a.r, a.n, n_rep);
        if (rc != ASW_OK) return rc;
        // shifted word as the guide shows it: border strips of every slice
        StatsDst<2> db{sp.half[sp.shifted], sp, a.H, a.W, epsf};
        const int nstrips = (a.W + (128 - a.r)) / (129 - a.r);  // strips of a two-column-per-lane walk
        const int n_border = (int)(((long long)a.n * 3 + nstrips - 1) / nstrips);  // ~3 strips of a slice touch its border columns
        rc = sp.shifted == 1 ? launch_walk<6>(s, s1, db, a.H, a.W, a.r, a.n, n_border) : launch_walk<6>(s, s0, db, a.H, a.W, a.r, a.n, n_border);
        if (rc != ASW_OK) return rc;
    }
    ABSrc<6, true> src{g, a.P, a.pscales, a.H, a.W};
    ABDst<6> dst{sp, a.ab, a.H, a.W};
    // one column per lane for the 7-plane a/b pass (6.2 ms): two columns need 128 VGPRs + 33 spilled (9.3 ms); two columns at a
    // 3-waves-per-SIMD register target (148 VGPRs, no spills) take the same time as one column (GuidedF 12.67 vs 12.60 ms)
    // (boxes wider than 32 do not leave outputs in a 64-column strip: those take the two-column form at 148 VGPRs)
    if (a.r > 32)
        rc = a.nan_safe ? launch_walk_t<7, 2, 1, 3, true>(s, src, dst, a.H, a.W, a.r, a.n) : launch_walk_t<7, 2, 1, 3>(s, src, dst, a.H, a.W, a.r, a.n);
    else
        rc = a.nan_safe ? launch_walk_t<7, 1, 1, 4, true>(s, src, dst, a.H, a.W, a.r, a.n) : launch_walk_t<7, 1, 1>(s, src, dst, a.H, a.W, a.r, a.n);
    if (rc != ASW_OK) return rc;
    QSrc<6> qs{a.ab, a.H, a.W};
    QDst<6, true> qd{g, a.q, a.H, a.W};
    return a.nan_safe ? launch_walk<7, 1, true>(s, qs, qd, a.H, a.W, a.r, a.n) : launch_walk<7>(s, qs, qd, a.H, a.W, a.r, a.n);  // two columns per lane: 4.5 ms, one: 5.3 ms
}

// interleaved C-channel 8U image -> BGRX word planes (channels 3w..3w+2 in plane w)
namespace {
__global__ __launch_bounds__(256) void k_pack_words(const uint8_t* __restrict__ img, size_t n, int C, int w, uint32_t* __restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = img + i * C + 3 * w;
    out[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}
}  // namespace

int launch_pack_words(hipStream_t s, const uint8_t* img, int H, int W, int C, int w, uint32_t* out)
{
    size_t n = (size_t)H * W;
    hipLaunchKernelGGL(k_pack_words, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, n, C, w, out);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

// one [slot][H][W][8] array for a 3-channel guide; three for a 6-channel guide (word A, word B, unshifted word: StatsSplit)
size_t guided_stats_floats(int C, int nstat, int H, int W) { return (size_t)nstat * H * W * 8 * (C == 3 ? 1 : 3); }
size_t guided_ab_floats(int C, int n, int H, int W) { return (size_t)n * H * W * (C == 3 ? 4 : 8); }
