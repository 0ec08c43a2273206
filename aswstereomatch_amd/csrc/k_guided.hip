// Guided-filter aggregation (getGuidedFilter, M.cpp:2766-2854) and the box-mean machinery it is made of.
//
// boxFilter(CV_32F, Size(k,k), normalised, BORDER_REFLECT_101) == f64 window sum * 1/(k*k) -> f32
// (SURVEY App. A-9).  One generic "column walk" kernel (k_box_walk) evaluates NP box means at once:
//   * a wavefront owns a strip of 128 input columns (two adjacent ones per lane; 128-(k-1) output columns) of a band of rows
//     and walks down the band; every lane keeps NP vertical running sums in f64 (add the entering row, subtract the leaving
//     one -- the sliding form of OpenCV's ColumnSum);
//   * the leaving row comes from a register ring of the last 15 rows' operands (RING: the 3-channel guided filter at 15x15)
//     or is fetched again (every other launch);
//   * per output row the vertical sums go through a wave-private LDS strip (no workgroup barrier, wavefront-scope fences
//     for the compiler) and every lane adds the neighbours of its two output columns;
//   * the producer (Src) and consumer (Dst) are functors, so normalisation, products, covariance, a = cov/(var+eps), b and q
//     are fused into the filters that need them and never hit HBM as separate planes.
// Layouts, variants and what bounds the launches: DESIGN.md 4.2 (rounds 1-2) and 4.2c (round 3).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "asw_device.h"
#include "asw_internal.h"

namespace {


constexpr int BW = 256;  // threads per block = 4 independent wavefronts
struct NoRaw {};
template <class D, bool P> struct DstRawSel { using type = typename D::Raw; };
template <class D> struct DstRawSel<D, true> { using type = typename D::Raw2; };

// Every wavefront owns a strip of 128 input columns (two adjacent ones per lane, 128-(k-1) output columns)
// of a band of rows and walks down the band on its own:
//   * vertical running sums in f64 registers (add the entering row, subtract the leaving one -- ColumnSum's
//     sliding form; the leaving row is re-fetched from cache or kept in a REGISTER ring (RING, below): an LDS ring cost
//     61 KB per workgroup and 2x the run time);
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
    static_assert(PF >= 0 && PF <= 3, "depth of the software pipeline (0: the loads of a step are issued at its own top)");
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
    const int spw = (slice_par & 1) ? 4 : 1;
    const int nzg = ((nslices + ND - 1) / ND + spw - 1) / spw;
    const int wj = blockIdx.x >> 3;
    // an XCD takes a contiguous run of regions: the regions it has in flight at any time are neighbouring strips of one band,
    // whose halo columns (and the 128-byte lines two strips share) are then fetched from HBM once
    const int rpx = (ngx * nby + 7) >> 3;
    const int reg = (slice_par & 2) ? (wj / nzg) * 8 + (blockIdx.x & 7) : (blockIdx.x & 7) * rpx + wj / nzg;
    if (wj / nzg >= rpx || reg >= ngx * nby) return;
    const int gx = reg % ngx, by = reg / ngx;
    const int xw = (slice_par & 1) ? gx : gx * 4 + wv;              // wavefront's strip index
    const int zg = (slice_par & 1) ? (wj % nzg) * 4 + wv : wj % nzg;  // wavefront's slice group
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
    // DPAIR: the consumer fetches the operands of the lane's two adjacent output columns with ONE set of vector loads and stores
    // both results with one store per plane (planar layouts: every memory instruction of a wavefront covers one dense run).
    // The L1 handles one 128-byte line per cycle: with per-pixel records (32 B of statistics, 16 B of a/b) and two columns per
    // lane an instruction touched 16-32 lines, and the a/b pass spent a third of its time in the L1 tag pipeline.
    constexpr bool DPAIR = CPL == 2 && Dst::PAIR;
    using DRawT = typename DstRawSel<Dst, DPAIR>::type;
    DRawT fD[NPH][DPAIR ? 1 : CPL][ND];
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
                if constexpr (DPAIR) {
                    if (c == 0) fD[SL][0][n] = dst.fetch2(yd, dcol[0][n]);
                } else {
                    fD[SL][c][n] = dst.fetch(yd, dcol[c][n]);
                }
            }
    };
    if constexpr (PF >= 1) issue(0, std::integral_constant<int, 0>());
    if constexpr (PF >= 2) issue(1, std::integral_constant<int, 1>());
    if constexpr (PF >= 3) issue(2, std::integral_constant<int, 2>());

    auto step = [&](int s, auto ph_c, auto sub_c, auto out_c) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr bool SUB = decltype(sub_c)::value, OUT = decltype(out_c)::value;
        issue(s + PF, std::integral_constant<int, (PH + PF) % NPH>());  // PF = 0: this step's own loads
        typename Src::Raw (&rn)[CPL][ND] = fN[PH];
        DRawT (&rd)[DPAIR ? 1 : CPL][ND] = fD[PH];
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
        // PAIRS (two adjacent columns per lane, compile-time odd window, finite data): the strip holds {V[2l] + V[2l+1], V[2l]} per
        // lane instead of {V[2l], V[2l+1]}; with T = the six pair sums of lanes l+1 .. l+6 the two windows are
        // (pair[l] + T) + V[2(l+7)] and (V[2l+1] + T) + pair[l+7]: 10 f64 additions for the two columns instead of 17, seven
        // ds_read_b128 instead of eight.  Another association of the same f64 sum (the oracle's: ascending; OpenCV's: sliding).
        constexpr bool PAIRS = CPL == 2 && KT > 0 && (KT & 1) && !NANSAFE;
        if constexpr (OUT) {
#pragma unroll
            for (int n = 0; n < ND; n++)
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    if constexpr (PAIRS) {
                        hs[(n * NP + p) * (SW + 2) + c0] = vs[0][n][p] + vs[1][n][p];
                        hs[(n * NP + p) * (SW + 2) + c0 + 1] = vs[0][n][p];
                    } else {
#pragma unroll
                        for (int c = 0; c < CPL; c++) hs[(n * NP + p) * (SW + 2) + c0 + c] = vs[c][n][p];
                    }
                }
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
                    if constexpr (PAIRS && (RING || WPE <= 3)) {  // (register targets of 4 wavefronts per SIMD have no room for it)
                        // All LDS reads of a group of planes are issued before the first addition (the scheduler, left alone,
                        // orders them plane by plane to save registers: ~16 exposed LDS round trips per step, and at two
                        // wavefronts per SIMD nobody hides them -- one wavefront alone spent two thirds of a step waiting).
                        constexpr int HP = (KT - 1) / 2;
                        constexpr int GRP = 2;  // planes whose reads are in flight together (16 registers each; all four: spills)
#pragma unroll
                        for (int p0 = 0; p0 < NP; p0 += GRP) {
                            double bb[GRP][2 * HP + 2];
#pragma unroll
                            for (int g = 0; g < GRP; g++)
                                if (p0 + g < NP) {
                                    const double* b = hs + (n * NP + p0 + g) * (SW + 2) + c0;
#pragma unroll
                                    for (int i = 2; i < 2 * HP + 2; i++)
                                        if (!(i & 1) || i == 2 * HP + 1) bb[g][i] = b[i];  // pair sums of lanes l+1 .. l+HP, V[even] of l+HP
                                }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int g = 0; g < GRP; g++)
                                if (p0 + g < NP) {
                                    const int p = p0 + g;
                                    double t = bb[g][2];
#pragma unroll
                                    for (int i = 2; i < HP; i++) t = t + bb[g][2 * i];
                                    const double s0 = ((vs[0][n][p] + vs[1][n][p]) + t) + bb[g][2 * HP + 1];
                                    const double s1 = (vs[1][n][p] + t) + bb[g][2 * HP];
                                    m[0][p] = (float)(s0 * scale);
                                    m[1][p] = (float)(s1 * scale);
                                }
                        }
                    } else
#pragma unroll
                    for (int p = 0; p < NP; p++) {
                        const double* b = hs + (n * NP + p) * (SW + 2) + c0;
                        if constexpr (PAIRS) {
                            constexpr int HP = (KT - 1) / 2;  // whole pairs in a window
                            double bb[2 * HP + 2];
#pragma unroll
                            for (int i = 2; i < 2 * HP + 2; i++) bb[i] = b[i];  // lanes l+1 .. l+HP: {pair, V[even]}
                            double t = bb[2];
#pragma unroll
                            for (int i = 2; i < HP; i++) t = t + bb[2 * i];
                            const double s0 = ((vs[0][n][p] + vs[1][n][p]) + t) + bb[2 * HP + 1];
                            const double s1 = (vs[1][n][p] + t) + bb[2 * HP];
                            m[0][p] = (float)(s0 * scale);
                            m[1][p] = (float)(s1 * scale);
                            continue;
                        }
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
                        if constexpr (DPAIR) {
                            dst.emit2(y, dcol[0][n], rd[0][n], m[0], m[1], out_col[1]);
                        } else {
                            dst.emit(y, dcol[0][n], rd[0][n], m[0]);
                            if constexpr (CPL > 1) {
                                if (out_col[CPL - 1]) dst.emit(y, dcol[CPL - 1][n], rd[CPL - 1][n], m[CPL - 1]);
                            }
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
    // The FIFO slot of a step is a compile-time phase, and the loops are unrolled by the number of slots: a run-time dispatch on
    // the phase inside the loop (if (ph == 0) ... else ...) makes the compiler copy the FIFO registers at the join, and those
    // copies WAIT for the loads issued at the top of the very same step -- the pipeline then hides one step's arithmetic, not
    // its depth in steps.
    auto run = [&](int s_end, auto sub_c, auto out_c) {
        // seams as straight-line code (a loop around a run-time phase dispatch costs hundreds of spilled registers):
        // up to NPH-1 steps until the phase is 0, the unrolled loop, up to NPH-1 steps behind it
        if constexpr (NPH > 1) { if (ph == 1 && s < s_end) { step(s, std::integral_constant<int, 1 % NPH>(), sub_c, out_c); s++; ph = 2 % NPH; } }
        if constexpr (NPH > 2) { if (ph == 2 && s < s_end) { step(s, std::integral_constant<int, 2 % NPH>(), sub_c, out_c); s++; ph = 3 % NPH; } }
        if constexpr (NPH > 3) { if (ph == 3 && s < s_end) { step(s, std::integral_constant<int, 3 % NPH>(), sub_c, out_c); s++; ph = 0; } }
        if (ph == 0) {
            for (; s + NPH <= s_end; s += NPH) {  // NPH steps, every FIFO slot a fixed set of registers
                step(s, std::integral_constant<int, 0>(), sub_c, out_c);
                if constexpr (NPH > 1) step(s + 1, std::integral_constant<int, 1 % NPH>(), sub_c, out_c);
                if constexpr (NPH > 2) step(s + 2, std::integral_constant<int, 2 % NPH>(), sub_c, out_c);
                if constexpr (NPH > 3) step(s + 3, std::integral_constant<int, 3 % NPH>(), sub_c, out_c);
            }
            if constexpr (NPH > 1) { if (s < s_end) { step(s, std::integral_constant<int, 0>(), sub_c, out_c); s++; ph = 1; } }
            if constexpr (NPH > 2) { if (s < s_end) { step(s, std::integral_constant<int, 1 % NPH>(), sub_c, out_c); s++; ph = 2; } }
            if constexpr (NPH > 3) { if (s < s_end) { step(s, std::integral_constant<int, 2 % NPH>(), sub_c, out_c); s++; ph = 3; } }
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
    static constexpr bool PAIR = false;
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

// a/b planes: C = 6: two float4 volumes [half][k][y][x] = {a_0..a_3}, {a_4, a_5, b, pad}
template <int C> struct ABStride { static constexpr int value = (C == 3) ? 4 : 8; };

// box(P), box(I_c*P) -> a_c = cov_c / den_c, b = meanP - sum_c a_c*meanI_c      (M.cpp:2780-2847)
// The register ring keeps the cost and the guide word(s) of a row: C/3 + 1 dwords per column and row.  (Keeping the cost only and
// fetching the leaving row's guide word again saves 30 registers and a wavefront per SIMD, and measured 0.3-0.4 ms slower.)
template <int C, bool SHIFT>
struct ABSrc {
    static constexpr int KEEP = C / 3 + 1;
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
#pragma unroll
        for (int i = 0; i < C / 3; i++) w[1 + i] = r.u[i];
    }
    typedef NoRaw LRaw;
    __device__ __forceinline__ LRaw leave_fetch(int, const Col&) const { return LRaw(); }
    __device__ __forceinline__ Raw leave(const uint32_t (&w)[KEEP], const LRaw&) const
    {
        Raw r;
#pragma unroll
        for (int i = 0; i < C / 3; i++) r.u[i] = w[1 + i];
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
    static constexpr bool PAIR = false;
    __device__ __forceinline__ bool active(int, int, int) const { return true; }
    StatsSplit sp;
    float* ab;
    int H, W;
    size_t hstride;  // floats between the two 16-byte halves of a pixel's record: the volume is [half][n][H][W] float4 (see k_q6_pair)
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
        c.ab = ab + ((size_t)k * H * W + x) * 4;
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
        float* dstp = c.ab + (size_t)y * W * 4;
        static_assert(C == 6, "halves {a_0, a_1, a_2, -} and {a_3, a_4, a_5, b}: one per guide word (k_ab6_pair / k_q6_pair)");
        *reinterpret_cast<float4*>(dstp) = make_float4(o[0], o[1], o[2], 0.0f);
        *reinterpret_cast<float4*>(dstp + hstride) = make_float4(o[3], o[4], o[5], o[6]);
    }
};

// box(a_c), box(b) -> q = sum_c box(a_c)*I_c + box(b)                           (M.cpp:2849-2852)
template <int C>
struct QSrc {
    const float* ab;
    int H, W;
    size_t hstride;  // see ABDst
    static constexpr int AS = ABStride<C>::value;
    struct Col { const float* ab; };
    struct Raw { float4 v[AS / 4]; };
    __device__ __forceinline__ Col col(int x, int k) const { return Col{ab + ((size_t)k * H * W + x) * 4}; }
    __device__ __forceinline__ Raw fetch(int y, const Col& c) const
    {
        Raw r;
        const float* p = c.ab + (size_t)y * W * 4;
#pragma unroll
        for (int i = 0; i < AS / 4; i++) r.v[i] = *reinterpret_cast<const float4*>(p + i * hstride);
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
        static_assert(C == 6, "see ABDst::emit");
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[4]; v[4] = t[5]; v[5] = t[6]; v[6] = t[7];
    }
};
template <int C, bool SHIFT>
struct QDst {
    static constexpr bool PAIR = false;
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

// ---- planar forms for the 3-channel guide (GuidedF_2, 3-channel getGuidedFilter) ----------------------------------------
// Statistics: [slot] x { R_0, R_1, R_2 : double [H][W] ; meanI_0..2 : float [H][W] }, R_c = 1 / (double)den_c with
// den_c = var_c + eps in f32 (M.cpp:2846).  The consumer needs den only as a divisor: a_c = cov_c / den_c =
// (float)((double)cov_c * R_c) -- three instructions instead of the eleven of an IEEE f32 division (two v_div_scale, v_rcp,
// five fma/mul, v_div_fmas, v_div_fixup) and the same bits: R_c and the product carry 2^-52 of relative error, and the quotient
// of two 24-bit numbers is never closer than 2^-49 to a rounding boundary of the 24-bit format (the classical double-precision
// argument; zero, infinite and NaN divisors behave as in the division; only a subnormal quotient that is an exact tie can round
// the other way).
// a/b: [slice] x { {a_0, a_1} ; {a_2, b} } x [strip][H][TW] float2, TW = the output columns of a strip of the a/b pass: STRIP-major,
// so the rows a wavefront produces are one contiguous stream.  In image order a wavefront stores 912 bytes, then jumps a row
// pitch (15 KB): measured with the pass's own store pattern (tools/ubench_wpattern.hip), that reaches 3.2 TB/s of HBM write
// bandwidth against 5.2-5.5 TB/s for contiguous rows -- the 4.25 GB a/b store was 1.3 ms of a 2.6 ms pass.
// With planes, the two adjacent columns of a lane are one 16-byte (or 8-byte) access per plane and a wavefront's instruction
// covers one dense run of memory.
struct ABTiles {
    int TW;              // columns per strip tile
    size_t tile;         // H * TW (float2 elements of one strip tile)
    size_t plane;        // nstrips * tile
    size_t slice_stride; // 2 * plane, in float2
    int H;
    __host__ __device__ size_t col_offset(int x) const { const int t = x / TW; return (size_t)t * tile + (x - t * TW); }
};
inline ABTiles ab_tiles(int H, int W, int r)
{
    ABTiles a;
    a.TW = 128 - (r - 1);  // output columns of a two-column-per-lane strip
    if (a.TW < 2) a.TW = 2;
    a.H = H;
    a.tile = (size_t)H * a.TW;
    a.plane = (size_t)((W + a.TW - 1) / a.TW) * a.tile;
    a.slice_stride = 2 * a.plane;
    return a;
}
struct StatsPlanes {
    double* R;   // [slot][3][H][W]
    float* M;    // [slot][3][H][W]
    size_t plane;  // H * W
};
__host__ __device__ inline StatsPlanes stats_planes(float* base, int nslot, int H, int W)
{
    StatsPlanes p;
    p.plane = (size_t)H * W;
    p.R = reinterpret_cast<double*>(base);
    p.M = reinterpret_cast<float*>(p.R + (size_t)nslot * 3 * p.plane);
    return p;
}
struct StatsDstP {
    static constexpr bool PAIR = true;
    StatsPlanes sp;
    int W;
    float epsf;
    typedef NoRaw Raw;
    typedef NoRaw Raw2;
    struct Col { double* r; float* m; };
    __device__ __forceinline__ bool active(int, int, int) const { return true; }
    __device__ __forceinline__ Col col(int x, int k) const { return Col{sp.R + (size_t)k * 3 * sp.plane + x, sp.M + (size_t)k * 3 * sp.plane + x}; }
    __device__ __forceinline__ Raw fetch(int, const Col&) const { return Raw(); }
    __device__ __forceinline__ Raw2 fetch2(int, const Col&) const { return Raw2(); }
    __device__ __forceinline__ void stat(const float (&m)[6], double (&R)[3]) const
    {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float mm = m[ch] * m[ch];
            float var = m[3 + ch] - mm;
            float den = 1.0f * epsf + var;  // scaleAdd(ones, eps, var)
            R[ch] = 1.0 / (double)den;
        }
    }
    __device__ __forceinline__ void emit(int y, const Col& c, const Raw&, const float (&m)[6]) const
    {
        double R[3];
        stat(m, R);
        const size_t row = (size_t)y * W;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) { c.r[ch * sp.plane + row] = R[ch]; c.m[ch * sp.plane + row] = m[ch]; }
    }
    __device__ __forceinline__ void emit2(int y, const Col& c, const Raw2&, const float (&m0)[6], const float (&m1)[6], bool second) const
    {
        double R0[3], R1[3];
        stat(m0, R0);
        stat(m1, R1);
        const size_t row = (size_t)y * W;
        if (second) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                *reinterpret_cast<double2*>(c.r + ch * sp.plane + row) = make_double2(R0[ch], R1[ch]);
                *reinterpret_cast<float2*>(c.m + ch * sp.plane + row) = make_float2(m0[ch], m1[ch]);
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) { c.r[ch * sp.plane + row] = R0[ch]; c.m[ch * sp.plane + row] = m0[ch]; }
        }
    }
};
// box(P), box(I_c*P) -> a_c, b from the planar statistics, to the planar a/b volume                       (M.cpp:2780-2847)
struct ABDstP {
    static constexpr bool PAIR = true;
    StatsPlanes sp;
    int per_slice;  // statistics slot = slice (per-slice guide) or 0
    float2* ab;     // strip-major tiles, see ABTiles
    ABTiles at;
    int W;
    struct Col { const double* r; const float* m; float2* ab; };
    struct Raw { double R[3]; float mean[3]; };
    struct Raw2 { double2 R[3]; float2 mean[3]; };
    __device__ __forceinline__ bool active(int, int, int) const { return true; }
    __device__ __forceinline__ Col col(int x, int k) const
    {
        const size_t slot = per_slice ? (size_t)k : 0;
        return Col{sp.R + slot * 3 * sp.plane + x, sp.M + slot * 3 * sp.plane + x, ab + (size_t)k * at.slice_stride + at.col_offset(x)};
    }
    __device__ __forceinline__ Raw fetch(int y, const Col& c) const
    {
        Raw r;
        const size_t row = (size_t)y * W;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) { r.R[ch] = c.r[ch * sp.plane + row]; r.mean[ch] = c.m[ch * sp.plane + row]; }
        return r;
    }
    __device__ __forceinline__ Raw2 fetch2(int y, const Col& c) const
    {
        Raw2 r;
        const size_t row = (size_t)y * W;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            r.R[ch] = *reinterpret_cast<const double2*>(c.r + ch * sp.plane + row);
            r.mean[ch] = *reinterpret_cast<const float2*>(c.m + ch * sp.plane + row);
        }
        return r;
    }
    __device__ __forceinline__ void ab_of(const double (&R)[3], const float (&mean)[3], const float (&m)[4], float (&o)[4]) const
    {
        const float meanP = m[0];
        float dot = 0.0f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float mI = mean[ch];
            float mp = mI * meanP;
            float cov = m[1 + ch] - mp;
            float ac = (float)((double)cov * R[ch]);  // == cov / den_c, see above
            o[ch] = ac;
            float pr = ac * mI;
            dot = (ch == 0) ? pr : dot + pr;  // operator*(Vec,Vec): left to right (M.cpp:22-31)
        }
        o[3] = meanP - dot;
    }
    __device__ __forceinline__ void emit(int y, const Col& c, const Raw& r, const float (&m)[4]) const
    {
        float o[4];
        ab_of(r.R, r.mean, m, o);
        const size_t row = (size_t)y * at.TW;
        c.ab[row] = make_float2(o[0], o[1]);
        c.ab[at.plane + row] = make_float2(o[2], o[3]);
    }
    __device__ __forceinline__ void emit2(int y, const Col& c, const Raw2& r, const float (&m0)[4], const float (&m1)[4], bool second) const
    {
        const double Ra[3] = {r.R[0].x, r.R[1].x, r.R[2].x}, Rb[3] = {r.R[0].y, r.R[1].y, r.R[2].y};
        const float ma[3] = {r.mean[0].x, r.mean[1].x, r.mean[2].x}, mb[3] = {r.mean[0].y, r.mean[1].y, r.mean[2].y};
        float o0[4], o1[4];
        ab_of(Ra, ma, m0, o0);
        ab_of(Rb, mb, m1, o1);
        const size_t row = (size_t)y * at.TW;
        if (second) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f va = {o0[0], o0[1], o1[0], o1[1]}, vb = {o0[2], o0[3], o1[2], o1[3]};
            __builtin_nontemporal_store(va, reinterpret_cast<v4f*>(c.ab + row));            // written once, read by the next launch:
            __builtin_nontemporal_store(vb, reinterpret_cast<v4f*>(c.ab + at.plane + row)); // keep it from displacing the statistics in L2
        } else {
            c.ab[row] = make_float2(o0[0], o0[1]);
            c.ab[at.plane + row] = make_float2(o0[2], o0[3]);
        }
    }
};
// box(a_c), box(b) from the planar a/b volume                                                              (M.cpp:2849-2850)
struct QSrcP {
    const float2* ab;  // strip-major tiles, see ABTiles
    ABTiles at;
    struct Col { const float2* p; };
    struct Raw { float2 u, v; };
    __device__ __forceinline__ Col col(int x, int k) const { return Col{ab + (size_t)k * at.slice_stride + at.col_offset(x)}; }
    __device__ __forceinline__ Raw fetch(int y, const Col& c) const
    {
        const size_t row = (size_t)y * at.TW;
        return Raw{c.p[row], c.p[at.plane + row]};
    }
    static constexpr int KEEP = 4;
    __device__ __forceinline__ void keep(const Raw& r, uint32_t (&w)[4]) const
    {
        w[0] = __float_as_uint(r.u.x); w[1] = __float_as_uint(r.u.y); w[2] = __float_as_uint(r.v.x); w[3] = __float_as_uint(r.v.y);
    }
    typedef NoRaw LRaw;
    __device__ __forceinline__ LRaw leave_fetch(int, const Col&) const { return LRaw(); }
    __device__ __forceinline__ Raw leave(const uint32_t (&w)[4], const LRaw&) const
    {
        return Raw{make_float2(__uint_as_float(w[0]), __uint_as_float(w[1])), make_float2(__uint_as_float(w[2]), __uint_as_float(w[3]))};
    }
    __device__ __forceinline__ void eval(const Raw& r, const Col&, float (&v)[4]) const { v[0] = r.u.x; v[1] = r.u.y; v[2] = r.v.x; v[3] = r.v.y; }
};
// q = sum_c box(a_c)*I_c + box(b), slice-independent 3-channel guide: guide pixels and results of the two columns as pairs  (M.cpp:2851-2852)
struct QDstP {
    static constexpr bool PAIR = true;
    GuideAccT<false> g;
    float* q;  // [n][H][W]
    int H, W;
    struct Col { const uint32_t* a; float2 sc; float* q; };
    struct Raw { uint32_t u; };
    struct Raw2 { uint2 u; };
    __device__ __forceinline__ bool active(int, int, int) const { return true; }
    __device__ __forceinline__ Col col(int x, int k) const { return Col{g.A + x, g.scales[0], q + (size_t)k * H * W + x}; }
    __device__ __forceinline__ Raw fetch(int y, const Col& c) const { return Raw{c.a[(size_t)y * W]}; }
    __device__ __forceinline__ Raw2 fetch2(int y, const Col& c) const
    {
        const uint32_t* p = c.a + (size_t)y * W;
        Raw2 r;
        r.u = make_uint2(p[0], p[1]);  // adjacent dwords: one 8-byte load
        return r;
    }
    __device__ __forceinline__ float q_of(uint32_t u, const Col& c, const float (&m)[4]) const
    {
        const float2 sc = c.sc;
        float I[3];
        I[0] = (float)(u & 0xffu) * sc.x + sc.y;
        I[1] = (float)((u >> 8) & 0xffu) * sc.x + sc.y;
        I[2] = (float)((u >> 16) & 0xffu) * sc.x + sc.y;
        float dot = 0.0f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float pr = m[ch] * I[ch];
            dot = (ch == 0) ? pr : dot + pr;
        }
        return dot + m[3];
    }
    __device__ __forceinline__ void emit(int y, const Col& c, const Raw& r, const float (&m)[4]) const { c.q[(size_t)y * W] = q_of(r.u, c, m); }
    __device__ __forceinline__ void emit2(int y, const Col& c, const Raw2& r, const float (&m0)[4], const float (&m1)[4], bool second) const
    {
        const float q0 = q_of(r.u.x, c, m0), q1 = q_of(r.u.y, c, m1);
        float* o = c.q + (size_t)y * W;
        if (second) {
            typedef float v2f __attribute__((ext_vector_type(2)));
            const v2f v = {q0, q1};
            __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(o));  // streamed out, read again only by the WTA sweep
        } else {
            o[0] = q0;
        }
    }
};

// getCostSAD_d (M.cpp:2442-2503): |grayL - grayR shifted| as f32, box mean
struct SadSrc {
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
    static constexpr bool PAIR = false;
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
struct WalkOpts { int band = 0; int wg_strips = -1; int interleave = 0; };

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
              : k == 15 ? k_box_walk<NP, CPL, ND, (RING ? 3 : WPE), NANSAFE, 15, 0, (RING ? 1 : PF), Src, Dst>
                        : k_box_walk<NP, CPL, ND, (RING ? 3 : WPE), NANSAFE, 0, 0, (RING ? 1 : PF), Src, Dst>;
    // four slices of one strip per workgroup when there are enough slices (1080p D=128: GuidedF 24.1 -> 22.9 ms, BLO1 -7 %,
    // GuidedF_2 -1 %); four strips of the one slice otherwise
    int slice_par = nzs >= 4 ? 1 : 0;
    if (o.wg_strips == 1) slice_par = 0;
    const int spw = slice_par ? 4 : 1;
    const int ngx = slice_par ? nxw : (nxw + 3) / 4, nby = (H + band - 1) / band, nzg = (nzs + spw - 1) / spw;
    const long long nwg = (long long)((ngx * nby + 7) / 8) * 8 * nzg;  // regions rounded up to a multiple of the 8 XCDs
    if (nwg > 0x7fffffffLL) return ASW_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(BW), lds, s, src, dst, H, W, k, band, nxw, n, ngx, nby, slice_par | ((o.interleave || !ring) ? 2 : 0));
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

// ---- fused a/b -> q walk: slice-independent 3-channel guide, 15x15 (GuidedF_2), finite costs ------------------------------------
// One walk does both box stages: the a/b values of a row go straight from the first stage's horizontal pass into the second
// stage's vertical running sums, so nothing of the 4.4 GB a/b volume touches memory: a step loads the cost and the guide word of
// the entering row, the guide statistics of the a/b row and the guide pixels of the q row -- all but the cost hit in L2 -- and
// stores two q values.  HBM traffic per 1080p D=128 frame: 4.3 GB instead of 13.6 GB.  Price: 2 x 14 halo columns per 128-column
// strip (100 outputs) and 28 warm-up rows per band.
// Borders: a/b at a virtual row / column -k is a/b(k) (BORDER_REFLECT_101 of the second boxFilter); the first stage evaluated at
// the virtual position sees the mirrored window of position k -- the same multiset of cost samples, summed in f64 in another
// order -- and takes the statistics of position k, so no border case exists in the walk.
// Used for frames that give it ~10 rounds of tall bands (guided_uses_fused: 1080p D=128, 4K; ASW_GUIDED_FUSED=0 / 1 force either
// path): there it runs as fast as the two passes within the spread of the boxes (kernel 3.65-3.83 against 3.42-3.60 ms, whole
// aggregation by the library's events 3.63 / 3.64 against 3.59 / 3.73 ms on two boxes); on small frames the two passes win.  A
// first form with both stages in ONE wavefront (both rings: 284-470 registers, one wavefront per SIMD) took 4.18-5.70 ms; the
// form below splits the stages over a PAIR of wavefronts.  Results are bit-identical to the two-pass path.
struct FusedArgs {
    GuideAccT<false> g;
    const float* P;          // raw cost volume [n][H][W]
    const float2* pscales;   // per-slice normalize() parameters
    StatsPlanes sp;
    float* q;                // [n][H][W]
    int H, W, n, band, nxw, nby;
};

// Producer / consumer pair: wavefront A runs the first stage of a strip (cost ring, column sums, horizontal pass, a/b arithmetic) and hands every a/b
// row through a double-buffered LDS slot to wavefront B, which runs the second stage (a/b ring, column sums, horizontal pass, q):
// each holds one ring -- 236 registers, TWO wavefronts per SIMD -- and the two stages of a strip run concurrently, one row
// apart, with one workgroup barrier per row.  A workgroup = NPAIR pairs (slices of one strip and band); one pair per workgroup
// measured 3.71 against 3.84 ms for two (the barrier then couples only the two wavefronts that exchange data).
// Counters (profiles/r03/guided_fused/): VALU busy 55 % of a SIMD's cycles; a wavefront waits 16 % of its cycles in s_waitcnt
// (average VMEM latency 840 cycles, LDS 118) and 22 % at the barrier; the kernel issues 1.04e9 VALU instructions against 8.3e8
// of the two passes (100 instead of 114 outputs per strip, the exchange).  Variants that lost (template parameters kept):
// the statistics of the next a/b row loaded at the end of the producer's step (STATF: 4.27-4.34 ms), the producer's LDS reads of
// all four planes in flight (HGRP = 4: 3.70-3.78), b computed by the consumer (4.04), s_setprio for the producer (3.81).
template <int PF, int HGRP, bool STATF, int NPAIR>
__global__ __launch_bounds__(128 * NPAIR) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_guided_pair3(FusedArgs a)
{
    constexpr int K = 15, HL = 7, SW = 128, XO1 = SW - (K - 1), XO2 = SW - 2 * (K - 1), NPL = 4, HP = (K - 1) / 2;
    constexpr int NPH = PF + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wv & 1, pr = wv >> 1;  // role 0: first stage (A), 1: second stage (B); pair 0 / 1
    // per pair: double hsA[NPL][SW+2] | double hsB[NPL][SW+2] | float xch[2][NPL][SW]
    constexpr int PAIR_BYTES = 2 * NPL * (SW + 2) * 8 + 2 * NPL * SW * 4;
    unsigned char* pbase = smem + (size_t)pr * PAIR_BYTES;
    double* hs = reinterpret_cast<double*>(pbase) + (role ? NPL * (SW + 2) : 0);
    float* xch = reinterpret_cast<float*>(pbase + 2 * NPL * (SW + 2) * 8);
    const int H = a.H, W = a.W;
    const int nzg = (a.n + NPAIR - 1) / NPAIR;
    const int wj = blockIdx.x >> 3;
    const int nreg = a.nxw * a.nby, rpx = (nreg + 7) >> 3;
    const int reg = (blockIdx.x & 7) * rpx + wj / nzg;
    if (wj / nzg >= rpx || reg >= nreg) return;  // whole workgroup
    const int xw = reg % a.nxw, by = reg / a.nxw;
    const int kz = (wj % nzg) * NPAIR + pr;
    const bool live = kz < a.n;                   // a pair without a slice still keeps the workgroup's barriers
    const int kzc = min(kz, a.n - 1);
    const int xo0 = xw * XO2, c0 = 2 * lane;
    const int y0 = by * a.band, y1 = min(H, y0 + a.band);
    const size_t plane = (size_t)H * W;
    // wave-uniform bases + 32-bit per-lane column offsets: the loads take the SGPR-base form, no 64-bit address arithmetic per lane
    const float* Pz = a.P + (size_t)kzc * plane;
    const uint32_t* GA = a.g.A;
    const double* RS = a.sp.R;
    const float* MS = a.sp.M;
    uint32_t xi[2], xs_[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        xi[c] = (uint32_t)reflect101_idx(xo0 - 2 * HL + c0 + c, W);
        xs_[c] = (uint32_t)reflect101_idx(xo0 - HL + c0 + c, W);
    }
    const int xq = xo0 + c0;
    const bool ab_lane = c0 < XO1, q_lane = live && c0 < XO2 && xq < W, q_second = c0 + 1 < XO2 && xq + 1 < W;
    const uint32_t xqc = (uint32_t)min(xq, W - 1);
    float* Qz = a.q + (size_t)kzc * plane;
    const float2 psc = a.pscales[kzc], gsc = a.g.scales[0];
    const double scale = 1.0 / ((double)K * (double)K);

    double vs[2][NPL];          // A: {P, I_c P} column sums; B: {a_c, b} column sums
    v16u ring[2][NPL];          // A uses [c][0..1] = {cost, guide word}; B all four planes
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int p = 0; p < NPL; p++) { vs[c][p] = 0.0; ring[c][p] = 0; }
    int slot = 0;
    const int iters = (y1 - y0) + 2 * (K - 1);  // A steps 0 .. iters-1; B handles a/b row t = it - (K-1) in iteration it

    float fP[NPH][2];
    uint32_t fG[NPH][2];
    double fRn[2][3];  // STATF: statistics of the NEXT iteration's a/b row, loaded at the end of a producer step (one set of
    float fMn[2][3];   // registers: they are consumed before the next load is issued, so nothing rotates)
    auto issue = [&](int s, auto slot_c) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_c)::value;
        const size_t rn = (size_t)reflect101_idx(y0 - 2 * HL + s, H) * W;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            fP[SL][c] = (Pz + rn)[xi[c]];
            fG[SL][c] = (GA + rn)[xi[c]];
        }
    };
    auto issue_stats = [&](int s) __attribute__((always_inline)) {
        const size_t ra = (size_t)reflect101_idx(y0 - 3 * HL + s, H) * W;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                fRn[c][ch] = (RS + (ch * plane + ra))[xs_[c]];
                fMn[c][ch] = (MS + (ch * plane + ra))[xs_[c]];
            }
    };
    if (role == 0) {
        if constexpr (PF >= 1) issue(0, std::integral_constant<int, 0>());
        if constexpr (PF >= 2) issue(1, std::integral_constant<int, 1>());
    }
    auto guide = [&](uint32_t u, float (&I)[3]) __attribute__((always_inline)) {
        I[0] = (float)(u & 0xffu) * gsc.x + gsc.y;
        I[1] = (float)((u >> 8) & 0xffu) * gsc.x + gsc.y;
        I[2] = (float)((u >> 16) & 0xffu) * gsc.x + gsc.y;
    };
    auto hpass = [&](bool reader, float (&m)[2][NPL], auto grp_c) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NPL; p++) {
            hs[p * (SW + 2) + c0] = vs[0][p] + vs[1][p];
            hs[p * (SW + 2) + c0 + 1] = vs[0][p];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int p = 0; p < NPL; p++) { m[0][p] = 0.0f; m[1][p] = 0.0f; }
        if (reader) {
            constexpr int GRP = decltype(grp_c)::value;
#pragma unroll
            for (int p0 = 0; p0 < NPL; p0 += GRP) {
                double bb[GRP][2 * HP + 2];
#pragma unroll
                for (int g = 0; g < GRP; g++) {
                    const double* b = hs + (p0 + g) * (SW + 2) + c0;
#pragma unroll
                    for (int i = 2; i < 2 * HP + 2; i++)
                        if (!(i & 1) || i == 2 * HP + 1) bb[g][i] = b[i];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < GRP; g++) {
                    const int p = p0 + g;
                    double t = bb[g][2];
#pragma unroll
                    for (int i = 2; i < HP; i++) t = t + bb[g][2 * i];
                    const double s0 = ((vs[0][p] + vs[1][p]) + t) + bb[g][2 * HP + 1];
                    const double s1 = (vs[1][p] + t) + bb[g][2 * HP];
                    m[0][p] = (float)(s0 * scale);
                    m[1][p] = (float)(s1 * scale);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // one iteration: A computes input row `it` (and, from it = K-1 on, hands a/b row it-(K-1) to B through slot it & 1); barrier;
    // B consumes that row (and, from it = 2(K-1) on, emits q row it - 2(K-1)).  B works on row t while A is already at row t+1.
    auto step = [&](int it, auto ph_c, auto sub1_c, auto out1_c, auto has2_c, auto sub2_c, auto out2_c) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr bool SUB1 = decltype(sub1_c)::value, OUT1 = decltype(out1_c)::value, HAS2 = decltype(has2_c)::value,
                       SUB2 = decltype(sub2_c)::value, OUT2 = decltype(out2_c)::value;
        float* xs = xch + (it & 1) * (NPL * SW);
        if (role == 0) {
            issue(it + PF, std::integral_constant<int, (PH + PF) % NPH>());
            double fR[2][3];
            float fM[2][3];
            if constexpr (OUT1) {
                if constexpr (!STATF) {
                    const size_t ra = (size_t)reflect101_idx(y0 - 3 * HL + it, H) * W;  // a/b row emitted in this iteration
#pragma unroll
                    for (int c = 0; c < 2; c++)
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            fR[c][ch] = (RS + (ch * plane + ra))[xs_[c]];
                            fM[c][ch] = (MS + (ch * plane + ra))[xs_[c]];
                        }
                }
            }
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if constexpr (SUB1) {
                    const float po = __uint_as_float(ring[c][0][slot]) * psc.x + psc.y;
                    float Io[3];
                    guide(ring[c][1][slot], Io);
                    vs[c][0] = vs[c][0] - (double)po;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) vs[c][1 + ch] = vs[c][1 + ch] - (double)(Io[ch] * po);
                }
                ring[c][0][slot] = __float_as_uint(fP[PH][c]);
                ring[c][1][slot] = fG[PH][c];
                const float pn = fP[PH][c] * psc.x + psc.y;
                float In[3];
                guide(fG[PH][c], In);
                vs[c][0] = vs[c][0] + (double)pn;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) vs[c][1 + ch] = vs[c][1 + ch] + (double)(In[ch] * pn);
            }
            slot = slot + 1 == K ? 0 : slot + 1;
            if constexpr (OUT1) {
                float m1[2][NPL];
                hpass(ab_lane, m1, std::integral_constant<int, HGRP>());
                float o[2][NPL];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float meanP = m1[c][0];
                    float dot = 0.0f;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const float mI = STATF ? fMn[c][ch] : fM[c][ch];
                        const float mp = mI * meanP;
                        const float cov = m1[c][1 + ch] - mp;
                        const float ac = (float)((double)cov * (STATF ? fRn[c][ch] : fR[c][ch]));
                        o[c][ch] = ac;
                        const float pr2 = ac * mI;
                        dot = (ch == 0) ? pr2 : dot + pr2;
                    }
                    o[c][3] = meanP - dot;
                }
#pragma unroll
                for (int p = 0; p < NPL; p++) *reinterpret_cast<float2*>(xs + p * SW + c0) = make_float2(o[0][p], o[1][p]);
            }
            if constexpr (STATF) issue_stats(it + 1);
        }
        uint2 fQ = make_uint2(0u, 0u);
        if constexpr (OUT2) {
            if (role == 1) {  // the q row's guide pixels: in flight across the barrier
                const uint32_t* pq = GA + (size_t)(y0 + it - 2 * (K - 1)) * W;
                fQ = make_uint2(pq[xqc], pq[xqc + 1]);
            }
        }
        __syncthreads();  // (a bare s_waitcnt lgkmcnt(0) + s_barrier, without the fence's vmcnt drain, measured the same)
        if constexpr (HAS2) {
            if (role == 1) {
                float o[2][NPL];
#pragma unroll
                for (int p = 0; p < NPL; p++) {
                    const float2 v = *reinterpret_cast<const float2*>(xs + p * SW + c0);
                    o[0][p] = v.x;
                    o[1][p] = v.y;
                }

#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int p = 0; p < NPL; p++) {
                        if constexpr (SUB2) vs[c][p] = vs[c][p] - (double)__uint_as_float(ring[c][p][slot]);
                        ring[c][p][slot] = __float_as_uint(o[c][p]);
                        vs[c][p] = vs[c][p] + (double)o[c][p];
                    }
                slot = slot + 1 == K ? 0 : slot + 1;
                if constexpr (OUT2) {
                    float m2[2][NPL];
                    hpass(c0 < XO2, m2, std::integral_constant<int, 2>());
                    float qv[2];
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        float I[3];
                        guide(c == 0 ? fQ.x : fQ.y, I);
                        float dot = 0.0f;
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            const float pr2 = m2[c][ch] * I[ch];
                            dot = (ch == 0) ? pr2 : dot + pr2;
                        }
                        qv[c] = dot + m2[c][3];
                    }
                    if (q_lane) {
                        float* o2 = Qz + (size_t)(y0 + it - 2 * (K - 1)) * W + xqc;
                        if (q_second) {
                            typedef float v2f __attribute__((ext_vector_type(2)));
                            const v2f v = {qv[0], qv[1]};
                            __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(o2));
                        } else {
                            o2[0] = qv[0];
                        }
                    }
                }
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    int it = 0, ph = 0;
    auto run = [&](int it_end, auto a1, auto b1, auto h2, auto a2, auto b2) __attribute__((always_inline)) {
        if constexpr (NPH > 1) { if (ph == 1 && it < it_end) { step(it, std::integral_constant<int, 1 % NPH>(), a1, b1, h2, a2, b2); it++; ph = 2 % NPH; } }
        if constexpr (NPH > 2) { if (ph == 2 && it < it_end) { step(it, std::integral_constant<int, 2 % NPH>(), a1, b1, h2, a2, b2); it++; ph = 0; } }
        if (ph == 0) {
            for (; it + NPH <= it_end; it += NPH) {
                step(it, std::integral_constant<int, 0>(), a1, b1, h2, a2, b2);
                if constexpr (NPH > 1) step(it + 1, std::integral_constant<int, 1 % NPH>(), a1, b1, h2, a2, b2);
                if constexpr (NPH > 2) step(it + 2, std::integral_constant<int, 2 % NPH>(), a1, b1, h2, a2, b2);
            }
            if constexpr (NPH > 1) { if (it < it_end) { step(it, std::integral_constant<int, 0>(), a1, b1, h2, a2, b2); it++; ph = 1; } }
            if constexpr (NPH > 2) { if (it < it_end) { step(it, std::integral_constant<int, 1 % NPH>(), a1, b1, h2, a2, b2); it++; ph = 2; } }
        }
    };
    run(min(K - 1, iters), F(), F(), F(), F(), F());          // A accumulates
    run(min(K, iters), F(), T(), T(), F(), F());              // first a/b row
    run(min(2 * (K - 1), iters), T(), T(), T(), F(), F());    // B accumulates
    run(min(2 * K - 1, iters), T(), T(), T(), F(), T());      // first q row
    run(iters, T(), T(), T(), T(), T());
}

// ---- q pass of the 6-channel guide (GuidedF / GuidedF_3), 15x15, finite costs: a PAIR of ring wavefronts per strip ---------------
// The seven planes {a_0..a_5, b} of a pixel are two 16-byte halves {a_0, a_1, a_2, -}, {a_3, a_4, a_5, b} -- one per guide word --, each a dense [n][H][W] float4
// volume of its own (as ONE 32-byte record, each wavefront of the pair below read every other 16 bytes: 3.40 ms).  A register ring for all of them (224 registers with two
// columns per lane) does not exist, so k_box_walk fetched the leaving row again and walked 32-row bands to keep that second read
// in L2 (44 % warm-up rows): 3.7-3.9 ms, memory-bound.  Here wavefront 0 of a pair takes the first half {a_0, a_1, a_2},
// wavefront 1 the other 16 {a_3, a_4, a_5, b}: each keeps ITS four planes of the last 15 rows in a register ring (128 registers,
// nothing is read twice, bands as tall as the launch geometry allows), runs its own column sums and horizontal pass, and the
// dot product q = sum_c mean(a_c) I_c + mean(b) (left to right, M.cpp:22-31, 2851-2852) is handed from wavefront 0 to wavefront 1
// through LDS after its third term: one workgroup barrier per output row.  A wavefront needs only ITS guide word.
struct Q6Args {
    GuideAccT<true> g;
    const float* ab;  // [2][n][H][W] float4: {a_0, a_1, a_2, -} planes, then {a_3, a_4, a_5, b}
    float* q;         // [n][H][W]
    int H, W, n, band, nxw, nby;
};

template <int PF, bool NANSAFE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_q6_pair(Q6Args a)
{
    constexpr int K = 15, HL = 7, SW = 128, XO = SW - (K - 1), NPL = 4, HP = (K - 1) / 2;
    constexpr int NPH = PF + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* hs = reinterpret_cast<double*>(smem) + (size_t)role * NPL * (SW + 2);   // [NPL][SW+2] per wavefront
    float* xch = reinterpret_cast<float*>(smem + 2 * NPL * (SW + 2) * 8);             // [2][SW]: partial dot products, double-buffered
    const int H = a.H, W = a.W;
    const int wj = blockIdx.x >> 3;
    const int nreg = a.nxw * a.nby, rpx = (nreg + 7) >> 3;
    const int reg = (blockIdx.x & 7) * rpx + wj / a.n;
    if (wj / a.n >= rpx || reg >= nreg) return;  // whole workgroup
    const int xw = reg % a.nxw, by = reg / a.nxw;
    const int kz = wj % a.n;
    const int xo0 = xw * XO, c0 = 2 * lane;
    const int y0 = by * a.band, y1 = min(H, y0 + a.band);
    const size_t plane = (size_t)H * W;
    const float4* pc[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int xi = reflect101_idx(xo0 - HL + c0 + c, W);
        pc[c] = reinterpret_cast<const float4*>(a.ab) + (size_t)role * a.n * plane + (size_t)kz * plane + xi;  // [half][n][H][W] float4
    }
    const int xq = xo0 + c0;
    const bool reader = c0 < XO, q_lane = reader && xq < W, q_second = c0 + 1 < XO && xq + 1 < W;
    const typename GuideAccT<true>::Col gcol0 = a.g.col(min(xq, W - 1), kz), gcol1 = a.g.col(min(xq + 1, W - 1), kz);
    float* qo = a.q + (size_t)kz * plane + min(xq, W - 1);
    const double scale = 1.0 / ((double)K * (double)K);

    double vs[2][NPL];
    v16u ring[2][NPL];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int p = 0; p < NPL; p++) { vs[c][p] = 0.0; ring[c][p] = 0; }
    int slot = 0;
    const int steps = (y1 - y0) + K - 1;

    float4 fN[NPH][2];
    auto issue = [&](int s, auto slot_c) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_c)::value;
        const size_t rn = (size_t)reflect101_idx(y0 - HL + s, H) * W;  // in float4 units
#pragma unroll
        for (int c = 0; c < 2; c++) fN[SL][c] = pc[c][rn];
    };
    if constexpr (PF >= 1) issue(0, std::integral_constant<int, 0>());
    if constexpr (PF >= 2) issue(1, std::integral_constant<int, 1>());

    auto step = [&](int s, auto ph_c, auto sub_c, auto out_c) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr bool SUB = decltype(sub_c)::value, OUT = decltype(out_c)::value;
        issue(s + PF, std::integral_constant<int, (PH + PF) % NPH>());
        const int y = y0 + s - (K - 1);
        uint32_t gw[2][1];  // this wavefront's guide word of the two output pixels
        if constexpr (OUT) {
            const size_t row = (size_t)y * W;
            gw[0][0] = role == 0 ? gcol0.a[row] : gcol0.b[row];  // every wavefront needs ITS guide word only
            gw[1][0] = role == 0 ? gcol1.a[row] : gcol1.b[row];
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float e[NPL] = {fN[PH][c].x, fN[PH][c].y, fN[PH][c].z, fN[PH][c].w};
#pragma unroll
            for (int p = 0; p < NPL; p++) {
                if constexpr (SUB) vs[c][p] = vs[c][p] - (double)__uint_as_float(ring[c][p][slot]);
                ring[c][p][slot] = __float_as_uint(e[p]);
                vs[c][p] = vs[c][p] + (double)e[p];
            }
        }
        slot = slot + 1 == K ? 0 : slot + 1;
        if constexpr (NANSAFE && OUT) {
            // a sliding sum never loses a NaN / inf once it has entered: rebuild a poisoned one from the 15 rows of its window, which
            // the ring holds (oldest first: slot now points at it) -- finite data never takes the branch (see k_box_walk)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                bool poisoned = false;
#pragma unroll
                for (int p = 0; p < NPL; p++) poisoned = poisoned || !__builtin_isfinite(vs[c][p]);
                if (poisoned) {
                    double acc[NPL];
#pragma unroll
                    for (int p = 0; p < NPL; p++) acc[p] = 0.0;
                    int idx = slot;
                    for (int j = 0; j < K; j++) {
#pragma unroll
                        for (int p = 0; p < NPL; p++) acc[p] = acc[p] + (double)__uint_as_float(ring[c][p][idx]);
                        idx = idx + 1 == K ? 0 : idx + 1;
                    }
#pragma unroll
                    for (int p = 0; p < NPL; p++) vs[c][p] = acc[p];
                }
            }
        }
        if constexpr (OUT) {
#pragma unroll
            for (int p = 0; p < NPL; p++) {
                hs[p * (SW + 2) + c0] = vs[0][p] + vs[1][p];
                hs[p * (SW + 2) + c0 + 1] = vs[0][p];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float m[2][NPL];
#pragma unroll
            for (int p = 0; p < NPL; p++) { m[0][p] = 0.0f; m[1][p] = 0.0f; }
            if (reader) {
                constexpr int GRP = 2;
#pragma unroll
                for (int p0 = 0; p0 < NPL; p0 += GRP) {
                    double bb[GRP][2 * HP + 2];
#pragma unroll
                    for (int g = 0; g < GRP; g++) {
                        const double* b = hs + (p0 + g) * (SW + 2) + c0;
#pragma unroll
                        for (int i = 2; i < 2 * HP + 2; i++)
                            if (!(i & 1) || i == 2 * HP + 1) bb[g][i] = b[i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < GRP; g++) {
                        const int p = p0 + g;
                        double t = bb[g][2];
#pragma unroll
                        for (int i = 2; i < HP; i++) t = t + bb[g][2 * i];
                        const double s0 = ((vs[0][p] + vs[1][p]) + t) + bb[g][2 * HP + 1];
                        const double s1 = (vs[1][p] + t) + bb[g][2 * HP];
                        m[0][p] = (float)(s0 * scale);
                        m[1][p] = (float)(s1 * scale);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float* xs = xch + (s & 1) * SW;
            const float2 sc[2] = {gcol0.sc, gcol1.sc};
            float I[2][3];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                I[c][0] = (float)(gw[c][0] & 0xffu) * sc[c].x + sc[c].y;
                I[c][1] = (float)((gw[c][0] >> 8) & 0xffu) * sc[c].x + sc[c].y;
                I[c][2] = (float)((gw[c][0] >> 16) & 0xffu) * sc[c].x + sc[c].y;
            }
            if (role == 0) {  // terms 0..2: word A
                float part[2];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    float dot = m[c][0] * I[c][0];
                    dot = dot + m[c][1] * I[c][1];
                    dot = dot + m[c][2] * I[c][2];
                    part[c] = dot;
                }
                *reinterpret_cast<float2*>(xs + c0) = make_float2(part[0], part[1]);
            }
            __syncthreads();
            if (role == 1) {  // terms 3..5: word B, then mean(b)
                const float2 part = *reinterpret_cast<const float2*>(xs + c0);
                float qv[2];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    float dot = c == 0 ? part.x : part.y;
                    dot = dot + m[c][0] * I[c][0];
                    dot = dot + m[c][1] * I[c][1];
                    dot = dot + m[c][2] * I[c][2];
                    qv[c] = dot + m[c][3];
                }
                if (q_lane) {
                    float* o = qo + (size_t)y * W;
                    if (q_second) {
                        typedef float v2f __attribute__((ext_vector_type(2)));
                        const v2f v = {qv[0], qv[1]};
                        __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(o));
                    } else {
                        o[0] = qv[0];
                    }
                }
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    int s = 0, ph = 0;
    auto run = [&](int s_end, auto sub_c, auto out_c) __attribute__((always_inline)) {
        if constexpr (NPH > 1) { if (ph == 1 && s < s_end) { step(s, std::integral_constant<int, 1 % NPH>(), sub_c, out_c); s++; ph = 2 % NPH; } }
        if constexpr (NPH > 2) { if (ph == 2 && s < s_end) { step(s, std::integral_constant<int, 2 % NPH>(), sub_c, out_c); s++; ph = 0; } }
        if (ph == 0) {
            for (; s + NPH <= s_end; s += NPH) {
                step(s, std::integral_constant<int, 0>(), sub_c, out_c);
                if constexpr (NPH > 1) step(s + 1, std::integral_constant<int, 1 % NPH>(), sub_c, out_c);
                if constexpr (NPH > 2) step(s + 2, std::integral_constant<int, 2 % NPH>(), sub_c, out_c);
            }
            if constexpr (NPH > 1) { if (s < s_end) { step(s, std::integral_constant<int, 0>(), sub_c, out_c); s++; ph = 1; } }
            if constexpr (NPH > 2) { if (s < s_end) { step(s, std::integral_constant<int, 1 % NPH>(), sub_c, out_c); s++; ph = 2; } }
        }
    };
    run(min(K - 1, steps), F(), F());
    run(min(K, steps), F(), T());
    run(steps, T(), T());
}

int launch_q6_pair(hipStream_t s, const GuidedLaunch& a, const GuideAccT<true>& g, int band_opt)
{
    constexpr int XO = 128 - 14;
    Q6Args f;
    f.g = g; f.ab = a.ab; f.q = a.q; f.H = a.H; f.W = a.W; f.n = a.n;
    f.nxw = (a.W + XO - 1) / XO;
    // ~10 rounds of the 1024 two-wavefront workgroups the chip holds; 14 warm-up rows per band
    int nb = (int)std::max<long long>(1, (10240 + (long long)f.nxw * a.n - 1) / ((long long)f.nxw * a.n));
    nb = std::min(nb, std::max(1, a.H / 30));
    f.band = band_opt >= 16 ? band_opt : (a.H + nb - 1) / nb;
    f.nby = (a.H + f.band - 1) / f.band;
    const long long nwg = (long long)((f.nxw * f.nby + 7) / 8) * 8 * a.n;
    if (nwg > 0x7fffffffLL) return ASW_ERR_BAD_ARGUMENT;
    const size_t lds = 2 * 4 * (128 + 2) * sizeof(double) + 2 * 128 * sizeof(float);
    if (a.nan_safe)
        hipLaunchKernelGGL((k_q6_pair<2, true>), dim3((unsigned)nwg), dim3(128), lds, s, f);
    else
        hipLaunchKernelGGL((k_q6_pair<2, false>), dim3((unsigned)nwg), dim3(128), lds, s, f);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

// ---- a/b pass of the 6-channel guide (GuidedF), 15x15, finite costs: a PAIR of wavefronts per strip, one per guide word ---------
// k_box_walk evaluates the seven plane sums {P, I_0 P .. I_5 P} of a column in one wavefront: one column per lane (two need 168
// registers), no pair sums, 32-row bands because the leaving row is fetched again -- 5.3 ms, VALU-bound (2.0e9 instructions).
// Here wavefront w of a pair takes guide word w: planes {P, I_3w P, I_3w+1 P, I_3w+2 P} -- the walk of the 3-channel filter: two
// columns per lane, pair sums, {cost, guide word} of the last 15 rows in a register ring, tall bands --, computes its three a_c
// from its word's statistics, and the dot product sum_c a_c meanI_c (left to right over the six channels, M.cpp:22-31, 2847) is
// handed from wavefront 0 to wavefront 1 through LDS after the third term; wavefront 1 finishes b.  Each writes its 16-byte half
// of the a/b volume ({a_0, a_1, a_2, -} / {a_3, a_4, a_5, b}).  The P plane is summed by both: 8 plane sums instead of 7.
struct AB6Args {
    GuideAccT<true> g;
    const float* P;
    const float2* pscales;
    StatsSplit sp;
    float* ab;  // [2][n][H][W] float4
    int H, W, n, band, nxw, nby;
};

template <int PF, int WPE, bool NANSAFE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_ab6_pair(AB6Args a)
{
    constexpr int K = 15, HL = 7, SW = 128, XO = SW - (K - 1), NPL = 4, HP = (K - 1) / 2;
    constexpr int NPH = PF + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* hs = reinterpret_cast<double*>(smem) + (size_t)role * NPL * (SW + 2);
    float* xch = reinterpret_cast<float*>(smem + 2 * NPL * (SW + 2) * 8);  // [2][SW] partial dot products
    const int H = a.H, W = a.W;
    const int wj = blockIdx.x >> 3;
    const int nreg = a.nxw * a.nby, rpx = (nreg + 7) >> 3;
    const int reg = (blockIdx.x & 7) * rpx + wj / a.n;
    if (wj / a.n >= rpx || reg >= nreg) return;  // whole workgroup
    const int xw = reg % a.nxw, by = reg / a.nxw;
    const int kz = wj % a.n;
    const int xo0 = xw * XO, c0 = 2 * lane;
    const int y0 = by * a.band, y1 = min(H, y0 + a.band);
    const size_t plane = (size_t)H * W;
    const float* pc[2];
    const uint32_t* gc[2];
    const float* st[2];  // statistics of this wavefront's word at the lane's two output columns (slot and shift resolved, see ABDst)
    float2 gsc = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int xi = reflect101_idx(xo0 - HL + c0 + c, W);
        const typename GuideAccT<true>::Col gcol = a.g.col(xi, kz);
        pc[c] = a.P + (size_t)kz * plane + xi;
        gc[c] = role == 0 ? gcol.a : gcol.b;
        gsc = gcol.sc;
        const int x = min(xo0 + c0 + c, W - 1);
        const float* base = a.sp.half[role];
        int slot = a.sp.per_slice ? kz : 0, xs = x;
        if (a.sp.rep) {
            if (role != a.sp.shifted) {
                slot = a.sp.rep[kz];
            } else if (a.sp.interior(x, kz)) {
                base = a.sp.unshifted;
                slot = a.sp.rep[kz];
                xs = x + a.sp.sgn * (a.sp.minD + kz);
            }
        }
        st[c] = base + ((size_t)slot * plane + xs) * SS8;
    }
    const int xq = xo0 + c0;
    const bool reader = c0 < XO;
    (void)xq;
    float* abo_row0 = a.ab + (size_t)role * a.n * plane * 4 + ((size_t)kz * plane + xo0) * 4;  // the strip's first output pixel, row 0
    const float2 psc = a.pscales[kz];
    const double scale = 1.0 / ((double)K * (double)K);

    double vs[2][NPL];
    v16u ring[2][2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int p = 0; p < NPL; p++) vs[c][p] = 0.0;
        ring[c][0] = 0; ring[c][1] = 0;
    }
    int slot = 0;
    const int steps = (y1 - y0) + K - 1;

    float fP[NPH][2];
    uint32_t fG[NPH][2];
    auto issue = [&](int s, auto slot_c) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_c)::value;
        const size_t rn = (size_t)reflect101_idx(y0 - HL + s, H) * W;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            fP[SL][c] = pc[c][rn];
            fG[SL][c] = gc[c][rn];
        }
    };
    if constexpr (PF >= 1) issue(0, std::integral_constant<int, 0>());
    if constexpr (PF >= 2) issue(1, std::integral_constant<int, 1>());
    auto guide = [&](uint32_t u, float (&I)[3]) __attribute__((always_inline)) {
        I[0] = (float)(u & 0xffu) * gsc.x + gsc.y;
        I[1] = (float)((u >> 8) & 0xffu) * gsc.x + gsc.y;
        I[2] = (float)((u >> 16) & 0xffu) * gsc.x + gsc.y;
    };

    auto step = [&](int s, auto ph_c, auto sub_c, auto out_c) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr bool SUB = decltype(sub_c)::value, OUT = decltype(out_c)::value;
        issue(s + PF, std::integral_constant<int, (PH + PF) % NPH>());
        const int y = y0 + s - (K - 1);
        float4 sa[2], sb[2];  // {mean_0..2, den_0}, {den_1, den_2, -, -} of the output row
        if constexpr (OUT) {
            const size_t row = (size_t)y * W * SS8;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float4* p = reinterpret_cast<const float4*>(st[c] + row);
                sa[c] = p[0];
                sb[c] = p[1];
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if constexpr (SUB) {
                const float po = __uint_as_float(ring[c][0][slot]) * psc.x + psc.y;
                float Io[3];
                guide(ring[c][1][slot], Io);
                vs[c][0] = vs[c][0] - (double)po;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) vs[c][1 + ch] = vs[c][1 + ch] - (double)(Io[ch] * po);
            }
            ring[c][0][slot] = __float_as_uint(fP[PH][c]);
            ring[c][1][slot] = fG[PH][c];
            const float pn = fP[PH][c] * psc.x + psc.y;
            float In[3];
            guide(fG[PH][c], In);
            vs[c][0] = vs[c][0] + (double)pn;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) vs[c][1 + ch] = vs[c][1 + ch] + (double)(In[ch] * pn);
        }
        slot = slot + 1 == K ? 0 : slot + 1;
        if constexpr (NANSAFE && OUT) {  // see k_q6_pair: the window's 15 rows {cost, guide word} are in the ring, oldest at `slot`
#pragma unroll
            for (int c = 0; c < 2; c++) {
                bool poisoned = false;
#pragma unroll
                for (int p = 0; p < NPL; p++) poisoned = poisoned || !__builtin_isfinite(vs[c][p]);
                if (poisoned) {
                    double acc[NPL];
#pragma unroll
                    for (int p = 0; p < NPL; p++) acc[p] = 0.0;
                    int idx = slot;
                    for (int j = 0; j < K; j++) {
                        const float pw = __uint_as_float(ring[c][0][idx]) * psc.x + psc.y;
                        float Iw[3];
                        guide(ring[c][1][idx], Iw);
                        acc[0] = acc[0] + (double)pw;
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) acc[1 + ch] = acc[1 + ch] + (double)(Iw[ch] * pw);
                        idx = idx + 1 == K ? 0 : idx + 1;
                    }
#pragma unroll
                    for (int p = 0; p < NPL; p++) vs[c][p] = acc[p];
                }
            }
        }
        if constexpr (OUT) {
#pragma unroll
            for (int p = 0; p < NPL; p++) {
                hs[p * (SW + 2) + c0] = vs[0][p] + vs[1][p];
                hs[p * (SW + 2) + c0 + 1] = vs[0][p];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float m[2][NPL];
#pragma unroll
            for (int p = 0; p < NPL; p++) { m[0][p] = 0.0f; m[1][p] = 0.0f; }
            if (reader) {
                constexpr int GRP = 2;
#pragma unroll
                for (int p0 = 0; p0 < NPL; p0 += GRP) {
                    double bb[GRP][2 * HP + 2];
#pragma unroll
                    for (int g = 0; g < GRP; g++) {
                        const double* b = hs + (p0 + g) * (SW + 2) + c0;
#pragma unroll
                        for (int i = 2; i < 2 * HP + 2; i++)
                            if (!(i & 1) || i == 2 * HP + 1) bb[g][i] = b[i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < GRP; g++) {
                        const int p = p0 + g;
                        double t = bb[g][2];
#pragma unroll
                        for (int i = 2; i < HP; i++) t = t + bb[g][2 * i];
                        const double s0 = ((vs[0][p] + vs[1][p]) + t) + bb[g][2 * HP + 1];
                        const double s1 = (vs[1][p] + t) + bb[g][2 * HP];
                        m[0][p] = (float)(s0 * scale);
                        m[1][p] = (float)(s1 * scale);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // a_c = cov_c / den_c of this word (M.cpp:2796-2846), the partial dot product in the reference's order
            float o[2][4], dotp[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float mean[3] = {sa[c].x, sa[c].y, sa[c].z}, den[3] = {sa[c].w, sb[c].x, sb[c].y};
                const float meanP = m[c][0];
                float dot = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const float mp = mean[ch] * meanP;
                    const float cov = m[c][1 + ch] - mp;
                    const float ac = cov / den[ch];
                    o[c][ch] = ac;
                    const float pr = ac * mean[ch];
                    dot = (ch == 0) ? pr : dot + pr;  // wavefront 1 re-associates below: its terms are ADDED to wavefront 0's sum one by one
                }
                dotp[c] = dot;
                o[c][3] = 0.0f;
            }
            float* xs = xch + (s & 1) * SW;
            if (role == 0) *reinterpret_cast<float2*>(xs + c0) = make_float2(dotp[0], dotp[1]);
            __syncthreads();
            if (role == 1) {
                const float2 part = *reinterpret_cast<const float2*>(xs + c0);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float mean[3] = {sa[c].x, sa[c].y, sa[c].z};
                    float dot = c == 0 ? part.x : part.y;  // ((p0 + p1) + p2), then + p3 + p4 + p5 left to right
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) dot = dot + o[c][ch] * mean[ch];
                    o[c][3] = m[c][0] - dot;  // b = meanP - dot
                }
            }
            // A lane holds two ADJACENT pixels: stored from here, an instruction would cover every other 16 bytes (as nontemporal
            // stores the halves of each 32-byte piece went out separately: 12.4 GB written for an 8.5 GB volume; as plain stores
            // the pass took 5.0-5.3 instead of 4.4 ms).  So the strip's 128 results are transposed through the (now idle) LDS strip:
            // instruction k then writes pixels 64 k + lane -- one dense kilobyte.
            {
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f* tr = reinterpret_cast<v4f*>(hs);
                tr[c0] = v4f{o[0][0], o[0][1], o[0][2], o[0][3]};
                tr[c0 + 1] = v4f{o[1][0], o[1][1], o[1][2], o[1][3]};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const v4f w0 = tr[lane], w1 = tr[64 + lane];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                float* dst = abo_row0 + (size_t)y * W * 4;
                if (xo0 + lane < W) __builtin_nontemporal_store(w0, reinterpret_cast<v4f*>(dst + 4 * lane));
                if (64 + lane < XO && xo0 + 64 + lane < W) __builtin_nontemporal_store(w1, reinterpret_cast<v4f*>(dst + 4 * (64 + lane)));
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    int s = 0, ph = 0;
    auto run = [&](int s_end, auto sub_c, auto out_c) __attribute__((always_inline)) {
        if constexpr (NPH > 1) { if (ph == 1 && s < s_end) { step(s, std::integral_constant<int, 1 % NPH>(), sub_c, out_c); s++; ph = 2 % NPH; } }
        if constexpr (NPH > 2) { if (ph == 2 && s < s_end) { step(s, std::integral_constant<int, 2 % NPH>(), sub_c, out_c); s++; ph = 0; } }
        if (ph == 0) {
            for (; s + NPH <= s_end; s += NPH) {
                step(s, std::integral_constant<int, 0>(), sub_c, out_c);
                if constexpr (NPH > 1) step(s + 1, std::integral_constant<int, 1 % NPH>(), sub_c, out_c);
                if constexpr (NPH > 2) step(s + 2, std::integral_constant<int, 2 % NPH>(), sub_c, out_c);
            }
            if constexpr (NPH > 1) { if (s < s_end) { step(s, std::integral_constant<int, 0>(), sub_c, out_c); s++; ph = 1; } }
            if constexpr (NPH > 2) { if (s < s_end) { step(s, std::integral_constant<int, 1 % NPH>(), sub_c, out_c); s++; ph = 2; } }
        }
    };
    run(min(K - 1, steps), F(), F());
    run(min(K, steps), F(), T());
    run(steps, T(), T());
}

int launch_ab6_pair(hipStream_t s, const GuidedLaunch& a, const GuideAccT<true>& g, const StatsSplit& sp, int band_opt)
{
    constexpr int XO = 128 - 14;
    AB6Args f;
    f.g = g; f.P = a.P; f.pscales = a.pscales; f.sp = sp; f.ab = a.ab; f.H = a.H; f.W = a.W; f.n = a.n;
    f.nxw = (a.W + XO - 1) / XO;
    int nb = (int)std::max<long long>(1, (10240 + (long long)f.nxw * a.n - 1) / ((long long)f.nxw * a.n));
    nb = std::min(nb, std::max(1, a.H / 30));
    f.band = band_opt >= 16 ? band_opt : (a.H + nb - 1) / nb;
    f.nby = (a.H + f.band - 1) / f.band;
    const long long nwg = (long long)((f.nxw * f.nby + 7) / 8) * 8 * a.n;
    if (nwg > 0x7fffffffLL) return ASW_ERR_BAD_ARGUMENT;
    const size_t lds = 2 * 4 * (128 + 2) * sizeof(double) + 2 * 128 * sizeof(float);
    // register target of three wavefronts per SIMD (168, 4 spilled): 4.38 against 4.65 ms at two (172); pipeline depth 2: 4.67
    if (a.nan_safe)
        hipLaunchKernelGGL((k_ab6_pair<1, 3, true>), dim3((unsigned)nwg), dim3(128), lds, s, f);
    else
        hipLaunchKernelGGL((k_ab6_pair<1, 3, false>), dim3((unsigned)nwg), dim3(128), lds, s, f);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_guided_fused3(hipStream_t s, const GuidedLaunch& a, const GuideAccT<false>& g, const StatsPlanes& sp, int band_opt)
{
    constexpr int XO2 = 128 - 28;
    FusedArgs f;
    f.g = g; f.P = a.P; f.pscales = a.pscales; f.sp = sp; f.q = a.q; f.H = a.H; f.W = a.W; f.n = a.n;
    f.nxw = (a.W + XO2 - 1) / XO2;
    // ~10 rounds of the 1024 two-wavefront workgroups the chip holds (four per CU); 28 warm-up rows per band
    int nb = (int)std::max<long long>(1, (10240 + (long long)f.nxw * a.n - 1) / ((long long)f.nxw * a.n));
    nb = std::min(nb, std::max(1, a.H / 60));
    f.band = band_opt >= 30 ? band_opt : (a.H + nb - 1) / nb;
    f.nby = (a.H + f.band - 1) / f.band;
    const long long nwg = (long long)((f.nxw * f.nby + 7) / 8) * 8 * a.n;
    if (nwg > 0x7fffffffLL) return ASW_ERR_BAD_ARGUMENT;
    const size_t lds = 2 * 4 * (128 + 2) * sizeof(double) + 2 * 4 * 128 * sizeof(float);
    hipLaunchKernelGGL((k_guided_pair3<1, 2, false, 1>), dim3((unsigned)nwg), dim3(128), lds, s, f);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

// the 3-channel guided filter (statistics, a/b, q) on the planar layouts, the two big passes in the forms AswTuning selects
template <bool SHIFT>
int launch_guided3(hipStream_t s, const GuidedLaunch& a, const GuideAccT<SHIFT>& g, int nstat)
{
    const AswTuning& t = *a.tune;
    const StatsPlanes sp = stats_planes(a.stats, nstat, a.H, a.W);
    int rc;
    {
        StatsSrc<3, 0, SHIFT> ss{g};
        StatsDstP sd{sp, a.W, (float)a.eps};
        rc = launch_walk<6>(s, ss, sd, a.H, a.W, a.r, nstat);
        if (rc != ASW_OK) return rc;
    }
    if constexpr (!SHIFT) {
        // one fused a/b -> q walk, no a/b volume (k_guided_pair3)
        if (guided_uses_fused(t, 3, nstat > 1, 0, a.nan_safe, a.H, a.W, a.n, a.r)) return launch_guided_fused3(s, a, g, sp, t.band_q);
    }
    const ABTiles at = ab_tiles(a.H, a.W, a.r);
    ABDstP dst{sp, nstat > 1 ? 1 : 0, reinterpret_cast<float2*>(a.ab), at, a.W};
    WalkOpts oab; oab.band = t.band_ab;
    WalkOpts oq; oq.band = t.band_q; oq.wg_strips = t.q_wg_strips;
    const int ring_ab = a.nan_safe ? 0 : t.ring_ab, ring_q = a.nan_safe ? 0 : t.ring_q;
    if (ring_ab) {  // ring = {cost, guide word} of the last 15 rows of both columns: 2 wavefronts per SIMD
        // (the ring holding the cost only and the leaving row's guide word fetched again -- 3 wavefronts per SIMD -- measured
        // 0.3-0.4 ms slower: two more loads per step in a pass whose address unit is as busy as its VALU)
        ABSrc<3, SHIFT> src{g, a.P, a.pscales, a.H, a.W};
        rc = launch_walk_t<4, 2, 1, 2, false, 1, 1>(s, src, dst, a.H, a.W, a.r, a.n, -1, oab);
    } else {
        ABSrc<3, SHIFT> src{g, a.P, a.pscales, a.H, a.W};
        rc = a.nan_safe ? launch_walk_t<4, 2, 1, 3, true>(s, src, dst, a.H, a.W, a.r, a.n, -1, oab)
                        : launch_walk_t<4, 2, 1, 3, false>(s, src, dst, a.H, a.W, a.r, a.n, -1, oab);
    }
    if (rc != ASW_OK) return rc;
    QSrcP qs{reinterpret_cast<const float2*>(a.ab), at};
    auto run_q = [&](const auto& qd) {
        // ring = {a_0, a_1, a_2, b} of the last 15 rows of both columns (128 registers): 2 wavefronts per SIMD.  One column per lane
        // (4 wavefronts) measured 2.75 ms against 1.7: twice the wavefront-rows through the one LDS of the CU
        // loads two steps ahead: 1.61 ms against 1.72 at one step (three: 1.58 with spilled registers)
        if (ring_q) return launch_walk_t<4, 2, 1, 2, false, 1, 2>(s, qs, qd, a.H, a.W, a.r, a.n, -1, oq);
        return a.nan_safe ? launch_walk<4, 1, true>(s, qs, qd, a.H, a.W, a.r, a.n, -1, oq) : launch_walk<4>(s, qs, qd, a.H, a.W, a.r, a.n, -1, oq);
    };
    if constexpr (SHIFT) {
        QDst<3, true> qd{g, a.q, a.H, a.W};
        return run_q(qd);
    } else {
        QDstP qd{g, a.q, a.H, a.W};
        return run_q(qd);
    }
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
        return launch_guided3<false>(s, a, g, 1);
    }
    GuideAccT<true> g{a.guideA, a.guideB, a.gscales, a.guide_per_slice ? 1 : 0, a.W, a.shiftA, a.shiftB, a.minD};
    if (a.C == 3) return launch_guided3<true>(s, a, g, nstat);
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
    const size_t hstride = (size_t)a.n * a.H * a.W * 4;
    ABDst<6> dst{sp, a.ab, a.H, a.W, hstride};
    // one column per lane for the 7-plane a/b pass (6.2 ms): two columns need 128 VGPRs + 33 spilled (9.3 ms); two columns at a
    // 3-waves-per-SIMD register target (148 VGPRs, no spills) take the same time as one column (GuidedF 12.67 vs 12.60 ms)
    // (boxes wider than 32 do not leave outputs in a 64-column strip: those take the two-column form at 148 VGPRs)
    if (a.r == 15 && a.tune->ab6_pair != 0)
        rc = launch_ab6_pair(s, a, g, sp, a.tune->band_ab);
    else if (a.r > 32)
        rc = a.nan_safe ? launch_walk_t<7, 2, 1, 3, true, 0, 0>(s, src, dst, a.H, a.W, a.r, a.n) : launch_walk_t<7, 2, 1, 3, false, 0, 0>(s, src, dst, a.H, a.W, a.r, a.n);
    else
        rc = a.nan_safe ? launch_walk_t<7, 1, 1, 4, true, 0, 0>(s, src, dst, a.H, a.W, a.r, a.n) : launch_walk_t<7, 1, 1, 4, false, 0, 0>(s, src, dst, a.H, a.W, a.r, a.n);
    if (rc != ASW_OK) return rc;
    if (a.r == 15 && a.tune->q6_pair != 0) return launch_q6_pair(s, a, g, a.tune->band_q);
    QSrc<6> qs{a.ab, a.H, a.W, hstride};
    QDst<6, true> qd{g, a.q, a.H, a.W};
    // two columns per lane: 4.5 ms, one: 5.3 ms.  No load FIFO here (PF = 0): 16 floats per row and lane in flight twice over would
    // cost the fourth wavefront per SIMD (5.1 ms at two)
    // (finite data: a register target of 3 wavefronts per SIMD buys the LDS reads of two planes in flight together, 3.84 against 4.01 ms)
    return a.nan_safe ? launch_walk_t<7, 2, 1, 4, true, 0, 0>(s, qs, qd, a.H, a.W, a.r, a.n) : launch_walk_t<7, 2, 1, 3, false, 0, 0>(s, qs, qd, a.H, a.W, a.r, a.n);
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
size_t guided_stats_floats(int C, int nstat, int H, int W)
{
    if (C == 3) return (size_t)nstat * H * W * 9 + 8;  // planar: three f64 + three f32 planes per slot, and room for the pair access at the last pixel
    return (size_t)nstat * H * W * 8 * 3;
}
// The fused walk pays 28 warm-up rows per band and 28 halo columns per strip: it matches the two passes when the frame gives
// it ~10 rounds of tall bands (1080p D=128: 3.64 against 3.73 ms, 4K D=64: 6.97 / 7.03, 720p D=96: 1.49 / 1.49) and loses on small
// ones (640x360 D=64: 0.33 / 0.27 ms, 1242x375 D=192: 1.63 / 1.37, 1080p D=32: 1.15 / 1.08) -- tools/time_fused_shapes.py.
bool guided_uses_fused(const AswTuning& t, int C, int guide_per_slice, int shifted, int nan_safe, int H, int W, int n, int r)
{
    if (C != 3 || guide_per_slice || shifted || nan_safe || r != 15 || H < 16 || t.guided_fused == 0) return false;
    if (t.guided_fused > 0) return true;
    const long long strips = (W + 99) / 100;
    return strips * n * H >= 2000000LL;
}

size_t guided_ab_floats(int C, int n, int H, int W, int r)
{
    if (C == 3) return (size_t)n * ab_tiles(H, W, r).slice_stride * 2 + 8;  // strip-major float2 tiles (ABTiles)
    return (size_t)n * H * W * 8;
}
