// Weighted-median aggregation for the 15x15 window (computeAdaptiveWeight_WeightedMedian, M.cpp:3228-3308), second form:
// "sort the neighbourhood once, let every pixel walk it".
//
// k_wmedian.hip sorts the 225 (cost, weight) pairs of every (pixel, d) from scratch: ~190 of its ~370 instructions per pair
// are the 36-step network.  But the costs of slice d are ONE plane shared by all pixels; only the weights are per pixel.
// For an 8x8 block of pixels the union of the windows is a 22x22 region of that plane (484 samples), and the multimap order
// of any window in it -- ascending cost, ties in row-major window order (M.cpp:3276-3283) -- is the order of the region
// sorted by (cost, row-major region index) restricted to the window's members: a translation keeps row-major order.  So
//   1. k_wm_sort_regions: one wavefront sorts the region of a (block, d) once (512 slots, 8 per lane, f64 keys
//      = cost bits * 1024 + region index: exact integers, compare-exchange = v_min_f64 / v_max_f64) -> list in memory;
//   2. k_wm_pick: a wavefront takes the list of a (block, d) (8 entries per lane, in sorted order) and, for each of the 64
//      pixels, turns every entry into (member of this pixel's window ? weight : -0.0) -- the pixel's 225 weights
//      (wL*wd, in LDS for the whole block, times wR, a row of 225 fetched per pixel) are written into a layout indexed by
//      position relative to the window, so an entry gathers its weight at (its position - the pixel's) with one LDS read;
//      every slot that is no window cell holds -0.0 (adds nothing, and marks "not a member") -- then forms the f64 prefix sums in sorted
//      order (8 per lane + a DPP wavefront scan), finds the first prefix above half of the total (M.cpp:3284-3291) and
//      returns the cost of the last member before it (or of the crossing element itself if it is the first, :3293-3301).
// No per-pixel sort: ~9 instructions per list entry and pixel instead of a 36-step network per pixel.
// The arithmetic is that of k_wmedian.hip (f32 weight product in the reference's order, f64 sums; the association of the
// prefix sums differs from the reference's sequential walk in the last bits only, as there).
#include <stdlib.h>

#include "asw_device.h"
#include "asw_internal.h"
#include "wm_network.h"

namespace {
using namespace wmnet;

constexpr int WIN = 15, HW = 7, NC = WIN * WIN;   // 225 window cells
constexpr int BW = 8, BH = 8, NPIX = BW * BH;     // pixel block
constexpr int RW = BW + 2 * HW, RH = BH + 2 * HW; // 22 x 22 region
constexpr int NREG = RW * RH;                     // 484 samples
constexpr int SLOTS = 512, KPL = 8;
constexpr uint32_t COST_BASE_BITS = 0x45800000u;  // 4096.0f: every TAD C+G cost lies in [4096, 16384) (k_wmedian.hip)
constexpr uint32_t POS_PAD = 1023u;               // position of the 28 padding slots: never inside a window
constexpr double KEY_PAD = 4398046511104.0;       // 2^42: above every real key (cost bits < 2^24, * 1024)
constexpr int WLS = NC + 1;                       // per-pixel weight row (+1: odd stride)
constexpr int PS = 47;                            // row stride of a position (row * 47 + column): 47 = 32 + 15, so the 64 consecutive window cells a
                                                  // wavefront writes at once (15 per row) fall into consecutive LDS banks
constexpr int LAY = WIN * PS + 1;                 // a pixel's weights laid out BY RELATIVE POSITION: 15 rows x 47 + the clamp slot
static_assert((RH - 1) * PS + RW - 1 < 1023, "positions stay below the padding position");
constexpr int PICK_WAVES = 8;

// ---- 1. sort the 22 x 22 cost region of every (block, d) --------------------------------------------------------------------
// grid (blocks, ceil(d_count / 4)), 256 threads: wavefront w sorts slice d_begin + 4 * blockIdx.y + w.
// listC / listP: [block][d_count][512] sorted cost bits / positions (row * 47 + column inside the region).
__global__ __launch_bounds__(256) void k_wm_sort_regions(const float* __restrict__ cost /* [numD][H][W] */, int H, int W, int nbx,
                                                         int d_begin, int d_count, uint32_t* __restrict__ listC,
                                                         uint16_t* __restrict__ listP)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // every XCD takes a contiguous run of (slice group, block) pairs: neighbouring blocks' regions overlap (22 of 8 columns /
    // rows) and share 128-byte lines of the cost plane, which then come from HBM once
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, vid = xcd * (nwg >> 3) + min(xcd, nwg & 7) + (lin >> 3);
    const int dd = (vid / (int)gridDim.x) * 4 + wv;
    if (dd >= d_count) return;  // whole wavefront
    const int blk = vid % (int)gridDim.x, by = blk / nbx, bx = blk - by * nbx;
    const int x0 = bx * BW, y0 = by * BH;
    const float* cp = cost + (size_t)(d_begin + dd) * H * W;
    LaneMasks lm;
#pragma unroll
    for (int b = 0; b < 6; b++) lm.m[b] = (lane & (1 << b)) ? 0u : 0xffffffffu;
    double key[KPL];
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int e = lane * KPL + r;
        if (e < NREG) {
            const int ey = e / RW, ex = e - ey * RW;
            // the window's sample at the REFLECT-padded position (M.cpp:665, 3273)
            const float c = cp[(size_t)reflect_idx(y0 - HW + ey, H) * W + reflect_idx(x0 - HW + ex, W)];
            key[r] = (double)(__float_as_uint(c) - COST_BASE_BITS) * 1024.0 + (double)(ey * PS + ex);
        } else {
            key[r] = KEY_PAD + (double)e;
        }
    }
    bitonic_sort<SLOTS>(key, lm);
    uint32_t oc[KPL], op[KPL];
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const bool pad = key[r] >= KEY_PAD;
        const uint32_t c24 = (uint32_t)(key[r] * (1.0 / 1024.0));             // exact: division by a power of two, truncation
        const uint32_t pos = (uint32_t)(key[r] - (double)c24 * 1024.0);
        oc[r] = pad ? 0u : c24 + COST_BASE_BITS;
        op[r] = pad ? POS_PAD : pos;
    }
    const size_t base = ((size_t)blk * d_count + dd) * SLOTS + (size_t)lane * KPL;
    uint4* pc = reinterpret_cast<uint4*>(listC + base);
    pc[0] = make_uint4(oc[0], oc[1], oc[2], oc[3]);
    pc[1] = make_uint4(oc[4], oc[5], oc[6], oc[7]);
    *reinterpret_cast<uint4*>(listP + base) = make_uint4(op[0] | (op[1] << 16), op[2] | (op[3] << 16), op[4] | (op[5] << 16), op[6] | (op[7] << 16));
}

// ---- 2. every pixel of the block walks the sorted region -------------------------------------------------------------------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double v)  // lanes without a source (row edge, masked rows) read 0
{
    const long long b = __double_as_longlong(v);
    int lo, hi;
    if constexpr (ROWMASK == 0xf) {  // shifts inside a row: bound_ctrl supplies the zeros, no 'old' operand to initialise
        lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, 0xf, 0xf, true);
        hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)((unsigned long long)b >> 32), CTRL, 0xf, 0xf, true);
    } else {                         // row broadcasts into some rows only: the other rows keep old = 0
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

// a[r] of lane l for wave-uniform r and l: a branch tree on the scalar unit and two v_readlane, instead of a chain of
// v_cndmask (each with its s_cmp / s_cselect) over all N registers
template <int N>
__device__ __forceinline__ uint32_t pick_reg(const uint32_t (&a)[N], int r, int l)
{
    static_assert(N == 6 || N == 8, "");
#define ASW_RL(i) (uint32_t)__builtin_amdgcn_readlane((int)a[(i) < N ? (i) : N - 1], l)
    if (r < 4) {
        if (r < 2) return r == 0 ? ASW_RL(0) : ASW_RL(1);
        return r == 2 ? ASW_RL(2) : ASW_RL(3);
    }
    if (r < 6) return r == 4 ? ASW_RL(4) : ASW_RL(5);
    return r == 6 ? ASW_RL(6) : ASW_RL(7);
#undef ASW_RL
}

__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_mov_dpp(v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_mov_dpp(v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_mov_dpp(v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_mov_dpp(v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// grid blocks * NSPLIT (one dimension, see the mapping below), 512 threads: wavefront w takes the slices d_begin + w, w + 8, ... and, for each, the 64 / NSPLIT pixels
// (whole rows of the block) of this workgroup's part.  NSPLIT = 2 halves the LDS weight slab: four workgroups per CU.
// COMPACT (NSPLIT = 4, two pixel rows per part): the windows of the part cover 16 of the region's 22 rows, so each wavefront
// first drops the entries of the other rows from its list (a prefix-sum scatter through LDS that keeps the sorted order):
// 352 entries in 384 slots, 6 per lane instead of 8 in every walk.
template <int NSPLIT, bool COMPACT>
__global__ __launch_bounds__(64 * PICK_WAVES) void k_wm_pick(const float* __restrict__ wLd /* [H][W][225] */,
                                                             const float* __restrict__ wRb /* [H][Wb][225] */,
                                                             const uint32_t* __restrict__ listC, const uint16_t* __restrict__ listP,
                                                             int H, int W, int nbx, int numD, int max_off, int d_begin, int d_count,
                                                             float* __restrict__ out /* [numD][H][W] */)
{
    constexpr int NPART = NPIX / NSPLIT;
    __shared__ float sWL[NPART * WLS];        // 57 856 B / NSPLIT: (wL .mul wd) of this part's pixels
    // 25 600 B (COMPACT): the weights of the pixel a wavefront is working on at its d, laid out by position RELATIVE to the window's first
    // cell (t = dy * 47 + dx + 7): a list entry gathers its weight with ONE LDS read at (its position - the pixel's), no table
    // in between.  Every slot that is not a window cell holds -0.0f: adding it changes no sum (x + -0.0 = x), and its bit
    // pattern says "not a member" (a weight product is never -0.0: both factors are >= +0).
    // COMPACT: the list holds region rows ly0 .. ly0 + 15 only and the part's pixels sit in rows ly0 / ly0 + 1, so an entry's row
    // relative to the window is -1 .. 15: with one guard row in front (and the clamp slot behind) EVERY address an entry can form
    // lies inside the layout -- no clamp, one v_add per gather instead of add / min / add.
    constexpr int LAYW = COMPACT ? (WIN + 2) * PS + 1 : LAY;
    constexpr int ROW0 = COMPACT ? PS : 0;   // slot of the window's first row
    __shared__ float sWR[PICK_WAVES][LAYW];
    constexpr int KE = COMPACT ? 6 : KPL;     // list entries per lane in the walk
    static_assert(!COMPACT || NSPLIT == 4, "the compacted list holds the 16 region rows of a two-row part");
    static_assert(64 * 6 <= LAY && LAY <= (WIN + 2) * PS, "the compaction buffer (original slot << 16 | position) borrows the wavefront's layout");
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // one-dimensional grid of (block, part) pairs; every XCD takes a contiguous run of them, so the parts of a block -- which
    // read the same sorted lists -- run on one XCD at about the same time and the lists come from HBM once, not once per part
    const int nwg = gridDim.x, lin = blockIdx.x;
    const int xcd = lin & 7, vid = xcd * (nwg >> 3) + min(xcd, nwg & 7) + (lin >> 3);
    const int blk = vid / NSPLIT, part = vid - blk * NSPLIT;
    const int by = blk / nbx, bx = blk - by * nbx;
    const int x0 = bx * BW, y0 = by * BH;
    const int Wb = W + max_off;
    const size_t plane = (size_t)H * W;

    const int p_begin = part * NPART;
    if (y0 + (p_begin >> 3) >= H) return;  // this part lies below the image (whole workgroup)
    for (int i = tid; i < NPART * WLS; i += 64 * PICK_WAVES) {
        const int p = p_begin + i / WLS, c = i % WLS;
        const int x = x0 + (p & 7), y = y0 + (p >> 3);
        sWL[i] = (c < NC && x < W && y < H) ? wLd[((size_t)y * W + x) * NC + c] : 0.0f;
    }
    float* wrp = sWR[wv];
    for (int i = lane; i < LAYW; i += 64) wrp[i] = -0.0f;
    // byte offset of window cell c = lane + 64 k in the layout
    int coff[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = min(lane + 64 * k, NC - 1), cy = c / WIN;
        coff[k] = 4 * (ROW0 + cy * PS + (c - cy * WIN) + HW);
    }
    __syncthreads();

    unsigned long long vmask = 0;
    for (int q = 0; q < NPART; q++)
        if (x0 + ((p_begin + q) & 7) < W && y0 + ((p_begin + q) >> 3) < H) vmask |= 1ull << q;
    for (int dd = wv; dd < d_count; dd += PICK_WAVES) {
        const int d = d_begin + dd;
        const size_t lbase = ((size_t)blk * d_count + dd) * SLOTS + (size_t)lane * KPL;
        const uint4 c0 = reinterpret_cast<const uint4*>(listC + lbase)[0], c1 = reinterpret_cast<const uint4*>(listC + lbase)[1];
        const uint4 pp = *reinterpret_cast<const uint4*>(listP + lbase);
        const uint32_t cst[KPL] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const int pos[KPL] = {(int)(pp.x & 0xffffu), (int)(pp.x >> 16), (int)(pp.y & 0xffffu), (int)(pp.y >> 16),
                              (int)(pp.z & 0xffffu), (int)(pp.z >> 16), (int)(pp.w & 0xffffu), (int)(pp.w >> 16)};
        float* od = out + (size_t)d * plane;
        int epos[KE];       // positions of the entries the walk visits
        uint32_t eslot[KE]; // COMPACT: their slot (lane * 8 + r) in the full list, where the cost bits are
        if constexpr (COMPACT) {
            const int ly0 = p_begin >> 3;
            bool keep[KPL];
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < KPL; r++) {
                keep[r] = (uint32_t)(pos[r] - ly0 * PS) < 16u * PS && pos[r] != (int)POS_PAD;  // rows ly0 .. ly0 + 15, no padding entries
                cnt += keep[r] ? 1 : 0;
            }
            uint32_t* cb = reinterpret_cast<uint32_t*>(wrp);  // the layout is rebuilt below
#pragma unroll
            // slots behind the last kept entry: a position of the kept rows whose column (40) is outside every window
            for (int i = 0; i < KE; i++) cb[lane + 64 * i] = (uint32_t)(ly0 * PS + 40);
            int o = wave_inclusive_scan(cnt) - cnt;
#pragma unroll
            for (int r = 0; r < KPL; r++) {
                if (keep[r]) cb[o] = ((uint32_t)(lane * KPL + r) << 16) | (uint32_t)pos[r];
                o += keep[r] ? 1 : 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < KE; i++) {
                const uint32_t e = cb[lane * KE + i];
                epos[i] = 4 * (int)(e & 0xffffu);  // four times the position: the byte offset into the layout needs no shift per pixel
                eslot[i] = e >> 16;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int i = lane; i < 64 * KE; i += 64) wrp[i] = -0.0f;  // the borrowed part of the layout: "not a member" again
        } else {
#pragma unroll
            for (int i = 0; i < KE; i++) { epos[i] = 4 * pos[i]; eslot[i] = 0; }
        }

        // right-image weight row of pixel p at this d: weightWinsR[y][x - offset + numDisparity - 1] (M.cpp:3274)
        float nxt[4];
        auto fetch = [&](int p) {
            const int x = x0 + (p & 7), y = y0 + (p >> 3);
            const float* row = wRb + ((size_t)y * Wb + (x - d + numD - 1)) * NC;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c = lane + 64 * k;
                nxt[k] = c < NC ? row[c] : 0.0f;
            }
        };
        // pixels of the part that lie in the image, as a bit mask (wave-uniform): the loop visits the set bits
        unsigned long long todo = vmask;
        fetch(p_begin);  // the first pixel of a part always exists
        while (todo) {
            const int p = p_begin + __builtin_ctzll(todo);
            todo &= todo - 1;
            // the pixel's 225 weights (wL .mul wd) .mul wR -- f32, in the reference's order, M.cpp:3274 -- are formed here,
            // one cell per lane (conflict-free reads of the pixel's wL row), so that the walk below gathers ONE value per
            // entry: the gathers hit random banks and were what bound the kernel (SQ_LDS_BANK_CONFLICT was half of the LDS cycles)
            const float* wl_row = sWL + (p - p_begin) * WLS;
            char* wq = reinterpret_cast<char*>(wrp);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (lane + 64 * k < NC) *reinterpret_cast<float*>(wq + coff[k]) = wl_row[lane + 64 * k] * nxt[k];
            if (todo) fetch(p_begin + __builtin_ctzll(todo));  // the next pixel's row: in flight under this pixel's arithmetic
            // LDS operations of a wavefront execute in order; the compiler is told that other lanes read these words
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            const int base4 = 4 * ((p >> 3) * PS + (p & 7) - HW);
            const char* wqe = wq + 4 * ROW0 - base4;  // COMPACT: wave-uniform, so a gather address is one addition
            double run = 0.0, pre[KE];
            uint32_t mb = 0;  // member flags of this lane's entries, entry r at bit KE - 1 - r
#pragma unroll
            for (int r = 0; r < KE; r++) {
                float w;  // the cell's weight; -0.0 outside the window
                if constexpr (COMPACT) {
                    w = *reinterpret_cast<const float*>(wqe + epos[r]);
                } else {
                    const uint32_t t4 = min((uint32_t)(epos[r] - base4), (uint32_t)(4 * (LAY - 1)));  // rows above the window wrap to huge values
                    w = *reinterpret_cast<const float*>(wq + t4);
                }
                run = r == 0 ? (double)w : run + (double)w;
                pre[r] = run;
                mb = (mb << 1) | (__float_as_uint(w) != 0x80000000u ? 1u : 0u);
            }
            // the next pixel's weights overwrite the layout: behind this pixel's gathers
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double incl = wave_inclusive_scan(run);
            const double half = readlane_f64(incl, 63) * 0.5;  // cv::sum(weight_img_win)[0] / 2, M.cpp:3284
            // first entry whose prefix (exclusive prefix of the lane + local prefix) exceeds half: compared as
            // pre[r] > half - excl, one subtraction per lane instead of one addition per entry
            const double thr = half - (incl - run);
            int first = KE;
#pragma unroll
            for (int r = KE - 1; r >= 0; r--)
                if (pre[r] > thr) first = r;
            const unsigned long long ball = __ballot(first < KE);
            float res = 0.0f;
            if (ball) {  // wave-uniform
                const int fl = __ffsll((long long)ball) - 1;
                const int fr = __builtin_amdgcn_readlane(first, fl);
                const uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)mb, fl);
                const uint32_t before = bits >> (KE - fr);  // members among entries 0 .. fr-1 of that lane, entry fr-1 at bit 0
                int pl = fl, pr = fr;  // the crossing element itself if nothing precedes it (M.cpp:3293-3296)
                if (before) {
                    pr = fr - 1 - __builtin_ctz(before);
                } else {
                    const unsigned long long lower = __ballot(mb != 0u) & ((1ull << fl) - 1ull);
                    if (lower) {
                        pl = 63 - __builtin_clzll(lower);
                        pr = KE - 1 - __builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)mb, pl));  // its last member
                    }
                }
                if constexpr (COMPACT) {  // (pl, pr) addresses the compacted list: back to the slot that holds the cost
                    const int slot = (int)pick_reg(eslot, pr, pl);
                    pl = slot >> 3;
                    pr = slot & 7;
                }
                res = __uint_as_float(pick_reg(cst, pr, pl));
            }
            if (lane == 0) od[(size_t)(y0 + (p >> 3)) * W + x0 + (p & 7)] = res;
        }
    }
}

}  // namespace

// Scratch for `d_count` slices handled by one pair of launches: sorted cost bits (u32) and positions (u16) of every block.
size_t wmedian_tile_list_slots(int H, int W, int d_count)
{
    const size_t nb = (size_t)((W + BW - 1) / BW) * ((H + BH - 1) / BH);
    return nb * (size_t)d_count * SLOTS;
}

// Slices [d_begin, d_begin + d_count) of the 15x15 weighted median; listC: u32[slots], listP: u16[slots].
int launch_wmedian_tile(hipStream_t s, const float* cost, const float* wLd, const float* wRb, int H, int W, int numD, int max_off,
                        int d_begin, int d_count, uint32_t* listC, uint16_t* listP, float* out, int nsplit)
{
    if (d_count <= 0 || d_begin < 0 || d_begin + d_count > numD) return ASW_ERR_BAD_ARGUMENT;
    const int nbx = (W + BW - 1) / BW, nby = (H + BH - 1) / BH;
    hipLaunchKernelGGL(k_wm_sort_regions, dim3((unsigned)(nbx * nby), (unsigned)((d_count + 3) / 4)), dim3(256), 0, s, cost, H, W, nbx,
                       d_begin, d_count, listC, listP);
    // A/B only: 1 | 2 | 4 = parts per block without compaction, 0 (default) = four parts with the compacted lists
    const dim3 blk(64 * PICK_WAVES);
    const unsigned nb = (unsigned)(nbx * nby);
    if (nsplit == 1)
        hipLaunchKernelGGL((k_wm_pick<1, false>), dim3(nb * 1), blk, 0, s, wLd, wRb, listC, listP, H, W, nbx, numD, max_off, d_begin, d_count, out);
    else if (nsplit == 2)
        hipLaunchKernelGGL((k_wm_pick<2, false>), dim3(nb * 2), blk, 0, s, wLd, wRb, listC, listP, H, W, nbx, numD, max_off, d_begin, d_count, out);
    else if (nsplit == 4)
        hipLaunchKernelGGL((k_wm_pick<4, false>), dim3(nb * 4), blk, 0, s, wLd, wRb, listC, listP, H, W, nbx, numD, max_off, d_begin, d_count, out);
    else
        hipLaunchKernelGGL((k_wm_pick<4, true>), dim3(nb * 4), blk, 0, s, wLd, wRb, listC, listP, H, W, nbx, numD, max_off, d_begin, d_count, out);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}
