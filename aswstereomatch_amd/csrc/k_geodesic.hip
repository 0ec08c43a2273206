// Geodesic-distance support weights and aggregation
// (getColorDist / getWinGeoDist / getGeodesicDist / computeAdaptiveWeight_geodesic, M.cpp:1321-1534).
//
// Weights.  For every pixel the reference relaxes a (win+2)^2 window of L1-colour geodesic distances
// with three raster passes (M.cpp:1339-1388): iterations 0 and 1 are the BACKWARD pass (R, BR, B, BL
// neighbours), iteration 2 the FORWARD pass (L, UL, U, UR) -- App. B-8.  A raster pass is a fixed point
// of itself (every cell already holds the minimum over the monotone paths the pass can build), so
// repeating it changes nothing: the kernel runs one backward and one forward sweep.  All values are
// exact small integers (steps <= 765), FLT_MAX + d == FLT_MAX in f32 is modelled by an integer INF
// that min() never prefers.  One thread owns one pixel; a sweep keeps two window rows of distances and
// of packed BGRX pixels in registers, the L1 colour distance is a single v_sad_u8.  Window state lives
// in cell-major planes W[cell][y][x] (coalesced, and exactly the tap-major layout the aggregation
// wants).
//
// Aggregation (M.cpp:1467-1531).  num += fl(fl(wL*wR)*c), den += fl(wL*wR) with f32 products and f64
// sums.  Every addend is an integer below 2^40 and there are win^2 <= 2^11 of them, so every partial
// sum is an integer below 2^53: the f64 sums are exact in ANY order and E = num/den is bit-identical
// to the reference whatever the schedule.  Tiling follows the bilateral kernel (64x4 pixel tile, DC
// disparities per thread, taps outer); wR rows are staged through LDS from the weight planes.
#include <algorithm>

#include "asw_device.h"
#include "asw_internal.h"

namespace {

constexpr uint32_t GEO_INF = 0x40000000u;


__global__ __launch_bounds__(256) void k_pack_bgrx(const uint8_t* __restrict__ bgr, size_t n, uint32_t* __restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)bgr[3 * i] | ((uint32_t)bgr[3 * i + 1] << 8) | ((uint32_t)bgr[3 * i + 2] << 16);
}

__device__ __forceinline__ uint32_t cdist(uint32_t a, uint32_t b) { return __builtin_amdgcn_sad_u8(a, b, 0u); }

// One thread = one pixel.  WIN is the (odd) window size; planes: [WIN*WIN][H][W] (u16 or f32).
// The backward sweep can only reach the cells above the centre and those left of it in its own row: every cell below the centre row
// (and right of the centre in it) still holds FLT_MAX when the forward sweep starts.  So the backward sweep runs rows h+1 .. 1
// only, and only those (h+1) * WIN intermediate values are kept for the forward sweep -- in LDS (u16, one column of 64 lanes
// per cell: 15 KB per wavefront at 15x15) when they fit in 16 KB, else in the output planes themselves.  Every plane is then
// written ONCE, after the forward sweep (the first version stored all 225 planes after the backward sweep, read them back and
// stored them again: 1.9 GB of traffic for 0.42 GB of tables at KITTI size).
template <int WIN, typename OutT>
__global__ __launch_bounds__(64) void k_geodesic_weights(const uint32_t* __restrict__ img, int H, int W, int backward,
                                                         int forward, OutT* __restrict__ planes)
{
    constexpr int h = WIN / 2, N = WIN + 2;
    constexpr int NKEEP = (h + 1) * WIN;
    constexpr bool KEEP_LDS = NKEEP * 64 * 2 <= 16 * 1024;
    __shared__ uint16_t skeep[KEEP_LDS ? NKEEP * 64 : 1];
    const int lane = threadIdx.x;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x;
    int col[N];  // image column of window column c (BORDER_REFLECT, pad h+1: M.cpp:1404)
#pragma unroll
    for (int c = 0; c < N; c++) col[c] = reflect_idx(x - h - 1 + c, W);

    auto store = [&](int r, int c, uint32_t v) {
        OutT o;
        if constexpr (sizeof(OutT) == 2) o = (OutT)(v >= GEO_INF ? 65535u : v);
        else o = v >= GEO_INF ? 3.402823466e+38f : (float)v;
        planes[(size_t)((r - 1) * WIN + (c - 1)) * plane + pix] = o;
    };
    // intermediate value of cell (r, c), r <= h + 1, between the sweeps (distances are below 65535 for windows up to 35x35)
    auto keep = [&](int r, int c, uint32_t v) {
        if constexpr (KEEP_LDS) skeep[((r - 1) * WIN + (c - 1)) * 64 + lane] = (uint16_t)(v >= GEO_INF ? 65535u : v);
        else store(r, c, v);
    };
    auto kept = [&](int r, int c) -> uint32_t {
        if constexpr (KEEP_LDS) {
            const uint32_t o = skeep[((r - 1) * WIN + (c - 1)) * 64 + lane];
            return o == 65535u ? GEO_INF : o;
        } else {
            const OutT o = planes[(size_t)((r - 1) * WIN + (c - 1)) * plane + pix];
            if constexpr (sizeof(OutT) == 2) return (o == 65535u) ? GEO_INF : (uint32_t)o;
            else return (o > 1e30f) ? GEO_INF : (uint32_t)o;
        }
    };

    // ---- backward sweep: r = h+1..1 (the rows below stay FLT_MAX), c = WIN..1; neighbours R, BR, B, BL (M.cpp:1367-1387) ----
    uint32_t dprev[N], pprev[N];  // distances / pixels of row r+1
#pragma unroll
    for (int c = 0; c < N; c++) { dprev[c] = GEO_INF; pprev[c] = 0u; }  // row h+2: all FLT_MAX, its pixels never matter
    if (backward) {
        for (int r = h + 1; r >= 1; r--) {
            uint32_t dcur[N], pcur[N];
            const uint32_t* row = img + (size_t)reflect_idx(y - h - 1 + r, H) * W;
#pragma unroll
            for (int c = 0; c < N; c++) pcur[c] = row[col[c]];
            dcur[N - 1] = GEO_INF;
            dcur[0] = GEO_INF;
#pragma unroll
            for (int c = WIN; c >= 1; c--) {
                uint32_t v = (r == h + 1 && c == h + 1) ? 0u : GEO_INF;  // M.cpp:1416-1417
                v = min(v, dcur[c + 1] + cdist(pcur[c + 1], pcur[c]));   // R
                v = min(v, dprev[c + 1] + cdist(pprev[c + 1], pcur[c])); // BR
                v = min(v, dprev[c] + cdist(pprev[c], pcur[c]));         // B
                v = min(v, dprev[c - 1] + cdist(pprev[c - 1], pcur[c])); // BL
                v = min(v, GEO_INF);                                      // FLT_MAX + d == FLT_MAX
                dcur[c] = v;
                if (forward) keep(r, c, v);
                else store(r, c, v);
            }
#pragma unroll
            for (int c = 0; c < N; c++) { dprev[c] = dcur[c]; pprev[c] = pcur[c]; }
        }
    } else {
        for (int r = h + 1; r >= 1; r--)
#pragma unroll
            for (int c = WIN; c >= 1; c--) {
                const uint32_t v = (r == h + 1 && c == h + 1) ? 0u : GEO_INF;
                if (forward) keep(r, c, v);
                else store(r, c, v);
            }
    }
    if (!forward) {  // fewer than three iterations: the rows below the centre keep their initial FLT_MAX
        for (int r = h + 2; r <= WIN; r++)
#pragma unroll
            for (int c = 1; c <= WIN; c++) store(r, c, GEO_INF);
        return;
    }

    // ---- forward sweep: r = 1..WIN, c = 1..WIN; neighbours L, UL, U, UR (M.cpp:1343-1362) ----
#pragma unroll
    for (int c = 0; c < N; c++) dprev[c] = GEO_INF;  // ring row 0
    {
        const uint32_t* row = img + (size_t)reflect_idx(y - h - 1, H) * W;
#pragma unroll
        for (int c = 0; c < N; c++) pprev[c] = row[col[c]];
    }
    for (int r = 1; r <= WIN; r++) {
        uint32_t dcur[N], pcur[N];
        const uint32_t* row = img + (size_t)reflect_idx(y - h - 1 + r, H) * W;
#pragma unroll
        for (int c = 0; c < N; c++) pcur[c] = row[col[c]];
        dcur[0] = GEO_INF;
        dcur[N - 1] = GEO_INF;
        const bool upper = r <= h + 1;  // wave-uniform
#pragma unroll
        for (int c = 1; c <= WIN; c++) {
            uint32_t v = upper ? kept(r, c) : GEO_INF;
            v = min(v, dcur[c - 1] + cdist(pcur[c - 1], pcur[c]));   // L
            v = min(v, dprev[c - 1] + cdist(pprev[c - 1], pcur[c])); // UL
            v = min(v, dprev[c] + cdist(pprev[c], pcur[c]));         // U
            v = min(v, dprev[c + 1] + cdist(pprev[c + 1], pcur[c])); // UR
            v = min(v, GEO_INF);
            dcur[c] = v;
            store(r, c, v);
        }
#pragma unroll
        for (int c = 0; c < N; c++) { dprev[c] = dcur[c]; pprev[c] = pcur[c]; }
    }
}

// planes [cells][H][W] f32 -> reference layout [H][W][win][win] (map<Point,Mat>, M.cpp:1420)
__global__ __launch_bounds__(256) void k_planes_to_windows(const float* __restrict__ planes, size_t npix, int cells,
                                                           float* __restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * cells) return;
    size_t p = i / cells;
    int c = (int)(i - p * cells);
    out[i] = planes[(size_t)c * npix + p];
}

// ---------------------------------------------------------------------------------------------
// aggregation
// ---------------------------------------------------------------------------------------------
constexpr int TW = 64, TH = 4, GG = 5;  // tile, taps per staging group; GDC = widest d-chunk of a kernel instance

struct GeoParams {
    int H, W, win, minD, nD;
    int flip;  // 1: mirrored problem (DISPARITY_RIGHT): images / weight planes are read at W-1-x, window columns reversed
    int cand_per_z;  // candidates per grid.z slice (multiple of 16): small frames split the d range over grid.z
    int c_begin;     // first candidate of this launch (> 0: the tail of a range whose head k_asw_geodesic_xq covers)
    int out_slice;   // >= 0 (grid.z == 1): winners go to slice out_slice of partE / partD (merged later) instead of disp
};

template <int DC, int GDC>
__device__ __forceinline__ void geo_chunk(const GeoParams& p, const uint32_t* __restrict__ imgR,
                                          const uint16_t* __restrict__ wL, const uint16_t* __restrict__ wR,
                                          float* __restrict__ vol, unsigned char* smem, int c0, double& bestE, float& bestD)
{
    const int win = p.win, h = win / 2, H = p.H, W = p.W;
    const int TR = TH + 2 * h, LW = TW + 2 * h, RWmax = TW + 2 * h + GDC - 1;
    constexpr int SWR = TW + DC - 1;
    // LDS carve-up (sized on the host for DC = GDC).  There is no cost tile: the colour-L1 cost of a tap and candidate is
    // one v_sad_u8 of two BGRX pixels read from the image tiles (a u16 cost tile was 45 KB at DC = 16 and held the
    // kernel at two wavefronts per SIMD).
    float* sWR = reinterpret_cast<float*>(smem);                                 // [2][GG][TH][TW+GDC-1]
    uint32_t* sL = reinterpret_cast<uint32_t*>(sWR + 2 * GG * TH * (TW + GDC - 1));  // [TR][LW]
    uint32_t* sR = sL + TR * LW;                                                 // [TR][RWmax]
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int d0 = p.minD + c0;
    const int RW = TW + 2 * h + DC - 1;
    const int sRx0 = x0 - h - (d0 + DC - 1);
    const size_t plane = (size_t)H * W;

    __syncthreads();
    for (int i = tid; i < TR * RW; i += 256) {
        int r = i / RW, c = i - r * RW;
        int yy = min(max(y0 - h + r, 0), H - 1), xx = min(max(sRx0 + c, 0), W - 1);
        sR[r * RWmax + c] = imgR[(size_t)yy * W + (p.flip ? W - 1 - xx : xx)];
    }
    // colour-L1 cost of tap (j,i) and candidate dd: |L(ny,nx) - R(ny, max(0,nx-d))|_1 (M.cpp:1490) with nx clamped to the
    // image.  Tile column of that right pixel: max(tcmin, tc0 - dd), tc0 = nx - d0 - sRx0.  Where nothing clamps (the
    // tile and its shifted window lie inside the image) this is tx + i + DC-1-dd: DC consecutive LDS words.
    const bool interior = sRx0 >= 0 && x0 - h >= 0 && x0 + TW + h <= W;
    const int tcmin = min(max(-sRx0, 0), RW - 1);
    double num[DC], den[DC];
#pragma unroll
    for (int dd = 0; dd < DC; dd++) { num[dd] = 0.0; den[dd] = 0.0; }
    const int x = x0 + tx, y = y0 + ty;
    const int xc = min(x, W - 1), yc = min(y, H - 1);
    const uint16_t* myWL = wL + (size_t)yc * W + (p.flip ? W - 1 - xc : xc);
    const uint32_t* myL = sL + ty * LW + tx;
    const uint32_t* myR = sR + ty * RWmax + tx;
    const float* myWR = sWR + ty * (TW + GDC - 1) + tx + (DC - 1);
    const int ntaps = win * win;

    // Staging positions of the right-image weights (each is used by up to DC (pixel,d) pairs): pass A = (row ty,
    // column tx), pass B = the DC-1 extra columns of the four rows (first TH*(DC-1) threads).  Column jj holds the
    // weight window of right pixel xr = max(0, x - d) with jj = tx + (DC-1) - dd.
    const int xrA = min(max(x0 - d0 - (DC - 1) + tx, 0), W - 1);
    const uint16_t* pA = wR + (size_t)yc * W + (p.flip ? W - 1 - xrA : xrA);
    float* dstA = sWR + ty * (TW + GDC - 1) + tx;
    constexpr int NEXTRA = TH * (DC - 1);
    const bool doB = (DC > 1) && tid < NEXTRA;
    const int rowB = (DC > 1) ? min(tid / (DC > 1 ? DC - 1 : 1), TH - 1) : 0;
    const int jjB = TW + tid - rowB * (DC - 1);
    const int xrB = min(max(x0 - d0 - (DC - 1) + jjB, 0), W - 1);
    const uint16_t* pB = wR + (size_t)min(y0 + rowB, H - 1) * W + (p.flip ? W - 1 - xrB : xrB);
    float* dstB = sWR + rowB * (TW + GDC - 1) + jjB;

    // window row / column of a tap (M.cpp:1481-1483), advanced without divisions; the mirrored problem reads the
    // weight planes with reversed window columns
    int js = 0, is = 0;  // staging cursor
    int j = 0, i = 0;    // accumulation cursor
    auto plane_of = [&](int tj, int ti) { return (size_t)(tj * win + (p.flip ? win - 1 - ti : ti)) * plane; };
    constexpr int GSTR = GG * TH * (TW + GDC - 1);  // floats per staging buffer (two buffers: software pipeline)
    float rA[GG], rB[GG], rW[GG];
    auto fetch_group = [&](int g0) {  // loads only: weights of the taps g0 .. g0+GG-1 into registers
        const int ng = min(GG, ntaps - g0);
#pragma unroll
        for (int tt = 0; tt < GG; tt++) {
            if (tt < ng) {
                const size_t po = plane_of(js, is);
                rA[tt] = (float)pA[po];
                rB[tt] = doB ? (float)pB[po] : 0.0f;
                rW[tt] = (float)myWL[po];  // this thread's own left-image weight
                if (++is == win) { is = 0; js++; }
            }
        }
    };
    auto store_group = [&](int buf) {
#pragma unroll
        for (int tt = 0; tt < GG; tt++) {
            dstA[buf * GSTR + tt * TH * (TW + GDC - 1)] = rA[tt];
            if (doB) dstB[buf * GSTR + tt * TH * (TW + GDC - 1)] = rB[tt];
        }
    };
    __syncthreads();  // cost tile complete, previous chunk's staging buffers free
    fetch_group(0);
    store_group(0);
    float wlv[GG];
#pragma unroll
    for (int tt = 0; tt < GG; tt++) wlv[tt] = rW[tt];
    __syncthreads();
    int buf = 0;
    for (int g0 = 0; g0 < ntaps; g0 += GG, buf ^= 1) {
        const int ng = min(GG, ntaps - g0);
        if (g0 + GG < ntaps) fetch_group(g0 + GG);  // next group's loads fly while this group is accumulated
#pragma unroll
        for (int tt = 0; tt < GG; tt++) {
            if (tt < ng) {
                const float wl = wlv[tt];
                const uint32_t pl = myL[j * LW + i];
                uint32_t pr[DC];
                if (interior) {
                    const uint32_t* q = myR + j * RWmax + i;
#pragma unroll
                    for (int dd = 0; dd < DC; dd++) pr[dd] = q[DC - 1 - dd];
                } else {
                    const uint32_t* q = sR + (ty + j) * RWmax;
                    const int tc0 = min(max(x0 - h + tx + i, 0), W - 1) - d0 - sRx0;
#pragma unroll
                    for (int dd = 0; dd < DC; dd++) pr[dd] = q[max(tcmin, tc0 - dd)];
                }
                const float* wr = myWR + buf * GSTR + tt * TH * (TW + GDC - 1);
#pragma unroll
                for (int dd = 0; dd < DC; dd++) {
                    float c = (float)cdist(pl, pr[dd]);
                    float ab = wl * wr[-dd];   // f32
                    float abc = ab * c;        // f32 (M.cpp:1488-1490)
                    num[dd] = num[dd] + (double)abc;
                    den[dd] = den[dd] + (double)ab;
                }
                if (++i == win) { i = 0; j++; }
            }
        }
        if (g0 + GG < ntaps) {
            store_group(buf ^ 1);  // that buffer was last read one iteration ago, before the barrier below
#pragma unroll
            for (int tt = 0; tt < GG; tt++) wlv[tt] = rW[tt];
        }
        __syncthreads();
    }
    if (x < W && y < H) {
#pragma unroll
        for (int dd = 0; dd < DC; dd++) {
            double E = num[dd] / den[dd];  // 0/0 -> NaN for windows flat in both images (App. B-9)
            if (vol) vol[((size_t)(c0 + dd) * H + y) * W + (p.flip ? W - 1 - x : x)] = (float)E;
            if (E < bestE) { bestE = E; bestD = (float)(d0 + dd); }
        }
    }
}

template <int GDC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_asw_geodesic(GeoParams p, const uint32_t* __restrict__ imgL,
                                                      const uint32_t* __restrict__ imgR, const uint16_t* __restrict__ wL,
                                                      const uint16_t* __restrict__ wR, float* __restrict__ vol,
                                                      float* __restrict__ disp, double* __restrict__ partE, float* __restrict__ partD)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int h = p.win / 2, TR = TH + 2 * h, LW = TW + 2 * h;
    float* sWR = reinterpret_cast<float*>(smem);
    uint32_t* sL = reinterpret_cast<uint32_t*>(sWR + 2 * GG * TH * (TW + GDC - 1));
    const int tid = threadIdx.x, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    for (int i = tid; i < TR * LW; i += 256) {
        int r = i / LW, c = i - r * LW;
        int yy = min(max(y0 - h + r, 0), p.H - 1), xx = min(max(x0 - h + c, 0), p.W - 1);
        sL[i] = imgL[(size_t)yy * p.W + (p.flip ? p.W - 1 - xx : xx)];
    }
    double bestE = 1.7976931348623157e308;
    float bestD = 0.0f;
    int c0 = p.c_begin + blockIdx.z * p.cand_per_z;  // this workgroup's candidate range: all of it, or one grid.z slice for small frames
    const int cEnd = min(p.nD, c0 + p.cand_per_z);
    if constexpr (GDC >= 16)
        for (; c0 + 16 <= cEnd; c0 += 16) geo_chunk<16, GDC>(p, imgR, wL, wR, vol, smem, c0, bestE, bestD);
    for (; c0 + 8 <= cEnd; c0 += 8) geo_chunk<8, GDC>(p, imgR, wL, wR, vol, smem, c0, bestE, bestD);
    if (cEnd - c0 >= 4) { geo_chunk<4, GDC>(p, imgR, wL, wR, vol, smem, c0, bestE, bestD); c0 += 4; }
    if (cEnd - c0 >= 2) { geo_chunk<2, GDC>(p, imgR, wL, wR, vol, smem, c0, bestE, bestD); c0 += 2; }
    if (cEnd - c0 >= 1) { geo_chunk<1, GDC>(p, imgR, wL, wR, vol, smem, c0, bestE, bestD); c0 += 1; }
    const int x = x0 + (tid & 63), y = y0 + (tid >> 6);
    if (x < p.W && y < p.H) {
        const size_t o = (size_t)y * p.W + (p.flip ? p.W - 1 - x : x);
        if (gridDim.z == 1 && p.out_slice < 0) {
            disp[o] = bestD;
        } else {  // per-slice winners, merged by launch_merge_slices
            const int z = gridDim.z == 1 ? p.out_slice : (int)blockIdx.z;
            partE[(size_t)z * p.H * p.W + o] = bestE;
            partD[(size_t)z * p.H * p.W + o] = bestD;
        }
    }
}

// A FEW candidates (the remainder of a range cut into xq passes: one candidate at the reference's numDisparity = 192, whose range
// is inclusive) on the un-mirrored problem: thread = one pixel, the 225 taps in a loop, both weight planes streamed once per
// candidate (coalesced u16 rows), the BGRX pixels from two LDS tiles.  The chunked kernel above spent 0.74 ms and 0.83 GB on that
// one candidate (a 16-wide chunk machinery around one lane of work); this is bound by reading the two 210 MB tables once.
//   fixed / other: the image the disparity map belongs to / the other one (LEFT: left / right), sgn = -1 (LEFT: q = max(0, x - d))
//   or +1 (RIGHT: q = min(W-1, x + d))                                                         (M.cpp:1467-1496, 1498-1520)
template <int WIN>
__global__ __launch_bounds__(256) void k_asw_geodesic_few(const uint32_t* __restrict__ imgF, const uint32_t* __restrict__ imgO,
                                                          const uint16_t* __restrict__ wF, const uint16_t* __restrict__ wO, int H, int W,
                                                          int minD, int c_begin, int c_end, int sgn, float* __restrict__ vol,
                                                          double* __restrict__ outE, float* __restrict__ outD)
{
    constexpr int h = WIN / 2, TR = TH + 2 * h, LW = TW + 2 * h;
    __shared__ uint32_t sF[TR * LW], sO[TR * LW];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    // every XCD takes a contiguous run of tiles (see k_asw_geodesic_xq): a tile row is 128 bytes of a u16 weight plane at an
    // arbitrary alignment, i.e. two 128-byte lines, each shared with a neighbour -- dealt round-robin, the neighbours sat on
    // other XCDs and every line came from HBM twice (0.82 GB fetched for 0.42 GB of tables)
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, vid = xcd * (nwg >> 3) + min(xcd, nwg & 7) + (lin >> 3);
    const int x0 = (vid % (int)gridDim.x) * TW, y0 = (vid / (int)gridDim.x) * TH;
    const int x = x0 + tx, y = y0 + ty, xc = min(x, W - 1), yc = min(y, H - 1);
    const size_t plane = (size_t)H * W;
    for (int i = tid; i < TR * LW; i += 256) {
        const int r = i / LW, c = i - r * LW;
        sF[i] = imgF[(size_t)min(max(y0 - h + r, 0), H - 1) * W + min(max(x0 - h + c, 0), W - 1)];
    }
    double bestE = 1.7976931348623157e308;
    float bestD = 0.0f;
    for (int cand = c_begin; cand < c_end; cand++) {
        const int d = minD + cand;
        __syncthreads();  // previous candidate's tile is dead (and sF is complete)
        for (int i = tid; i < TR * LW; i += 256) {
            const int r = i / LW, c = i - r * LW;
            // the other image's sample of window column nx = clamp(x0 - h + c): column max(0, nx - d) / min(W-1, nx + d)
            const int nx = min(max(x0 - h + c, 0), W - 1);
            const int ox = min(max(nx + sgn * d, 0), W - 1);
            sO[i] = imgO[(size_t)min(max(y0 - h + r, 0), H - 1) * W + ox];
        }
        __syncthreads();
        const int q = min(max(xc + sgn * d, 0), W - 1);
        const uint16_t* pf = wF + (size_t)yc * W + xc;
        const uint16_t* po = wO + (size_t)yc * W + q;
        double num = 0.0, den = 0.0;
        for (int j = 0; j < WIN; j++)
#pragma unroll
            for (int i = 0; i < WIN; i++) {
                const size_t cell = (size_t)(j * WIN + i) * plane;
                const float wl = (float)pf[cell], wr = (float)po[cell];
                const float c = (float)cdist(sF[(ty + j) * LW + tx + i], sO[(ty + j) * LW + tx + i]);
                const float ab = wl * wr;   // f32
                const float abc = ab * c;   // f32 (M.cpp:1488-1490)
                num = num + (double)abc;    // exact integers: any order
                den = den + (double)ab;
            }
        if (x < W && y < H) {
            const double E = num / den;  // 0/0 -> NaN for windows flat in both images (App. B-9)
            if (vol) vol[((size_t)cand * H + y) * W + x] = (float)E;
            if (E < bestE) { bestE = E; bestD = (float)d; }
        }
    }
    if (x < W && y < H) {
        outE[(size_t)y * W + x] = bestE;
        outD[(size_t)y * W + x] = bestD;
    }
}

template <int WIN, typename OutT>
int launch_weights_t(hipStream_t s, const uint32_t* img, int H, int W, int backward, int forward, OutT* planes)
{
    dim3 grid((W + 63) / 64, H);
    hipLaunchKernelGGL((k_geodesic_weights<WIN, OutT>), grid, dim3(64), 0, s, img, H, W, backward, forward, planes);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

template <typename OutT>
int launch_weights(hipStream_t s, const uint32_t* img, int H, int W, int win, int iter, OutT* planes)
{
    // iterations 0,1 -> backward, 2,3 -> forward, others nothing (iterCount/2, M.cpp:1341,1365)
    const int backward = iter >= 1, forward = iter >= 3;
    switch (win) {
#define GEO_CASE(w) case w: return launch_weights_t<w, OutT>(s, img, H, W, backward, forward, planes);
        GEO_CASE(1) GEO_CASE(3) GEO_CASE(5) GEO_CASE(7) GEO_CASE(9) GEO_CASE(11) GEO_CASE(13) GEO_CASE(15) GEO_CASE(17)
        GEO_CASE(19) GEO_CASE(21) GEO_CASE(23) GEO_CASE(25) GEO_CASE(27) GEO_CASE(29) GEO_CASE(31) GEO_CASE(33) GEO_CASE(35)
#undef GEO_CASE
    default: return ASW_ERR_BAD_ARGUMENT;  // windows above 35 are not instantiated
    }
}

}  // namespace

int launch_pack_bgrx(hipStream_t s, const uint8_t* bgr, int H, int W, uint32_t* out)
{
    size_t n = (size_t)H * W;
    hipLaunchKernelGGL(k_pack_bgrx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bgr, n, out);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

int launch_geodesic_weights_u16(hipStream_t s, const uint32_t* img, int H, int W, int win, int iter, uint16_t* planes)
{
    return launch_weights<uint16_t>(s, img, H, W, win, iter, planes);
}

int launch_geodesic_weights_f32(hipStream_t s, const uint32_t* img, int H, int W, int win, int iter, float* planes)
{
    return launch_weights<float>(s, img, H, W, win, iter, planes);
}

int launch_planes_to_windows(hipStream_t s, const float* planes, int H, int W, int cells, float* out)
{
    size_t n = (size_t)H * W * cells;
    hipLaunchKernelGGL(k_planes_to_windows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, planes, (size_t)H * W, cells, out);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

namespace {
template <int GDC>
size_t geo_lds_bytes(int win)
{
    const int h = win / 2, TR = TH + 2 * h, LW = TW + 2 * h, RWmax = TW + 2 * h + GDC - 1;
    return (size_t)2 * GG * TH * (TW + GDC - 1) * 4 + (size_t)TR * LW * 4 + (size_t)TR * RWmax * 4;
}
template <int GDC>
int launch_geo_t(hipStream_t s, GeoParams p, const uint32_t* imgL, const uint32_t* imgR, const uint16_t* wL,
                 const uint16_t* wR, float* vol, float* disp, double* partE, float* partD)
{
    const size_t lds = geo_lds_bytes<GDC>(p.win);
    auto kern = k_asw_geodesic<GDC>;
    if (lds > 64 * 1024)
        ASW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((p.W + TW - 1) / TW, (p.H + TH - 1) / TH);
    // frames with few tiles (KITTI: 1880 for 1024 resident workgroups) split the candidate range over grid.z in
    // multiples of 16 until there are ~4096 workgroups; a second tiny launch merges the per-slice winners
    const int tiles = grid.x * grid.y, chunks16 = (p.nD - p.c_begin + 15) / 16;
    int nz = 1;
    if (p.c_begin == 0 && p.out_slice < 0 && partE && partD && tiles < 4096) nz = std::min(chunks16, std::min(8, (4096 + tiles - 1) / tiles));
    const int chunks_per_z = (chunks16 + nz - 1) / nz;
    nz = (chunks16 + chunks_per_z - 1) / chunks_per_z;
    p.cand_per_z = chunks_per_z * 16;
    grid.z = nz;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p, imgL, imgR, wL, wR, vol, disp, partE, partD);
    ASW_HIP_TRY(hipGetLastError());
    if (nz > 1) return launch_merge_slices(s, partE, partD, nz, (size_t)p.H * p.W, disp);
    return ASW_OK;
}
}  // namespace

int launch_asw_geodesic(hipStream_t s, const uint32_t* imgL, const uint32_t* imgR, const uint16_t* wL, const uint16_t* wR,
                        int H, int W, int win, int minD, int nD, int flip, float* vol, float* disp, double* partE, float* partD,
                        int c_begin, int out_slice)
{
    if (c_begin < 0 || c_begin >= nD || (out_slice >= 0 && !(partE && partD))) return ASW_ERR_BAD_ARGUMENT;
    GeoParams p{H, W, win, minD, nD, flip, 0, c_begin, out_slice};
    // 16-wide d-chunks: 25 KB of LDS at win 15 (45 KB at win 35)
    if (geo_lds_bytes<16>(win) <= 160 * 1024) return launch_geo_t<16>(s, p, imgL, imgR, wL, wR, vol, disp, partE, partD);
    return ASW_ERR_BAD_ARGUMENT;
}

// candidates [c_begin, nD) of a 15x15 problem, winners to outE / outD ([H][W]); imgF / wF: the fixed image and its weight planes
int launch_asw_geodesic_few(hipStream_t s, const uint32_t* imgF, const uint32_t* imgO, const uint16_t* wF, const uint16_t* wO, int H,
                            int W, int minD, int c_begin, int nD, bool right, float* vol, double* outE, float* outD)
{
    if (c_begin < 0 || c_begin >= nD || !outE || !outD) return ASW_ERR_BAD_ARGUMENT;
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH);
    hipLaunchKernelGGL(k_asw_geodesic_few<15>, grid, dim3(256), 0, s, imgF, imgO, wF, wO, H, W, minD, c_begin, nD, right ? 1 : -1, vol,
                       outE, outD);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}
