// Guided-filter aggregation (getGuidedFilter, M.cpp:2766-2854) and the box-mean machinery it is made of.
//
// boxFilter(CV_32F, Size(k,k), normalised, BORDER_REFLECT_101) == f64 window sum * 1/(k*k) -> f32
// (SURVEY App. A-9).  One generic "column walk" kernel evaluates NP box means at once:
//   * a 256-thread block owns 256 input columns (256-(k-1) output columns) of a band of rows and walks
//     down the band; every thread keeps NP vertical running sums in f64 (add the entering row, subtract
//     the leaving one -- the same sliding form as OpenCV's ColumnSum), the leaving row's values come
//     from a per-thread ring in LDS (or are recomputed when the ring would not fit);
//   * per output row the vertical sums go through LDS (double-buffered, one barrier per row) and each
//     output column adds its k neighbours in ascending order (conflict-free ds_read_b64);
//   * the producer (Src) and consumer (Dst) are functors, so normalisation, products, covariance,
//     a = cov/(var+eps), b and q are fused into the filters that need them and never hit HBM as
//     separate planes.
// The kernels are bound by f64 VALU + HBM streaming of the a/b planes, see DESIGN.md.
#include "asw_internal.h"

namespace {

__device__ __forceinline__ int reflect_idx(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p - 1 : 2 * len - 1 - p;
    return p;
}
__device__ __forceinline__ int reflect101_idx(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

constexpr int BW = 256;  // block width = input columns per block

template <int NP, class Src, class Dst>
__global__ __launch_bounds__(BW) void k_box_walk(Src src, Dst dst, int H, int W, int k, int band, int use_ring)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double* hs = reinterpret_cast<double*>(smem);                               // [2][NP][BW]
    float* ring = reinterpret_cast<float*>(smem + (size_t)2 * NP * BW * sizeof(double));  // [k][BW][NP]
    const int t = threadIdx.x, kz = blockIdx.z;
    const int hl = k / 2;  // OpenCV anchor = k/2 (also for even k)
    const int XO = BW - (k - 1);
    const int xo0 = blockIdx.x * XO;
    const int xin = reflect101_idx(xo0 - hl + t, W);
    const int y0 = blockIdx.y * band, y1 = min(H, y0 + band);
    const double scale = 1.0 / ((double)k * (double)k);
    const bool out_thread = (t < XO) && (xo0 + t < W);
    double vs[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) vs[p] = 0.0;

    const int steps = (y1 - y0) + k - 1;
    for (int s = 0; s < steps; s++) {
        const int yy = reflect101_idx(y0 - hl + s, H);
        float v[NP];
        src(yy, xin, kz, v);
        float* slot = ring + ((size_t)(s % k) * BW + t) * NP;
        if (s >= k) {  // the row leaving the window (ColumnSum: SUM -= Sm)
            float o[NP];
            if (use_ring) {
#pragma unroll
                for (int p = 0; p < NP; p++) o[p] = slot[p];
            } else {
                src(reflect101_idx(y0 - hl + s - k, H), xin, kz, o);
            }
#pragma unroll
            for (int p = 0; p < NP; p++) vs[p] = vs[p] - (double)o[p];
        }
#pragma unroll
        for (int p = 0; p < NP; p++) vs[p] = vs[p] + (double)v[p];
        if (use_ring) {
#pragma unroll
            for (int p = 0; p < NP; p++) slot[p] = v[p];
        }
        if (s >= k - 1) {
            double* buf = hs + (size_t)(s & 1) * NP * BW;
#pragma unroll
            for (int p = 0; p < NP; p++) buf[p * BW + t] = vs[p];
            __syncthreads();
            if (out_thread) {
                float m[NP];
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    const double* b = buf + p * BW + t;
                    double sum = 0.0;
                    for (int i = 0; i < k; i++) sum = sum + b[i];
                    m[p] = (float)(sum * scale);
                }
                dst(y0 + s - (k - 1), xo0 + t, kz, m);
            }
        }
    }
}

// ---- guide access: normalised guide channels I_c(y,x) for slice k --------------------------------
// mode 0: 3 channels from A.  mode 1: channels 0-2 from A, 3-5 from B shifted by the slice's disparity
// through the REFLECT pad (M.cpp:2907-2912).  mode 2: C interleaved channels in A (public getGuidedFilter).
struct GuideAcc {
    const uint8_t* A;
    const uint8_t* B;
    const float2* scales;  // normalize() scale/shift per slice (index k * scale_stride)
    int scale_stride;
    int W, mode, minD, C;
    template <int C0, int N>
    __device__ __forceinline__ void load(int y, int x, int k, float (&I)[N]) const
    {
        const float2 sc = scales[k * scale_stride];
#pragma unroll
        for (int c = 0; c < N; c++) {
            const int ch = C0 + c;
            int u;
            if (mode == 2) u = A[((size_t)y * W + x) * C + ch];
            else if (ch < 3) u = A[((size_t)y * W + x) * 3 + ch];
            else u = B[((size_t)y * W + reflect_idx(x - (minD + k), W)) * 3 + (ch - 3)];
            I[c] = (float)u * sc.x + sc.y;  // convertTo 8u->32f with float scale/shift (App. A-10)
        }
    }
};

// box(I_c), box(I_c*I_c) -> meanI_c, den_c = (corrI_c - meanI_c^2) + eps      (M.cpp:2778, 2796-2799, 2846)
template <int C0>
struct StatsSrc {
    GuideAcc g;
    __device__ __forceinline__ void operator()(int y, int x, int k, float (&v)[6]) const
    {
        float I[3];
        g.template load<C0, 3>(y, x, k, I);
#pragma unroll
        for (int c = 0; c < 3; c++) { v[c] = I[c]; v[3 + c] = I[c] * I[c]; }
    }
};
template <int C0>
struct StatsDst {
    float* meanI;  // [kslot][C][H][W]
    float* den;
    int H, W, C;
    float epsf;
    __device__ __forceinline__ void operator()(int y, int x, int k, const float (&m)[6]) const
    {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            size_t o = (((size_t)k * C + C0 + c) * H + y) * W + x;
            float mm = m[c] * m[c];
            float var = m[3 + c] - mm;
            meanI[o] = m[c];
            den[o] = 1.0f * epsf + var;  // scaleAdd(ones, eps, var)
        }
    }
};

// box(P), box(I_c*P) -> a_c = cov_c / den_c, b = meanP - sum_c a_c*meanI_c      (M.cpp:2780-2847)
template <int C>
struct ABSrc {
    GuideAcc g;
    const float* P;          // raw cost volume [n][H][W]
    const float2* pscales;   // per-slice normalize() parameters
    int H, W;
    __device__ __forceinline__ void operator()(int y, int x, int k, float (&v)[C + 1]) const
    {
        float I[C];
        g.template load<0, C>(y, x, k, I);
        const float2 sc = pscales[k];
        float p = P[((size_t)k * H + y) * W + x] * sc.x + sc.y;
        v[0] = p;
#pragma unroll
        for (int c = 0; c < C; c++) v[1 + c] = I[c] * p;
    }
};
template <int C>
struct ABDst {
    const float* meanI;
    const float* den;
    float* ab;  // [n][C+1][H][W]
    int H, W, stat_stride;  // stat_stride: 1 when the guide statistics depend on the slice, else 0
    __device__ __forceinline__ void operator()(int y, int x, int k, const float (&m)[C + 1]) const
    {
        const float meanP = m[0];
        float dot = 0.0f;
        const int ks = k * stat_stride;
#pragma unroll
        for (int c = 0; c < C; c++) {
            size_t so = (((size_t)ks * C + c) * H + y) * W + x;
            float mI = meanI[so];
            float mp = mI * meanP;
            float cov = m[1 + c] - mp;
            float ac = cov / den[so];
            ab[(((size_t)k * (C + 1) + c) * H + y) * W + x] = ac;
            float pr = ac * mI;
            dot = (c == 0) ? pr : dot + pr;  // operator*(Vec,Vec): left to right (M.cpp:22-31)
        }
        ab[(((size_t)k * (C + 1) + C) * H + y) * W + x] = meanP - dot;
    }
};

// box(a_c), box(b) -> q = sum_c box(a_c)*I_c + box(b)                           (M.cpp:2849-2852)
template <int C>
struct QSrc {
    const float* ab;
    int H, W;
    __device__ __forceinline__ void operator()(int y, int x, int k, float (&v)[C + 1]) const
    {
#pragma unroll
        for (int c = 0; c < C + 1; c++) v[c] = ab[(((size_t)k * (C + 1) + c) * H + y) * W + x];
    }
};
template <int C>
struct QDst {
    GuideAcc g;
    float* q;  // [n][H][W]
    int H, W;
    __device__ __forceinline__ void operator()(int y, int x, int k, const float (&m)[C + 1]) const
    {
        float I[C];
        g.template load<0, C>(y, x, k, I);
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < C; c++) {
            float pr = m[c] * I[c];
            dot = (c == 0) ? pr : dot + pr;
        }
        q[((size_t)k * H + y) * W + x] = dot + m[C];
    }
};

// getCostSAD_d (M.cpp:2442-2503): |grayL - grayR shifted| as f32, box mean
struct SadSrc {
    const uint8_t* gl;
    const uint8_t* gr;
    int W, minD, disp_type;
    __device__ __forceinline__ void operator()(int y, int x, int k, float (&v)[1]) const
    {
        const int d = minD + k;
        int a, b;
        if (disp_type == ASW_DISPARITY_LEFT) {
            a = gl[(size_t)y * W + x];
            b = gr[(size_t)y * W + reflect_idx(x - d, W)];
        } else {
            a = gl[(size_t)y * W + reflect_idx(x + d, W)];
            b = gr[(size_t)y * W + x];
        }
        v[0] = (float)abs(a - b);
    }
};
struct PlaneDst {
    float* out;
    int H, W;
    __device__ __forceinline__ void operator()(int y, int x, int k, const float (&m)[1]) const
    {
        out[((size_t)k * H + y) * W + x] = m[0];
    }
};
struct PlaneSrc {
    const float* in;
    int H, W;
    __device__ __forceinline__ void operator()(int y, int x, int k, float (&v)[1]) const
    {
        v[0] = in[((size_t)k * H + y) * W + x];
    }
};

template <int NP, class Src, class Dst>
int launch_walk(hipStream_t s, const Src& src, const Dst& dst, int H, int W, int k, int n)
{
    if (k < 1 || k > BW - 32) return ASW_ERR_BAD_ARGUMENT;
    const int XO = BW - (k - 1);
    int band = 64;
    if (band < 2 * k) band = 2 * k;  // keep the warm-up overhead (k-1 rows per band) below ~50 %
    size_t hs_bytes = (size_t)2 * NP * BW * sizeof(double);
    size_t ring_bytes = (size_t)k * BW * NP * sizeof(float);
    int use_ring = (hs_bytes + ring_bytes <= 64 * 1024) ? 1 : 0;
    auto kern = k_box_walk<NP, Src, Dst>;
    size_t lds = hs_bytes + (use_ring ? ring_bytes : 0);
    if (!use_ring && hs_bytes + ring_bytes <= 150 * 1024) {
        // a bigger ring still beats recomputing the leaving row; ask for the larger LDS carve-out
        use_ring = 1;
        lds = hs_bytes + ring_bytes;
        ASW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    dim3 grid((W + XO - 1) / XO, (H + band - 1) / band, n);
    hipLaunchKernelGGL(kern, grid, dim3(BW), lds, s, src, dst, H, W, k, band, use_ring);
    ASW_HIP_TRY(hipGetLastError());
    return ASW_OK;
}

}  // namespace

int launch_box_filter(hipStream_t s, const float* in, float* out, int n, int H, int W, int k)
{
    PlaneSrc src{in, H, W};
    PlaneDst dst{out, H, W};
    return launch_walk<1>(s, src, dst, H, W, k, n);
}

int launch_cost_sad(hipStream_t s, const uint8_t* gl, const uint8_t* gr, int H, int W, int disp_type, int win, int minD,
                    int numD, float* cost)
{
    SadSrc src{gl, gr, W, minD, disp_type};
    PlaneDst dst{cost, H, W};
    return launch_walk<1>(s, src, dst, H, W, win, numD);
}

int launch_guided(hipStream_t s, const GuidedLaunch& a)
{
    GuideAcc g;
    g.A = a.guideA; g.B = a.guideB; g.scales = a.gscales; g.scale_stride = a.guide_per_slice ? 1 : 0;
    g.W = a.W; g.mode = a.mode; g.minD = a.minD; g.C = a.C;
    const int nstat = a.guide_per_slice ? a.n : 1;
    const float epsf = (float)a.eps;
    int rc;
    // 1. guide statistics: meanI_c, den_c (once when the guide does not depend on the slice)
    {
        StatsSrc<0> src{g};
        StatsDst<0> dst{a.meanI, a.den, a.H, a.W, a.C, epsf};
        rc = launch_walk<6>(s, src, dst, a.H, a.W, a.r, nstat);
        if (rc != ASW_OK) return rc;
        if (a.C == 6) {
            StatsSrc<3> src2{g};
            StatsDst<3> dst2{a.meanI, a.den, a.H, a.W, a.C, epsf};
            rc = launch_walk<6>(s, src2, dst2, a.H, a.W, a.r, nstat);
            if (rc != ASW_OK) return rc;
        }
    }
    // 2. a, b    3. q
    if (a.C == 3) {
        ABSrc<3> src{g, a.P, a.pscales, a.H, a.W};
        ABDst<3> dst{a.meanI, a.den, a.ab, a.H, a.W, a.guide_per_slice ? 1 : 0};
        rc = launch_walk<4>(s, src, dst, a.H, a.W, a.r, a.n);
        if (rc != ASW_OK) return rc;
        QSrc<3> qs{a.ab, a.H, a.W};
        QDst<3> qd{g, a.q, a.H, a.W};
        return launch_walk<4>(s, qs, qd, a.H, a.W, a.r, a.n);
    } else {
        ABSrc<6> src{g, a.P, a.pscales, a.H, a.W};
        ABDst<6> dst{a.meanI, a.den, a.ab, a.H, a.W, a.guide_per_slice ? 1 : 0};
        rc = launch_walk<7>(s, src, dst, a.H, a.W, a.r, a.n);
        if (rc != ASW_OK) return rc;
        QSrc<6> qs{a.ab, a.H, a.W};
        QDst<6> qd{g, a.q, a.H, a.W};
        return launch_walk<7>(s, qs, qd, a.H, a.W, a.r, a.n);
    }
}
