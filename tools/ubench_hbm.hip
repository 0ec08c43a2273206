// Micro-benchmark: streaming write / read / copy bandwidth of one MI355X with 16-byte-per-lane accesses, and the a/b-store
// pattern of the guided filter (two interleaved 16-byte stores per lane pair).  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm.hip -o tools/ubench_hbm
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k_write(float4* out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ __launch_bounds__(256) void k_write_nt(float4* out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    typedef float v4f __attribute__((ext_vector_type(4)));
    for (; i < n; i += stride) { v4f v = {1.f, 2.f, 3.f, (float)i}; __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(out) + i); }
}
// lane l writes elements 2l and 2l+1 of its wave's 128-element chunk with two store instructions (the CPL = 2 emit)
__global__ __launch_bounds__(256) void k_write_pair(float4* out, size_t n)
{
    const int lane = threadIdx.x & 63;
    size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (; w * 128 + 127 < n; w += nw) {
        out[w * 128 + 2 * lane] = make_float4(1.f, 2.f, 3.f, (float)w);
        out[w * 128 + 2 * lane + 1] = make_float4(1.f, 2.f, 3.f, (float)w);
    }
}
__global__ __launch_bounds__(256) void k_read(const float4* in, size_t n, float* sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) *sink = acc;
}
__global__ __launch_bounds__(256) void k_copy(const float4* in, float4* out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

int main()
{
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    float4 *a, *b;
    float* sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 32768}) {
        for (int which = 0; which < 5; which++) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                switch (which) {
                case 0: hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, a, n); break;
                case 1: hipLaunchKernelGGL(k_write_nt, dim3(blocks), dim3(256), 0, 0, a, n); break;
                case 2: hipLaunchKernelGGL(k_write_pair, dim3(blocks), dim3(256), 0, 0, a, n); break;
                case 3: hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, sink); break;
                case 4: hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); break;
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const char* names[] = {"write 16B/lane", "write nontemporal", "write pair-interleaved", "read 16B/lane", "copy (read+write)"};
            const double moved = which == 4 ? 2.0 * bytes : (double)bytes;
            printf("blocks %6d  %-24s %7.3f ms  %6.2f TB/s\n", blocks, names[which], best, moved / best * 1e-9);
        }
    }
    return 0;
}
