// Micro-benchmark: per-instruction VALU issue cost on gfx950 for the ops the ASW kernels lean on.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define ITERS 4096
#define COMMA ,
#define REP8(x) x x x x x x x x

#define KERNEL(name, decl, body, sink)                                           \
    __global__ __launch_bounds__(256) void name(float* out, int n)               \
    {                                                                            \
        decl;                                                                    \
        for (int i = 0; i < n; i++) {                                            \
            REP8(body)                                                           \
        }                                                                        \
        sink;                                                                    \
    }

// 4 independent chains, 2 ops each per REP -> 8*8 = 64 instrs per iteration (declared per kernel)
KERNEL(k_fma_f32, float a0 = threadIdx.x; float a1 = a0 + 1; float a2 = a0 + 2; float a3 = a0 + 3; float b = 1.0001f;,
       asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n"
                    "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3)

KERNEL(k_fma_f64, double a0 = threadIdx.x; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 1.0001;,
       asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                    "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3))

KERNEL(k_add_f64, double a0 = threadIdx.x; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 1.0001;,
       asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                    "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3))

KERNEL(k_mul_f64, double a0 = threadIdx.x; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 1.0001;,
       asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                    "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3))

KERNEL(k_cvt_f64_f32, double a0 = 0; double a1 = 0; double a2 = 0; double a3 = 0; float b = threadIdx.x;,
       asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4\n"
                    "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3))

KERNEL(k_cvt_f64_u32, double a0 = 0; double a1 = 0; double a2 = 0; double a3 = 0; unsigned b = threadIdx.x;,
       asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %4\n v_cvt_f64_u32 %2, %4\n v_cvt_f64_u32 %3, %4\n"
                    "v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %4\n v_cvt_f64_u32 %2, %4\n v_cvt_f64_u32 %3, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3))

KERNEL(k_cvt_f32_ubyte, float a0 = 0; float a1 = 0; float a2 = 0; float a3 = 0; unsigned b = threadIdx.x * 0x01010101u;,
       asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n"
                    "v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3)

KERNEL(k_and_shift, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 0x12345678u;,
       asm volatile("v_and_b32 %0, 0xff, %4\n v_lshrrev_b32 %1, 24, %4\n v_and_b32 %2, 0xff, %4\n v_lshrrev_b32 %3, 24, %4\n"
                    "v_and_b32 %0, 0xff, %4\n v_lshrrev_b32 %1, 24, %4\n v_and_b32 %2, 0xff, %4\n v_lshrrev_b32 %3, 24, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3))

KERNEL(k_minmax_u32, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 0x12345678u;,
       asm volatile("v_min_u32 %0, %0, %4\n v_max_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_max_u32 %3, %3, %4\n"
                    "v_min_u32 %0, %0, %4\n v_max_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_max_u32 %3, %3, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3))

KERNEL(k_cvt_ubyte8, float a0 = 0; float a1 = 0; float a2 = 0; float a3 = 0; unsigned b = threadIdx.x * 0x01010101u;,
       asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n"
                    "v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3)

KERNEL(k_mul_f32, float a0 = threadIdx.x; float a1 = a0 + 1; float a2 = a0 + 2; float a3 = a0 + 3; float b = 1.0001f;,
       asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                    "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3)

KERNEL(k_pk_mul_f32, float2 a0 = make_float2(1 COMMA 2); float2 a1 = make_float2(3 COMMA 4); float2 a2 = make_float2(5 COMMA 6); float2 a3 = make_float2(7 COMMA 8); float2 b = make_float2(1.0001f COMMA 1.0002f);,
       asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                    "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.x + a2.y + a3.y)

KERNEL(k_bfe_u32, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 0x12345678u;,
       asm volatile("v_bfe_u32 %0, %4, 8, 8\n v_bfe_u32 %1, %4, 8, 8\n v_bfe_u32 %2, %4, 16, 8\n v_bfe_u32 %3, %4, 16, 8\n"
                    "v_bfe_u32 %0, %4, 8, 8\n v_bfe_u32 %1, %4, 8, 8\n v_bfe_u32 %2, %4, 16, 8\n v_bfe_u32 %3, %4, 16, 8\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3))

// ONE dependent chain per wave (every instruction reads the previous result): exposes the result latency that the 4-chain
// kernels above hide
KERNEL(k_add_f64_dep, double a0 = threadIdx.x; double b = 1.0001;,
       asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                    "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                    : "+v"(a0) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = (float)a0)
KERNEL(k_add_f64_dep2, double a0 = threadIdx.x; double a1 = a0 + 1; double b = 1.0001;,
       asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n"
                    "v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2\n"
                    : "+v"(a0), "+v"(a1) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1))
KERNEL(k_add_f32_dep, float a0 = threadIdx.x; float b = 1.0001f;,
       asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                    "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                    : "+v"(a0) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = a0)
KERNEL(k_cvt_add_dep, double a0 = threadIdx.x; double t = 0; float b = 1.0001f;,
       asm volatile("v_cvt_f64_f32 %1, %2\n v_add_f64 %0, %0, %1\n v_cvt_f64_f32 %1, %2\n v_add_f64 %0, %0, %1\n"
                    "v_cvt_f64_f32 %1, %2\n v_add_f64 %0, %0, %1\n v_cvt_f64_f32 %1, %2\n v_add_f64 %0, %0, %1\n"
                    : "+v"(a0), "+v"(t) : "v"(b));,
       out[blockIdx.x * blockDim.x + threadIdx.x] = (float)a0)

typedef void (*kern_t)(float*, int);

int main()
{
    float* out;
    hipMalloc(&out, 1 << 24);
    struct { const char* name; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma_f32}, {"v_mul_f32", k_mul_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_bfe_u32", k_bfe_u32},
        {"v_cvt_f32_ubyteN (x1.5)", k_cvt_f32_ubyte}, {"v_cvt_f32_ubyteN", k_cvt_ubyte8}, {"v_and/lshrrev_b32", k_and_shift}, {"v_min/max_u32", k_minmax_u32}, {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64},
        {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f64_u32", k_cvt_f64_u32},
        {"v_add_f64 1 chain", k_add_f64_dep}, {"v_add_f64 2 chains", k_add_f64_dep2}, {"v_add_f32 1 chain", k_add_f32_dep},
        {"cvt_f64_f32+add_f64 chain", k_cvt_add_dep}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wavesPerSimd : {1, 2, 3, 4}) {
        int blocks = 256 * wavesPerSimd;  // 256-thread blocks: 4 waves each -> wavesPerSimd waves per SIMD
        printf("--- %d wave(s) per SIMD (grid %d x 256)\n", wavesPerSimd, blocks);
        for (auto& k : ks) {
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 16);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, ITERS);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double instr_per_wave = (double)ITERS * 64;
            // each SIMD runs wavesPerSimd waves; cycles per wave-instruction at an assumed 2.4 GHz
            double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * wavesPerSimd);
            printf("%-18s %8.3f ms   %.2f cycles/wave-instr (@2.4GHz)\n", k.name, ms, cyc);
        }
    }
    return 0;
}
