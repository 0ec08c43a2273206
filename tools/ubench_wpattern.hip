// Micro-benchmark: HBM write bandwidth of the a/b pass's store pattern.  Every wavefront writes one dense chunk per "row" and
// plane; the chunks of a wavefront are either contiguous in memory (tile layout) or a row stride apart (image layout), the
// wavefronts of the chip are spread over slices as in the real launch.  A VALU delay loop paces the stores.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_wpattern.hip -o tools/ubench_wpattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// grid: one wavefront per (strip, band, slice); 4 wavefronts (4 slices) per workgroup
__global__ __launch_bounds__(256) void k_pat(float4* out, int nstrips, int nbands, int nslices, int rows, int chunk16 /* float4 per chunk: <= 64 */,
                                             size_t row_stride16, size_t strip_stride16, size_t band_stride16, size_t slice_stride16,
                                             size_t plane_stride16, int delay)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wj = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int nzg = nslices / 4;
    const int nreg = nstrips * nbands, rpx = (nreg + 7) / 8;
    const int reg = xcd * rpx + wj / nzg;
    if (wj / nzg >= rpx || reg >= nreg) return;
    const int strip = reg % nstrips, band = reg / nstrips, slice = (wj % nzg) * 4 + wv;
    float4* base = out + slice * slice_stride16 + band * band_stride16 + strip * strip_stride16;
    float acc = (float)lane;
    for (int r = 0; r < rows; r++) {
        for (int i = 0; i < delay; i++) acc = acc * 1.0001f + 0.5f;
        if (lane < chunk16) {
            base[r * row_stride16 + lane] = make_float4(acc, 1.f, 2.f, 3.f);
            base[plane_stride16 + r * row_stride16 + lane] = make_float4(acc, 4.f, 5.f, 6.f);
        }
    }
    if (acc == 1.2345f) lds[0] = acc;
}

int main(int argc, char** argv)
{
    const int H = 1080, W = 1920, nslices = 128, nstrips = 17, nbands = 8, rows = 135;
    const int chunk16 = 57;  // 114 columns x 8 B = 912 B
    const size_t plane16 = (size_t)H * W * 8 / 16;
    const size_t bytes = (size_t)nslices * 2 * plane16 * 16;
    float4* buf;
    if (hipMalloc(&buf, bytes + (64 << 20)) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int nwg = ((nstrips * nbands + 7) / 8) * 8 * (nslices / 4);
    const double total = (double)nstrips * nbands * nslices * rows * chunk16 * 16 * 2;
    for (int lds_kb : {80, 40}) {  // 80 KB per workgroup: 2 workgroups = 8 wavefronts per CU; 40 KB: 16
        for (int delay : {0, 200, 400}) {
            for (int layout = 0; layout < 3; layout++) {
                size_t row_s, strip_s, band_s, slice_s, plane_s;
                const char* name;
                if (layout == 0) {  // image layout: [slice][plane][H][W]
                    row_s = (size_t)W * 8 / 16; strip_s = chunk16; band_s = (size_t)rows * row_s; slice_s = 2 * plane16; plane_s = plane16;
                    name = "image rows (row stride 15 KB)";
                } else if (layout == 1) {  // tile layout: a wavefront's rows are contiguous
                    row_s = chunk16; strip_s = (size_t)rows * chunk16; band_s = (size_t)nstrips * strip_s; slice_s = 2 * plane16; plane_s = plane16;
                    name = "tiles (rows of a wavefront contiguous)";
                } else {  // tile layout, both planes of a row adjacent
                    row_s = 2 * chunk16; strip_s = (size_t)rows * row_s; band_s = (size_t)nstrips * strip_s; slice_s = 2 * plane16; plane_s = chunk16;
                    name = "tiles, planes interleaved per row";
                }
                float best = 1e9f;
                for (int rep = 0; rep < 3; rep++) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k_pat, dim3(nwg), dim3(256), lds_kb * 1024, 0, buf, nstrips, nbands, nslices, rows, chunk16, row_s, strip_s,
                                       band_s, slice_s, plane_s, delay);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("waves/CU %2d  delay %3d  %-40s %7.3f ms  %5.2f TB/s\n", lds_kb == 80 ? 8 : 16, delay, name, best, total / best * 1e-9);
            }
        }
    }
    return 0;
}
