// Device-side helpers shared by the kernels: OpenCV border index maps.
#pragma once
#include <hip/hip_runtime.h>

// BORDER_REFLECT: fedcba|abcdefgh|hgfedcb -- copyMakeBorder(..., BORDER_REFLECT) of the reference
__device__ __forceinline__ int reflect_idx(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p - 1 : 2 * len - 1 - p;
    return p;
}

// BORDER_REFLECT_101 (App. A-3): gfedcb|abcdefgh|gfedcba -- the default border of boxFilter / filter2D
__device__ __forceinline__ int reflect101_idx(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
