// Stable in-wavefront sorting network shared by the weighted-median kernels (k_wmedian.hip, k_wmedian_tile.hip).
// Keys carry their own tie-break (window / region index in the low bits), so a plain bitonic network sorts stably.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace wmnet {

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    // every control used here is a permutation (each lane has a source), so no 'old' value: no v_mov 0 in front of each move
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}

// value of lane (lane ^ LM)
template <int LM>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v)
{
    if constexpr (LM == 1) return dpp_mov<0xB1>(v);         // quad_perm [1,0,3,2]
    else if constexpr (LM == 2) return dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    else if constexpr (LM == 3) return dpp_mov<0x1B>(v);    // quad_perm [3,2,1,0]
    else if constexpr (LM == 7) return dpp_mov<0x141>(v);   // row_half_mirror
    else if constexpr (LM == 15) return dpp_mov<0x140>(v);  // row_mirror
    else if constexpr (LM == 8) return dpp_mov<0x128>(v);   // row_ror:8 == xor 8 inside a 16-lane row
    else if constexpr (LM == 4 || LM == 16 || LM == 31) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (LM << 10) | 0x1f);
    else return (uint32_t)__shfl_xor((int)v, LM);
}

// Sorting network: merges of size K = 2, 4, ..., 256 over element e = lane*4 + r.  Every merge starts with a
// MIRROR step (partner e ^ (K-1)) and continues with half-cleaners (partner e ^ j): all comparisons then sort
// ascending -- the lower index takes the minimum -- so intra-lane exchanges are a bare v_min/v_max pair and
// cross-lane ones select with one of six loop-invariant lane masks (bit 0..5 of the lane id).
struct LaneMasks { uint32_t m[6]; };  // m[b] = all ones where bit b of the lane id is clear (loop-invariant VGPRs)

template <int LM>
__device__ __forceinline__ unsigned long long lane_xor(unsigned long long v)
{
    const uint32_t lo = lane_xor<LM>((uint32_t)v), hi = lane_xor<LM>((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// f64 keys (exact integers below 2^53): the compare-exchange is a v_min_f64 / v_max_f64 pair
template <int LM>
__device__ __forceinline__ double lane_xor(double v)
{
    return __longlong_as_double((long long)lane_xor<LM>((unsigned long long)__double_as_longlong(v)));
}

__device__ __forceinline__ uint32_t kmin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t kmax(uint32_t a, uint32_t b) { return a < b ? b : a; }
__device__ __forceinline__ unsigned long long kmin(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned long long kmax(unsigned long long a, unsigned long long b) { return a < b ? b : a; }
// keys are never NaN: the bare instructions (fmin / fmax add a canonicalising v_max_f64 x, x, x per operand: +40 % instructions)
__device__ __forceinline__ double kmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double kmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// The network is written for KPL keys per lane (element e = lane*KPL + r) of type T (u32: the 256-slot fast path;
// u64: the general path for windows above 15x15).
template <int LM, int RX, int BITLOG, class T, int KPL>  // partner = (lane ^ LM, r ^ RX); the lane with bit BITLOG clear is the lower one
__device__ __forceinline__ void cross_step(T (&key)[KPL], const LaneMasks& lm)
{
    const bool lower = lm.m[BITLOG] != 0u;
    T o[KPL];
#pragma unroll
    for (int r = 0; r < KPL; r++) o[r] = lane_xor<LM>(key[r ^ RX]);
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const T mn = kmin(key[r], o[r]), mx = kmax(key[r], o[r]);
        key[r] = lower ? mn : mx;  // lower lane keeps the minimum, upper lane the maximum
    }
}

template <int RX, class T, int KPL>  // intra-lane: pairs (r, r ^ RX)
__device__ __forceinline__ void local_step(T (&key)[KPL])
{
#pragma unroll
    for (int r = 0; r < KPL; r++) {
        const int q = r ^ RX;
        if (q > r) {
            const T a = key[r], b = key[q];
            key[r] = kmin(a, b);
            key[q] = kmax(a, b);
        }
    }
}

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v / 2); }

template <int J, class T, int KPL>  // half-cleaner steps j = J, J/2, ..., 1 (element distance)
__device__ __forceinline__ void half_cleaners(T (&key)[KPL], const LaneMasks& lm)
{
    if constexpr (J >= KPL) {
        cross_step<J / KPL, 0, ilog2(J / KPL)>(key, lm);
        half_cleaners<J / 2>(key, lm);
    } else if constexpr (J >= 1) {
        local_step<J>(key);
        if constexpr (J > 1) half_cleaners<J / 2>(key, lm);
    }
}

template <int K, class T, int KPL>
__device__ __forceinline__ void merge_sorted_halves(T (&key)[KPL], const LaneMasks& lm)
{
    if constexpr (K <= KPL) {
        local_step<K - 1>(key);  // mirror inside the lane: (r, r ^ (K-1))
        if constexpr (K > 2) half_cleaners<K / 4>(key, lm);
    } else {
        cross_step<K / KPL - 1, KPL - 1, ilog2(K / (2 * KPL))>(key, lm);  // mirror: lane ^ (K/KPL-1), register ^ (KPL-1)
        half_cleaners<K / 4>(key, lm);
    }
}

template <int K, class T, int KPL>
__device__ __forceinline__ void bitonic_sort(T (&key)[KPL], const LaneMasks& lm)
{
    if constexpr (K > 2) bitonic_sort<K / 2>(key, lm);
    merge_sorted_halves<K>(key, lm);
}


}  // namespace wmnet
