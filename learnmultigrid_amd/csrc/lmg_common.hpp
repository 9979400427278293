// Shared helpers of the gfx950 kernels (wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lmg.h"

#define LMG_WAVE 64

#define LMG_CHECK_LAUNCH()                                          \
    do {                                                            \
        hipError_t e__ = hipGetLastError();                         \
        if (e__ != hipSuccess) return LMG_ERR_LAUNCH;               \
    } while (0)

static inline hipStream_t lmg_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool lmg_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Sum over the 64 lanes of a wave, fixed butterfly order (deterministic).
__device__ __forceinline__ double lmg_wave_sum(double v)
{
#pragma unroll
    for (int off = LMG_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, LMG_WAVE);
    return v;
}

// Block-wide sum in fixed order; result valid in thread 0.  s_red holds BLOCK/64 doubles.
template <int BLOCK>
__device__ __forceinline__ double lmg_block_sum(double v, double *s_red)
{
    v = lmg_wave_sum(v);
    const int lane = threadIdx.x & (LMG_WAVE - 1), wave = threadIdx.x / LMG_WAVE;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < BLOCK / LMG_WAVE; ++w) tot += s_red[w];
    }
    return tot;
}
