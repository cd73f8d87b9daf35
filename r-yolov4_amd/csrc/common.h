// Shared host/device helpers for libryolo_hip.so (gfx950 only — no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RY_OK 0
#define RY_ERR_ARG 1
#define RY_ERR_WORKSPACE 2
#define RY_ERR_LAUNCH 3
#define RY_ERR_UNSUPPORTED 4

#define RY_CHECK_LAUNCH()                                     \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return RY_ERR_LAUNCH;          \
    } while (0)

static inline int64_t ry_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- bf16 <-> f32 (round-to-nearest-even), device side -------------------------------------------------
#include "ryolo_params.h"   // bf16_t + POD parameter blocks

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }

// ---- wave64 reductions --------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
