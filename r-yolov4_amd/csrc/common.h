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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE function attribute: one flag word per (kernel, device), set with a relaxed
// atomic so that two host threads racing through the first launch both end up with the attribute in place (the call is idempotent).
#include <atomic>
struct RyLdsAttr { std::atomic<unsigned long long> done{0}; };     // bit d = device d configured (devices >= 64: always re-set)
static inline int ry_max_dynamic_lds(RyLdsAttr& st, const void* fn, int bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RY_ERR_LAUNCH;
    const unsigned long long bit = dev < 64 ? 1ull << dev : 0ull;
    if (bit && (st.done.load(std::memory_order_acquire) & bit)) return RY_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return RY_ERR_LAUNCH;
    if (bit) st.done.fetch_or(bit, std::memory_order_release);
    return RY_OK;
}

// ---- bf16 <-> f32 (round-to-nearest-even), device side -------------------------------------------------
#include "ryolo_params.h"   // bf16_t + POD parameter blocks

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// fp32 -> bf16, round-to-nearest-even: gfx950 has a packed hardware convert (v_cvt_pk_bf16_f32); clang emits exactly that
// instruction for a float2 -> __bf16x2 vector conversion (the hand-written integer RNE cost ~6 VALU per element).
typedef __bf16 ry_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ry_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi)
{
    const ry_f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, ry_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

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
