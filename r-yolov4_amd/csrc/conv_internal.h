// Pieces shared by the convolution translation units (conv.hip, conv3x3.hip): epilogue / activation codes, MFMA vector
// types, the XCD-aware tile order.  Internal to libryolo_hip.so.
#pragma once
#include "common.h"
#include "params.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

enum { EPI_RAW = 0, EPI_STATS = 1, EPI_AFFINE_ACT = 2, EPI_F32_BIAS = 3, EPI_ACCUM = 4,
       EPI_AFFINE_ACT_R = 5 };      // stem only: the conv output is rounded to bf16 BEFORE the affine map (= what the two-pass training path stores)
enum { ACT_LINEAR = 0, ACT_MISH = 1, ACT_LEAKY = 2, ACT_SILU = 3 };

__device__ __forceinline__ float act_fwd(float u, int act)
{
    if (act == ACT_SILU) return u * __builtin_amdgcn_rcpf(1.f + __expf(-u));
    if (act == ACT_LEAKY) return u > 0.f ? u : 0.1f * u;
    if (act == ACT_MISH) {
        if (u > 20.f) return u;
        const float n = __expf(u), w = n * (n + 2.f);       // tanh(softplus(u)) = (n^2 + 2n) / (n^2 + 2n + 2)
        return u * w * __builtin_amdgcn_rcpf(w + 2.f);
    }
    return u;
}

// d act / du, same expressions as elementwise.hip's act_d (one v_exp_f32, v_rcp_f32 instead of IEEE division)
__device__ __forceinline__ float act_bwd(float u, int act)
{
    if (act == ACT_SILU) { const float s = __builtin_amdgcn_rcpf(1.f + __expf(-u)); return s * (1.f + u * (1.f - s)); }
    if (act == ACT_LEAKY) return u > 0.f ? 1.f : 0.1f;
    if (act == ACT_MISH) {
        if (u > 20.f) return 1.f;
        const float n = __expf(u), w = n * (n + 2.f);
        const float t = w * __builtin_amdgcn_rcpf(w + 2.f);
        const float sg = n * __builtin_amdgcn_rcpf(1.f + n);
        return t + u * (1.f - t * t) * sg;
    }
    return 1.f;
}

// ---- LDS transposed reads (ds_read_b64_tr_b16) next to LDS-DMA -----------------------------------------------------------------
// hipcc's waitcnt pass does not know what the ds_read_tr BUILTIN reads, so with LDS-DMA (global_load_lds) in flight it puts an
// s_waitcnt vmcnt(0) in front of the first transposed read of every loop iteration: the DMA just issued for a LATER stage is
// drained before the current stage is consumed, i.e. every K step pays a full memory round trip (seen in the ISA of the 3x3
// ring weight gradient and of the stem backward; plain ds_read_b128 / ds_read_b32 through ordinary pointers are not affected).
// Kernels that prefetch with LDS-DMA therefore issue their transposed reads through inline asm and count lgkmcnt themselves:
// lds_wait<N>(frag) = "at most N of MY later LDS reads may still be in flight", tied to the fragment so that its consumer cannot be
// scheduled above the wait.  (Reads the compiler issues on its own only make these waits more conservative: LDS returns in order.)
typedef __attribute__((ext_vector_type(4))) short ry_s16x4;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ ry_s16x4 lds_tr16(unsigned addr)
{
    ry_s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
// 8 consecutive K values (pixels) of one channel: two transposed reads, `second` bytes apart
__device__ __forceinline__ bf16x8 lds_tr16x2(unsigned addr, unsigned second)
{
    const ry_s16x4 lo = lds_tr16(addr), hi = lds_tr16(addr + second);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
template <int N> __device__ __forceinline__ void lds_wait(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
template <int N> __device__ __forceinline__ void lds_wait2(bf16x8& f, bf16x8& g) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f), "+v"(g) : "n"(N)); }

// bijective XCD remap (cdna guide T1): workgroup b runs on XCD b%8; give each XCD a contiguous tile range
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// exact n / d for 0 <= n < 2^16, 1 <= d < 2^16 with a float reciprocal (prologue index math; hipcc's generic 32-bit
// division is ~40 instructions, the 64-bit one several hundred)
__device__ __forceinline__ int small_div(int n, int d, float rd)
{
    int q = (int)((float)n * rd);
    if (q * d > n) q--;
    if ((q + 1) * d <= n) q++;
    return q;
}

// ---- 3x3 stride-1 halo-patch kernel (conv3x3.hip) -------------------------------------------------------------------
// geometry chosen on the host; `mode` 0 = not eligible, 1 = 2-D tiles (TH x TW output pixels of one image),
// 2 = flat runs of 256 consecutive output pixels
struct P3Geom {
    int mode, BN;
    int TH, TW, PW;            // tile rows / cols (2-D), patch row pitch in pixels (TW + 2, or the image width for flat runs)
    int R, P, TP;              // patch rows, 16-row DMA pieces, tap steps that carry one piece per wave (= ceil(P / 4))
    int tilesW, tilesPerImg;   // 2-D tiling of one image
    int gn;
    int64_t gm;
    unsigned lds_bytes;
    int tdh[9], tdw[9], twi[9];   // taps as scalars (kernarg dwords: read with s_load, unlike the byte arrays of TapClass)
    float rPW, rTW;            // reciprocals for the prologue's small exact divisions
};
bool p3_geometry(const ConvGemmParams& p, P3Geom& g);
int p3_launch(const ConvGemmParams& p, const P3Geom& g, hipStream_t stream);

// ---- 3x3 stride-1 weight gradient over a sliding halo ring (conv3x3.hip) ------------------------------------------------
// K runs over PADDED pixel coordinates (image framed by one zero pixel on every side), so every tap is a constant row
// offset into one ring of input rows and no per-element masks exist.
struct W3Geom {
    int ok;
    int PWp, HPp;              // padded width / height (W + 2, H + 2)
    int64_t Mp;                // padded pixels = NB * HPp * PWp
    int RX;                    // ring rows (power of two)
    int gx, gc;                // output-channel tiles of 128, input-channel chunks of 32
    int splitk, slabs, co64;   // K ranges; fp32 slabs written (2 per range for the <= 64 output-channel variant)
    int64_t kchunk;            // padded pixels per split (multiple of 32)
    int toff[9];               // dh * PWp + dw per tap (caller's tap order)
    unsigned lds_bytes;
};
bool w3_geometry(const WgradParams& p, W3Geom& g);
int w3_launch(const WgradParams& p, const W3Geom& g, hipStream_t stream);
