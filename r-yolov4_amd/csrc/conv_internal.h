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

// Folded BatchNorm + activation of FOUR accumulator values (inference epilogues, EPI_AFFINE_ACT): v[q] = act(v[q] * sc[q] + sf[q]).  `act` is
// wave-uniform; called per element through act_fwd every value paid the whole `if` chain (three scalar compare + branch pairs, and Mish's
// `u > 20` early return as a DIVERGENT branch: 1 008 s_cbranch in the affine instantiation of the persistent pointwise kernel, r06 ISA count).  Here
// the chain is walked once per quad and every arm is branch-free; the values are bit-identical to act_fwd's (same expressions, the Mish
// cut-off as a select).
template <int N> __device__ __forceinline__ void act_affine_vec(float (&v)[N], const float (&sc)[N], const float (&sf)[N], int act)
{
    if (act == ACT_SILU) {
#pragma unroll
        for (int q = 0; q < N; q++) { const float u = v[q] * sc[q] + sf[q]; v[q] = u * __builtin_amdgcn_rcpf(1.f + __expf(-u)); }
    } else if (act == ACT_MISH) {
#pragma unroll
        for (int q = 0; q < N; q++) {
            const float u = v[q] * sc[q] + sf[q];
            const float n = __expf(u), w = n * (n + 2.f);
            const float r = u * w * __builtin_amdgcn_rcpf(w + 2.f);
            v[q] = u > 20.f ? u : r;
        }
    } else if (act == ACT_LEAKY) {
#pragma unroll
        for (int q = 0; q < N; q++) { const float u = v[q] * sc[q] + sf[q]; v[q] = u > 0.f ? u : 0.1f * u; }
    } else {
#pragma unroll
        for (int q = 0; q < N; q++) v[q] = v[q] * sc[q] + sf[q];
    }
}
__device__ __forceinline__ void act_affine_quad(float (&v)[4], const float (&sc)[4], const float (&sf)[4], int act) { act_affine_vec<4>(v, sc, sf, act); }

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

// d act / du of N values with ONE walk of the (wave-uniform) activation chain and branch-free arms — bit-identical to N calls of act_bwd
// (same expressions; Mish's `u > 20` cut-off as a select).  The fused first-layer backward (stem.hip) is instruction-rate bound: per element the
// chain was three scalar compare + branch pairs plus a divergent early return.
template <int N> __device__ __forceinline__ void act_bwd_vec(const float (&u)[N], float (&d)[N], int act)
{
    if (act == ACT_SILU) {
#pragma unroll
        for (int q = 0; q < N; q++) { const float s = __builtin_amdgcn_rcpf(1.f + __expf(-u[q])); d[q] = s * (1.f + u[q] * (1.f - s)); }
    } else if (act == ACT_MISH) {
#pragma unroll
        for (int q = 0; q < N; q++) {
            const float n = __expf(u[q]), w = n * (n + 2.f);
            const float t = w * __builtin_amdgcn_rcpf(w + 2.f);
            const float sg = n * __builtin_amdgcn_rcpf(1.f + n);
            const float r = t + u[q] * (1.f - t * t) * sg;
            d[q] = u[q] > 20.f ? 1.f : r;
        }
    } else if (act == ACT_LEAKY) {
#pragma unroll
        for (int q = 0; q < N; q++) d[q] = u[q] > 0.f ? 1.f : 0.1f;
    } else {
#pragma unroll
        for (int q = 0; q < N; q++) d[q] = 1.f;
    }
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
// the same read with a compile-time byte offset (ds_read's 16-bit immediate): one address register serves several rows
template <int OFF> __device__ __forceinline__ ry_s16x4 lds_tr16_off(unsigned addr)
{
    ry_s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
// waits tied to the two 64-bit HALVES of fragments: the halves are joined into the MFMA operand after the wait
template <int N> __device__ __forceinline__ void lds_wait_h(ry_s16x4& a, ry_s16x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int N> __device__ __forceinline__ void lds_wait_h2(ry_s16x4& a, ry_s16x4& b, ry_s16x4& c, ry_s16x4& d)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
__device__ __forceinline__ bf16x8 join_halves(ry_s16x4 lo, ry_s16x4 hi)
{
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

// ---- persistent weight-stationary 3x3 kernel for 64 -> <= 64 channels (conv3x3_ws.hip): weights in registers, double-buffered whole-K patches,
// one workgroup per CU walking 2-D tiles
struct Ws3Geom {
    int TH, TW, PW;            // tile rows / cols, patch row pitch in pixels (TW + 2)
    int R, NP;                 // patch rows, 8-row (1 KiB) DMA pieces
    int tilesW, tilesPerImg;
    int nwg;                   // persistent workgroups (a multiple of 8) = partial-statistics rows of the EPI_STATS epilogue
    int64_t ntiles;
    unsigned patch_bytes, lds_bytes;
    int tdh[9], tdw[9], twi[9];
    float rPW, rTW;
};
bool ws3_geometry(const ConvGemmParams& p, Ws3Geom& g);
int ws3_launch(const ConvGemmParams& p, const Ws3Geom& g, hipStream_t stream);

// ---- 256 x 256 (256 x 128) tiled pointwise GEMM for long reductions (gemm256.hip): 8 waves, 64-channel K steps, two 64-KiB LDS-DMA stages
struct G256Geom {
    int BN, gn;
    int64_t gm;                // pixel tiles of 256 = partial-statistics rows of the EPI_STATS epilogue
    unsigned lds_bytes;
};
bool g256_geometry(const ConvGemmParams& p, G256Geom& g);
int g256_launch(const ConvGemmParams& p, const G256Geom& g, hipStream_t stream);

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
    int step64;                // 1: 64-pixel K steps (conv3x3_wgrad64_kernel), 0: 32-pixel steps
    int64_t kchunk;            // padded pixels per split (multiple of 32)
    int toff[9];               // dh * PWp + dw per tap (caller's tap order)
    int tap_of[9];             // caller's tap index of the tap (dh, dw) at 3 * (dh + 1) + (dw + 1) (the 64-pixel kernels walk the taps row by row)
    int mirror;                // 1: the ring is followed by a copy of its first 16 rows (conv3x3_wgrad64_kernel: immediate row offsets never wrap)
    int pd;                    // conv3x3_wgrad8_kernel: prefetch distance in steps (1 or 2)
    int v8;                    // 0: the 4-wave kernels of conv3x3.hip; 2 / 4: conv3x3_wgrad8_kernel<NCO> (conv3x3_wgrad8.hip), gc then counts 64-channel chunks
    unsigned lds_bytes;
    // exact n / d for 0 <= n < 2^31 as (mulhi(n, m) >> s): d = HPp * PWp (padded pixels per image) and d = PWp (padded row length) —
    // the DMA address generation decomposes a padded pixel index without loops or branches
    unsigned m_img, s_img, m_row, s_row;
};
bool w3_geometry(const WgradParams& p, W3Geom& g);
int w3_launch(const WgradParams& p, const W3Geom& g, hipStream_t stream);
bool w8_geometry(const WgradParams& p, W3Geom& g);           // conv3x3_wgrad8.hip, called by w3_geometry on a W3Geom whose padded sizes / taps / magic numbers are set
int w8_launch(const WgradParams& p, const W3Geom& g, hipStream_t stream);

// ---- pointwise weight gradient on 256 x 256 tiles, 8 waves (wgrad1x1_8w.hip): eligibility + split (what ryolo_conv_wgrad_plan reports), launch
bool w1x8_geometry(const WgradParams& p, int* splitk, int64_t* kchunk, int* gx, int* gy);
int w1x8_launch(const WgradParams& p, hipStream_t stream);

// ---- streaming 3x3 stride-2 forward for 32 input channels (conv3x3s2_c32.hip; r06): weights in LDS, B fragments straight from global memory --
struct S2cGeom {
    int ok;
    int cblocks;               // 32-pixel column blocks per output row
    int64_t tiles;             // wave tiles = NB * cblocks * OH
    int nwg;                   // persistent workgroups (= statistics rows)
    unsigned lds_bytes;
};
bool s2c_geometry(const ConvGemmParams& p, S2cGeom& g);
int s2c_launch(const ConvGemmParams& p, const S2cGeom& g, hipStream_t stream);
bool s2c_dgrad_geometry(const ConvGemmParams& p, S2cGeom& g);      // its data gradient in the space-to-depth form (ConvGemmParams.s2d_cin == 32)
int s2c_dgrad_launch(const ConvGemmParams& p, const S2cGeom& g, hipStream_t stream);

// ---- weight-stationary persistent 1x1 GEMM (gemm1x1.hip): Cin <= 256, identity grid, bf16 epilogues ---------------------------------
struct Ws1Geom {
    int ok;
    int nk;                    // 32-channel K steps
    int gridN, wgn;            // 128-channel n tiles; workgroups per n tile (grid = gridN * wgn <= CUs, one workgroup per CU)
    int stats_rows;            // partial-statistics rows = waves per n tile (one row per wave for the whole launch)
    int s2d;                   // 1: the space-to-depth data-gradient instantiation (2 x 2 taps, depth-to-space store)
    int pool;                  // 1: the instantiation that adds a MaxPool2d(2, 2) gradient in its store (ConvGemmParams.pool_idx)
    unsigned lds_bytes;
};
bool ws1_geometry(const ConvGemmParams& p, Ws1Geom& g);
int ws1_launch(const ConvGemmParams& p, const Ws1Geom& g, hipStream_t stream);
