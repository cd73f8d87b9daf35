// Pointwise (1x1, stride 1) convolution with a LONG reduction (Cin >= 512) as a 256 x BN tiled GEMM with 8 waves — gfx950 only.
// Reference rows served: SURVEY.md §8a M1/M2 (`Conv`, model/utils.py:6-32): the 1x1 layers of the deep half of the network (512 ... 2048
// input channels at 25^2 / 50^2 / 100^2 maps: the ELAN / SPPCSPC transitions of yolov7 and their data gradients) — 45 launches and 6.6 ms of
// the batch-64 step on conv_gemm_kernel<128, 128, ..., T1> at 620 TF/s = 25 % of the MFMA peak and 29 % of the HBM peak: bound by neither
// (VERDICT r3 item 3).
//
// What holds the generic kernel there (DESIGN §4.1 phase counters): a 128 x 128 tile moves 16 KiB from L2 into LDS per 32-deep K step for
// 32 MFMAs of the workgroup (64 FLOP per L2 byte; three resident workgroups stream ~17 TB/s of L2 -> LDS traffic, half of what the L2s
// deliver), behind one barrier and a counted DMA wait per 8 MFMAs of a wave.  This kernel quadruples the tile:
//   * 256 pixels x BN = 256 (128) output channels per workgroup, 8 waves as 2 (pixels) x 4 (channels): 128 x 64 (128 x 32) per wave =
//     128 (64) accumulator registers; K steps of 64 input channels: 64 KiB (48 KiB) per step for 256 (128) MFMAs of the workgroup — 256 FLOP
//     per L2 byte — and 32 (16) MFMAs per wave between barriers;
//   * both operands by LDS-DMA into two 64-KiB stages (whole 128-byte K rows: every request is a full line; the bank swizzle — 16-byte slot
//     ^= (row >> 1) & 7, conflict-free for the 32 consecutive rows of a fragment read — is applied on the SOURCE address); the stage of
//     step k + 1 is requested right after the barrier that ends step k - 1 and lands under the 1 000+ matrix cycles of step k;
//   * fragment reads through inline asm with counted lgkmcnt (next to LDS-DMA in flight hipcc's waitcnt pass drains vmcnt(0) in front of
//     plain LDS reads: conv3x3_ws.hip), one 16-channel sub-step ahead of their MFMAs; two waves per SIMD cover each other's waits;
//   * operands swapped as in the other GEMMs (A = weights, B = pixels): a lane owns 4 consecutive channels of one pixel, the epilogue
//     stages packed 8-byte pieces in LDS and stores whole 128-byte (64-byte) row segments; BatchNorm statistics from the staged tile.
// One workgroup per CU (128 KiB of LDS): the epilogue is NOT overlapped (that is what the persistent 3x3 kernel does and this one could
// learn next); tiles are ordered XCD-locally, channel tiles fastest (the pixel rows of a tile are re-read from L2 by its channel siblings).
#include "conv_internal.h"
#include <type_traits>

#define G256_BM 256
#define G256_BK 64
#define G256_STAGE (2 * 256 * 128)                             // bytes of one stage at BN = 256: pixel rows then weight rows, 128 bytes each

extern __shared__ __attribute__((aligned(1024))) unsigned char g256_lds[];

template <int U, int N> struct G256Unroll {
    template <class F> static __device__ __forceinline__ void run(F& f)
    {
        f(std::integral_constant<int, U>{});
        G256Unroll<U + 1, N>::run(f);
    }
};
template <int N> struct G256Unroll<N, N> {
    template <class F> static __device__ __forceinline__ void run(F&) {}
};
template <int N> __device__ __forceinline__ void g256_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ bf16x8 g256_rd128(unsigned addr)
{
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
typedef unsigned g256_u4 __attribute__((ext_vector_type(4)));
typedef unsigned g256_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ g256_u4 g256_rd128u(unsigned addr)
{
    g256_u4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ g256_u2 g256_rd64u(unsigned addr)
{
    g256_u2 r;
    asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ void g256_wr64(unsigned addr, unsigned lo, unsigned hi)
{
    const g256_u2 v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// all LDS operations of this wave have completed; the fragments named are usable from here on
template <int NF> __device__ __forceinline__ void g256_wait_lds0(bf16x8 (&f)[NF])
{
    if constexpr (NF == 6)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]));
}
__device__ __forceinline__ void g256_wait_lds0(g256_u4& a, g256_u4& b, g256_u4& c, g256_u4& d)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void g256_wait_lds0(g256_u2& a, g256_u2& b, g256_u2& c, g256_u2& d)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// BN = 256: wave tile 128 pixels x 64 channels (TN = 2 channel blocks); BN = 128: 128 x 32 (TN = 1)
template <int BN, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm256_kernel(const ConvGemmParams p, const int gn)
{
    constexpr int TN = BN / 128;                               // 32-channel blocks per wave
    constexpr int WTN = 32 * TN;                               // channels per wave
    constexpr int NF = 4 + TN;                                 // fragments per 16-channel sub-step: 4 pixel blocks + TN channel blocks
    constexpr unsigned W_OFF = 256 * 128;                      // weight rows follow the pixel rows inside a stage
    constexpr unsigned STAGE = (256 + BN) * 128;
    constexpr int NPIECE = (256 + BN) / 8 / 8;                 // 1-KiB DMA pieces per wave and stage (8 rows each): 8 / 6
    constexpr bool STATS = EPI == EPI_STATS, ACCUM = EPI == EPI_ACCUM, AFFINE = EPI == EPI_AFFINE_ACT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int h = lane >> 5, l31 = lane & 31;
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int mb = tile / gn, nb = tile - mb * gn;
    const int64_t m0 = (int64_t)mb * G256_BM;
    const int n0 = nb * BN;
    const unsigned lbase = lds_addr(g256_lds);
    const int nk = p.Cin / G256_BK;
#ifdef G256_TIMING
    const unsigned long long T0 = __builtin_readcyclecounter();
#endif

    // ---- DMA sources: piece q = wave + 8 u covers stage rows 8 q ... 8 q + 7 (rows 0..255 pixels, 256.. weights); lane -> (row, 16-byte slot);
    //      the slot holds logical chunk slot ^ ((row >> 1) & 7) of the row's 128 bytes.  Rows past M / Nout read the zero page. ------------------
    const bf16_t* dsrc[NPIECE];
#pragma unroll
    for (int u = 0; u < NPIECE; u++) {
        const int q = wave + 8 * u;
        const int row = 8 * q + (lane >> 3);
        const int c8 = ((lane & 7) ^ ((row >> 1) & 7)) << 3;    // element offset of the lane's chunk inside the K step
        if (row < 256) {
            const int64_t m = m0 + row;
            dsrc[u] = m < M ? p.A + m * p.ldA + c8 : nullptr;
        } else {
            const int n = n0 + row - 256;
            dsrc[u] = n < p.Nout ? p.W + (int64_t)n * p.wtaps * p.Cin + c8 : nullptr;
        }
    }
    auto issue_stage = [&](int k, unsigned soff) {
#pragma unroll
        for (int u = 0; u < NPIECE; u++) {
            const bf16_t* src = dsrc[u] ? dsrc[u] + k * G256_BK : p.zeros;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(g256_lds + soff + (unsigned)(wave + 8 * u) * 1024u), 16, 0, 0);
        }
    };
    // ---- fragment addresses inside a stage (sub-step ks: ^ (ks << 5)) -------------------------------------------------------------------------
    unsigned pa[4], wa[TN];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned row = (unsigned)(wm * 128 + i * 32 + l31);
        pa[i] = (row << 7) + ((((row >> 1) & 7u) ^ (unsigned)h) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const unsigned row = (unsigned)(wn * WTN + j * 32 + l31);
        wa[j] = W_OFF + (row << 7) + ((((row >> 1) & 7u) ^ (unsigned)h) << 4);
    }

    f32x16 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // inference (EPI_AFFINE_ACT, r06 — the eval tape ran these layers on the generic kernel's 128 x 128 tile until then): the folded BatchNorm
    // coefficients of the tile's BN columns go into the 2 KiB behind the two operand stages BEFORE any LDS-DMA is in flight (a plain LDS write next
    // to DMA in flight makes hipcc drain vmcnt); the K loop's barriers publish them, the epilogue reads them back per accumulator quad
    float* const cco = reinterpret_cast<float*>(g256_lds + 2 * STAGE);      // [2][BN]
    if constexpr (AFFINE) {
        if (tid < BN) {
            const int n = n0 + tid;
            cco[tid] = n < p.Nout ? p.scale[n] : 0.f;
            cco[BN + tid] = n < p.Nout ? p.shift[n] : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    issue_stage(0, 0u);
#ifdef G256_TIMING
    const unsigned long long T1 = __builtin_readcyclecounter();
#endif
    for (int k = 0; k < nk; k++) {
        const unsigned soff = (k & 1) ? STAGE : 0u;
        g256_wait_vm<0>();                                      // my pieces of step k (requested a whole step ago) have landed ...
        __builtin_amdgcn_s_barrier();                           // ... everyone's have, and everyone is done with step k - 1
        unsigned sb = lbase + soff;
        asm volatile("" : "+v"(sb));                            // (opaque: keeps the 4 x NF fragment addresses of a step out of loop-invariant registers)
        bf16x8 fr[2][NF];
        auto rd_set = [&](int ks, bf16x8 (&f)[NF]) {
#pragma unroll
            for (int i = 0; i < 4; i++) f[i] = g256_rd128(sb + (pa[i] ^ (unsigned)(ks << 5)));
#pragma unroll
            for (int j = 0; j < TN; j++) f[4 + j] = g256_rd128(sb + (wa[j] ^ (unsigned)(ks << 5)));
        };
        rd_set(0, fr[0]);                                       // first fragments requested BEFORE the next stage's DMA is issued: their LDS latency
        __builtin_amdgcn_sched_barrier(0);                      // runs under the ~70 address / M0 / DMA instructions below
        if (k + 1 < nk) issue_stage(k + 1, (k & 1) ? 0u : STAGE);
        auto sub = [&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            __builtin_amdgcn_sched_barrier(0);
            g256_wait_lds0<NF>(fr[ks & 1]);                     // the set of this sub-step (requested one sub-step = 8 / 4 MFMAs ago)
            if constexpr (ks + 1 < 4) rd_set(ks + 1, fr[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ks & 1][4 + j], fr[ks & 1][i], acc[i][j], 0, 0, 0);
        };
        G256Unroll<0, 4>::run(sub);
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef G256_TIMING
    const unsigned long long T2 = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_barrier();                               // operand stages dead: LDS becomes the output staging (nothing is in flight: the
                                                                // last step requested no stage)
    // ---- epilogue: acc[i][j] = pixel block i x channel block j, lane owns pixel l31, channels 8 g4 + 4 h + (0..3) of the block.  Staging block
    //      of the wave: 128 pixel rows x WTN channels (2 WTN bytes per row), 16-byte chunks at position chunk ^ (row & (CHK - 1)). -------------------
    constexpr int CHK = WTN / 8;                               // chunks per staged row: 8 / 4
    constexpr unsigned ROWB = WTN * 2;
    const unsigned sbase = lbase + (unsigned)wave * (128u * ROWB);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int row = i * 32 + l31;
                const int chunk = j * 4 + g4;
                float v[4] = {acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
                if constexpr (AFFINE) {
                    const int cl = wn * WTN + j * 32 + 8 * g4 + 4 * h;    // column inside the tile (columns >= Nout hold zeros and are never stored)
                    const float4 sc = *reinterpret_cast<const float4*>(cco + cl), sh = *reinterpret_cast<const float4*>(cco + BN + cl);
                    const float sc4[4] = {sc.x, sc.y, sc.z, sc.w}, sf4[4] = {sh.x, sh.y, sh.z, sh.w};
                    act_affine_quad(v, sc4, sf4, p.act);
                }
                g256_wr64(sbase + (unsigned)row * ROWB + (unsigned)(((chunk ^ (row & (CHK - 1))) << 4) | (h << 3)),
                          pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
    // (wave-local hand-off: the wave's own LDS operations are ordered)
    const int ch = lane % CHK, r0 = lane / CHK;                // store: chunk ch of staged row it * RPI + r0
    constexpr int RPI = 64 / CHK;                              // rows per store instruction: 8 / 16
    const int ncol = n0 + wn * WTN + ch * 8;
    const bool col_ok = ncol < p.Nout;
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int NQ = WTN / 4;                                // channel quads: 16 / 8
    const int cq = lane % NQ, rg = lane / NQ;                  // statistics: quad cq, rows rg + RGS k
    constexpr int RGS = 64 / NQ;                               // 4 / 8
    if (STATS) {
        // rows past M hold exact zeros (zero-page operands), columns past Nout likewise
#pragma unroll
        for (int k4 = 0; k4 < 128 / RGS; k4 += 4) {
            g256_u2 w[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = rg + RGS * (k4 + q);
                w[q] = g256_rd64u(sbase + (unsigned)row * ROWB + (unsigned)((((cq >> 1) ^ (row & (CHK - 1))) << 4) | ((cq & 1) << 3)));
            }
            g256_wait_lds0(w[0], w[1], w[2], w[3]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float f0 = __uint_as_float(w[q].x << 16), f1 = __uint_as_float(w[q].x & 0xffff0000u);
                const float f2 = __uint_as_float(w[q].y << 16), f3 = __uint_as_float(w[q].y & 0xffff0000u);
                ssum[0] += f0; ssq[0] += f0 * f0;
                ssum[1] += f1; ssq[1] += f1 * f1;
                ssum[2] += f2; ssq[2] += f2 * f2;
                ssum[3] += f3; ssq[3] += f3 * f3;
            }
        }
    }
#pragma unroll
    for (int g0 = 0; g0 < 128 / RPI; g0 += 4) {
        bf16_t* optr[4];
        bool lv[4];
        uint4 oldv[4];
        g256_u4 sv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = (g0 + q) * RPI + r0;
            const int64_t m = m0 + wm * 128 + row;
            lv[q] = (m < M) & col_ok;
            optr[q] = reinterpret_cast<bf16_t*>(p.out) + (lv[q] ? m : m0) * p.ldC + (col_ok ? ncol : n0);
            if (ACCUM) oldv[q] = *reinterpret_cast<const uint4*>(optr[q]);
            sv[q] = g256_rd128u(sbase + (unsigned)row * ROWB + (unsigned)((ch ^ (row & (CHK - 1))) << 4));
        }
        g256_wait_lds0(sv[0], sv[1], sv[2], sv[3]);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint4 v = make_uint4(sv[q].x, sv[q].y, sv[q].z, sv[q].w);
            if (ACCUM) {
                const unsigned* a = reinterpret_cast<const unsigned*>(&v);
                const unsigned* b = reinterpret_cast<const unsigned*>(&oldv[q]);
                unsigned w[4];
#pragma unroll
                for (int e = 0; e < 4; e++)
                    w[e] = pack_bf2(__uint_as_float(a[e] << 16) + __uint_as_float(b[e] << 16),
                                    __uint_as_float(a[e] & 0xffff0000u) + __uint_as_float(b[e] & 0xffff0000u));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            if (lv[q]) *reinterpret_cast<uint4*>(optr[q]) = v;
        }
    }
#ifdef G256_TIMING
    if (p.bias && tid == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias)) + (size_t)blockIdx.x * 4;
        dbg[0] = T0; dbg[1] = T1; dbg[2] = T2; dbg[3] = __builtin_readcyclecounter();
    }
#endif
    if (STATS) {
        // per-tile partial row: lanes park their 8 sums, one thread per channel folds the 2 pixel halves x RGS row groups in a fixed order
        __syncthreads();                                        // every wave is done with its staging block
        float* part = reinterpret_cast<float*>(g256_lds);       // [wave][RGS][2][WTN]
        float* mine = part + ((wave * RGS + rg) * 2) * WTN + cq * 4;
        *reinterpret_cast<float4*>(mine) = make_float4(ssum[0], ssum[1], ssum[2], ssum[3]);
        *reinterpret_cast<float4*>(mine + WTN) = make_float4(ssq[0], ssq[1], ssq[2], ssq[3]);
        __syncthreads();
        if (tid < BN && n0 + tid < p.Nout) {
            const int wnc = tid / WTN, cc = tid % WTN;
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < 2; w++)
#pragma unroll
                for (int r = 0; r < RGS; r++) {
                    const float* src = part + (((w * 4 + wnc) * RGS + r) * 2) * WTN + cc;
                    sm += src[0];
                    sq += src[WTN];
                }
            float* st = p.stats + (int64_t)mb * 2 * p.Nout;
            st[n0 + tid] = sm;
            st[p.Nout + n0 + tid] = sq;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- host side
static int g256_mode()
{
    static const int v = [] { const char* e = getenv("RYOLO_GEMM_256"); return e ? atoi(e) : 1; }();      // 0 off, 1 by size, 2 every eligible launch (tests)
    return v;
}

bool g256_geometry(const ConvGemmParams& p, G256Geom& g)
{
    g = G256Geom{};
    if (!g256_mode() || (p.pipe & 0xff) != 1) return false;
    const TapClass& tc = p.cls[0];
    if (p.nclasses != 1 || tc.ntaps != 1 || tc.dh[0] || tc.dw[0] || tc.widx[0] || p.sh != 1 || p.sw != 1 || p.IH != p.OH || p.IW != p.OW) return false;
    if (p.oh_mul != 1 || p.ow_mul != 1 || p.OHf != p.OH || p.OWf != p.OW || tc.oh_add || tc.ow_add) return false;
    if (p.pool_idx || p.s2d_cin || !p.zeros) return false;
    if (p.epi != EPI_RAW && p.epi != EPI_STATS && p.epi != EPI_ACCUM && p.epi != EPI_AFFINE_ACT) return false;
    if (p.epi == EPI_AFFINE_ACT && (!p.scale || !p.shift)) return false;
    if (p.Cin % G256_BK || p.Nout % 8 || p.ldA % 8 || p.ldC % 8 || p.Nout < 128) return false;
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    if (M <= 0) return false;
    g.BN = p.Nout <= 128 ? 128 : 256;
    g.gm = ry_cdiv(M, G256_BM);
    g.gn = (int)ry_cdiv(p.Nout, g.BN);
    if (g.gm * g.gn > 0x7fffffff) return false;
    if (g256_mode() < 2) {
        // long reductions only (short-K pointwise layers are memory streams the generic kernel already runs at 4.3 - 5.2 TB/s), and grids of at
        // least ~2.4 rounds of one workgroup per CU (a 256-wide tile on a 1.2-round grid idles half the chip in its second round)
        // (and 256-column tiles only: the 256 x 128 form measured 443 TF/s on 512 -> 128 @100^2 against 487 on the generic kernel — half the MFMAs
        // per barrier for the same pixel stream; it stays reachable with RYOLO_GEMM_256=2 for the tests)
        static const int mink = [] { const char* e = getenv("RYOLO_GEMM_256_MINK"); return e ? atoi(e) : 512; }();      // A/B knob
        if (p.Cin < mink || g.gm * g.gn < 600 || g.BN != 256) return false;
    }
    g.lds_bytes = 2u * (256u + (unsigned)g.BN) * 128u + (p.epi == EPI_AFFINE_ACT ? 2u * (unsigned)g.BN * 4u : 0u);
    return true;
}

template <int BN, int EPI> static int g256_launch_t(const ConvGemmParams& p, const G256Geom& g, hipStream_t stream)
{
    static RyLdsAttr attr;
    if (ry_max_dynamic_lds(attr, reinterpret_cast<const void*>(&gemm256_kernel<BN, EPI>), 160 * 1024)) return RY_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm256_kernel<BN, EPI>), dim3((unsigned)(g.gm * g.gn)), dim3(512), g.lds_bytes, stream, p, g.gn);
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}

int g256_launch(const ConvGemmParams& p, const G256Geom& g, hipStream_t stream)
{
    if (g.BN == 256) {
        if (p.epi == EPI_STATS) return g256_launch_t<256, EPI_STATS>(p, g, stream);
        if (p.epi == EPI_ACCUM) return g256_launch_t<256, EPI_ACCUM>(p, g, stream);
        if (p.epi == EPI_AFFINE_ACT) return g256_launch_t<256, EPI_AFFINE_ACT>(p, g, stream);
        return g256_launch_t<256, EPI_RAW>(p, g, stream);
    }
    if (p.epi == EPI_STATS) return g256_launch_t<128, EPI_STATS>(p, g, stream);
    if (p.epi == EPI_ACCUM) return g256_launch_t<128, EPI_ACCUM>(p, g, stream);
    if (p.epi == EPI_AFFINE_ACT) return g256_launch_t<128, EPI_AFFINE_ACT>(p, g, stream);
    return g256_launch_t<128, EPI_RAW>(p, g, stream);
}
