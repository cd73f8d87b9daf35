// Weight-stationary persistent 1x1 convolution GEMM for gfx950 (MI355X): forward and stride-1 data gradient of the pointwise
// convolutions of the reference's conv stack (model/utils.py:13-23 `Conv` with k = 1, the ELAN / MP / SPPCSPC transitions of
// model/backbone.py and model/neck.py) when the reduction length is short (Cin <= 256).
//
// Why its own kernel.  On the generic implicit GEMM (conv.hip) a K = 128 ... 256 layer spends as long around its K loop as in it (tile
// prologue, pipeline fill, store loop, statistics: tools/gemm_phases.py), re-fetches the 128 x K weight tile for every 128-pixel tile, and
// issues one LDS-DMA instruction per two MFMAs with a workgroup barrier per 32-channel step.  Here
//   * ONE workgroup per CU (8 waves) stays resident for the whole launch and keeps its 128 x K weight tile in LDS (<= 64 KiB);
//   * every WAVE is an independent stream over 64-pixel x 128-channel output tiles: it DMAs its own 64 input rows per 32-channel step
//     (global_load_lds, 3-stage private ring, counted vmcnt) and nobody else reads them, so the steady state has NO workgroup barrier —
//     the two waves of a SIMD drift out of phase and one's epilogue runs under the other's MFMAs;
//   * a 64 x 128 wave tile is 16 MFMAs (v_mfma_f32_32x32x16_bf16) per 12 ds_read_b128 and per 4 DMA instructions (generic: 8 per 8 per 4);
//   * BatchNorm batch statistics are accumulated in registers over ALL tiles of a wave and written once: rows = waves of the launch
//     (a 200 x 200 x 64-image layer leaves 2048 partial rows instead of 20 000).
// Epilogues: raw bf16 store, bf16 store + batch statistics, folded BatchNorm + activation (inference plans), accumulate.
#include "conv_internal.h"
#include <stdlib.h>

#define WS_BN 128                 // output channels per workgroup (the resident weight tile)
#define WS_TM 64                  // pixels per wave tile
#define WS_WAVES 8
#define WS_STG 3                  // ring stages per wave: [64 px][32 ch] bf16 = 4 KiB each
#define WS_RING (WS_STG * WS_TM * 32)      // elements per wave

template <int N> __device__ __forceinline__ void ws_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// S2D instantiations: the space-to-depth form of a stride-2 3x3 data gradient with 32 input channels (ConvGemmParams.s2d_cin == 32: ONE
// stride-1 GEMM over the dY grid, 2 x 2 taps, Nout = 4 parity blocks x 32 channels — the second layer of every backbone, the slowest launch of
// the step on the generic kernel: 8-16 K steps that re-fetch a 64 KiB weight tile per 128 pixels).  K step k is (tap, 32-channel chunk):
// a row's source is its pixel shifted by the tap (zero page outside the image), and quarter j of the epilogue IS parity block (ph, pw) = (j >> 1,
// j & 1): row (img, a, b) of the tile goes to pixel (2a + ph, 2b + pw) of the full-resolution gradient, 64 bytes each.
struct WsTaps {
    int koff[8];               // K step -> element offset of its source relative to the row's own pixel: (dh * IW + dw) * ldA + chunk * 32
    int wkoff[8];              // K step -> element offset inside a weight row: widx * Cin + chunk * 32
    int ktap[8];               // K step -> tap (bit of the row's validity mask)
    int dh[4], dw[4], ntaps;
    unsigned m_img, s_img, m_row, s_row;       // exact n / (OH * OW) and n / OW for n < 2^31 as (mulhi(n, m) >> s); m == 0: divisor 1
};

// POOL instantiations: pointwise data gradients that also carry the gradient of a MaxPool2d(2, 2) of the same tensor (ConvGemmParams.pool_idx,
// MaxConv blocks): out[(h, w), n] += pool_dz[(h / 2, w / 2), n] where the argmax offset of that window equals (h & 1) * 2 + (w & 1) — the arithmetic
// of the generic kernel's store loop (round the GEMM value to bf16, add, round; then the accumulate epilogue).  On the generic kernel these launches
// lose the identity-grid fast path (814 us for 128 -> 256 @200^2 against ~500 for the plain layer).
template <int EPI, bool S2D = false, bool POOL = false>
__global__ __launch_bounds__(512, 1) void gemm1x1_ws_kernel(const ConvGemmParams p, const int nk, const int gridN, const int wgn, const WsTaps tk)
{
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: tile indices, ring addresses and DMA destinations stay in SGPRs
    const int id = xcd_remap(blockIdx.x, gridDim.x);              // an XCD's workgroups: contiguous ids, n tile fastest -> the n tiles of a pixel range share one L2
    const int nb = id % gridN, slot = id / gridN;
    const int n0 = nb * WS_BN;
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    const int64_t tiles = (M + WS_TM - 1) / WS_TM;
    const int64_t wrow = (int64_t)p.wtaps * p.Cin;                // weight row stride (elements)
    const int swz = (lane >> 4) & 3;                              // (row >> 2) & 3 of every DMA row this lane fetches (row = 16 q + lane / 4)

    // ---- the weight tile: nk planes of [128 channels][32] bf16, each laid out like a B stage of the generic kernel ----------------
    {
        const int r = wave * 16 + (lane >> 2);                    // plane row of this lane's piece
        const bool ok = n0 + r < p.Nout;
        const bf16_t* src = p.W + (int64_t)(n0 + r) * wrow + (((lane & 3) ^ swz) << 3);
        for (int c = 0; c < nk; c++)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(ok ? src + (S2D ? tk.wkoff[c] : c * 32) : p.zeros), (lds_void_t*)(smem + c * (WS_BN * 32) + wave * 512), 16, 0, 0);
    }
    bf16_t* ring = smem + nk * (WS_BN * 32) + wave * WS_RING;
    const bf16_t* abase = p.A + (int64_t)(lane >> 2) * p.ldA + (((lane & 3) ^ swz) << 3);
    const int64_t qoff = (int64_t)16 * p.ldA;
    ws_wait_vm<0>();
    __syncthreads();                                               // the only workgroup barrier of the launch

    const int h = lane >> 5, l31 = lane & 31;
    // fragment read offsets (elements) inside a [rows][32] plane / stage: row r, 16-byte slot sl ^ ((r >> 2) & 3)
    int fo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) fo[ks] = (l31 * 4 + ((ks * 2 + h) ^ ((l31 >> 2) & 3))) * 8;   // + 32 rows: row bits above 4 do not change the swizzle

    constexpr bool stats = EPI == EPI_STATS, accum = EPI == EPI_ACCUM, affine = EPI == EPI_AFFINE_ACT;
    float ssum[stats ? 4 : 1][4], ssq[stats ? 4 : 1][4];
#pragma unroll
    for (int a = 0; a < (stats ? 4 : 1); a++)
#pragma unroll
        for (int b = 0; b < 4; b++) { ssum[a][b] = 0.f; ssq[a][b] = 0.f; }

    // The ring is ONE continuous stream of 32-channel stages across the wave's tiles: global step s lives in slot s % 3, and during step s
    // the four pieces of step s + 2 are issued — the first two stages of the NEXT tile while this tile's last two steps compute, so they
    // are in flight (or landed) behind the epilogue, whose quarter-tile staging uses the one slot the last step has just freed.
    // vmcnt is one in-order counter for loads and stores: a step waits for "everything but the stage after mine [and the 16 stores of the
    // previous epilogue, which were issued after the first two stages of this tile]".
    const int64_t tstride = (int64_t)wgn * WS_WAVES;
    const bool full_n = n0 + WS_BN <= p.Nout;
    int64_t t = (int64_t)slot * WS_WAVES + wave;
    // S2D: (image, row, column) of the lane's four rows of tile t_ (row = 16 q + lane / 4 — the DMA rows and the store rows of a lane are the
    // same four): validity of every tap as bit (8 q + tap), and the output pixel index (img * OHf + 2 a) * OWf + 2 b
    auto row_info = [&](int64_t t_, unsigned& vmask, int64_t* pix) {
        vmask = 0;
        const int HW = p.OH * p.OW;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t m = t_ * WS_TM + q * 16 + (lane >> 2);
            const unsigned mm = (unsigned)(m < M ? m : 0);
            const unsigned img = tk.m_img ? __umulhi(mm, tk.m_img) >> tk.s_img : mm;
            const unsigned rem = mm - img * (unsigned)HW;
            const unsigned a = tk.m_row ? __umulhi(rem, tk.m_row) >> tk.s_row : rem;
            const unsigned b = rem - a * (unsigned)p.OW;
            if (m < M) {
#pragma unroll
                for (int tp_ = 0; tp_ < 4; tp_++)
                    if (tp_ < tk.ntaps && (unsigned)((int)a + tk.dh[tp_]) < (unsigned)p.IH && (unsigned)((int)b + tk.dw[tp_]) < (unsigned)p.IW) vmask |= 1u << (8 * q + tp_);
            }
            if (pix) pix[q] = ((int64_t)img * p.OHf + 2 * a) * p.OWf + 2 * b;
        }
    };
    auto issue = [&](const bf16_t* tp_, int rows_, unsigned vm_, int k, int sl, int q) {
        const bf16_t* src;
        if constexpr (S2D) src = (vm_ >> (8 * q + tk.ktap[k])) & 1 ? tp_ + q * qoff + tk.koff[k] : p.zeros;
        else src = q * 16 + (lane >> 2) < rows_ ? tp_ + q * qoff + k * 32 : p.zeros;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(ring + sl * (WS_TM * 32) + q * 512), 16, 0, 0);
    };
    if (t < tiles) {
        const bf16_t* tp = abase + t * WS_TM * p.ldA;
        int rows_left = (int)(M - t * WS_TM < WS_TM ? M - t * WS_TM : WS_TM);
        unsigned vmask = 0, vmask_next = 0;
        if constexpr (S2D) row_info(t, vmask, nullptr);
#pragma unroll
        for (int q = 0; q < 4; q++) issue(tp, rows_left, vmask, 0, 0, q);
#pragma unroll
        for (int q = 0; q < 4; q++) issue(tp, rows_left, vmask, 1, 1, q);
        int cur = 0;                                               // slot of the stage consumed next
        bool first = true;
        while (true) {
            const int64_t tn = t + tstride;
            const bool has_next = tn < tiles;
            const bf16_t* tpn = abase + tn * WS_TM * p.ldA;
            const int rows_next = has_next ? (int)(M - tn * WS_TM < WS_TM ? M - tn * WS_TM : WS_TM) : 0;
            if constexpr (S2D) { if (has_next) row_info(tn, vmask_next, nullptr); }
            f32x16 acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
            for (int k = 0; k < nk; k++) {
                if (k == nk - 1 && !has_next) ws_wait_vm<0>();      // the wave's very last stage: nothing newer was issued
                else if (!first && k < 2 && full_n) ws_wait_vm<4 + 16>();   // newer than my stage: the next stage and the previous epilogue's 16 stores
                                                                    // (a ragged n tile may skip whole quarters' stores: it waits for them too, which is always safe)
                else ws_wait_vm<4>();
                const bf16_t* sa = ring + cur * (WS_TM * 32);
                const bf16_t* sb = smem + k * (WS_BN * 32);
                bf16x8 af[2][2], bfr[2][4];
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
#pragma unroll
                    for (int i = 0; i < 2; i++) af[ks][i] = *reinterpret_cast<const bf16x8*>(sa + i * (32 * 32) + fo[ks]);
#pragma unroll
                    for (int j = 0; j < 4; j++) bfr[ks][j] = *reinterpret_cast<const bf16x8*>(sb + j * (32 * 32) + fo[ks]);
                }
                __builtin_amdgcn_sched_barrier(0);
                // the stage two steps ahead: this tile's, or the next tile's first / second
                const bool here = k + 2 < nk, more = here || has_next;
                const bf16_t* ip = here ? tp : tpn;
                const int irows = here ? rows_left : rows_next, ik = here ? k + 2 : k + 2 - nk;
                const unsigned ivm = here ? vmask : vmask_next;
                const int isl = cur == 0 ? 2 : cur - 1;             // (cur + 2) % 3
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int ks = q >> 3, i = (q >> 2) & 1, j = q & 3;
                    // A operand = weights, B operand = pixels: the accumulator holds the transposed tile (a lane owns 4 consecutive channels of one pixel)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
                    if ((q & 3) == 3) {                             // four DMA pieces in the MFMA shadow
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) issue(ip, irows, ivm, ik, isl, q >> 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                cur = cur == 2 ? 0 : cur + 1;
            }
            // ---- epilogue: four 32-channel quarters through the slot the last step consumed: [64 px][32 ch] bf16, 16-byte chunk c of row r
            // at chunk c ^ ((r >> 1) & 3) (8-byte writes of a lane quad and the 16-byte row reads both spread over the banks) ---------------
            bf16_t* stage = ring + (cur == 0 ? 2 : cur - 1) * (WS_TM * 32);
            const int64_t m0 = t * WS_TM;
            int64_t pix[4] = {0, 0, 0, 0};
            if constexpr (S2D) { unsigned unused; row_info(t, unused, pix); }
            int64_t pool_pp[4] = {0, 0, 0, 0};                      // POOL: pooled pixel and window offset of the lane's four rows
            unsigned pool_want[4] = {0, 0, 0, 0};
            if constexpr (POOL) {
                const int HW = p.OH * p.OW;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int64_t m = t * WS_TM + q * 16 + (lane >> 2);
                    const unsigned mm = (unsigned)(m < M ? m : 0);
                    const unsigned img = tk.m_img ? __umulhi(mm, tk.m_img) >> tk.s_img : mm;
                    const unsigned rem = mm - img * (unsigned)HW;
                    const unsigned a = tk.m_row ? __umulhi(rem, tk.m_row) >> tk.s_row : rem;
                    const unsigned b = rem - a * (unsigned)p.OW;
                    pool_pp[q] = ((int64_t)img * (p.OH >> 1) + (a >> 1)) * (p.OW >> 1) + (b >> 1);
                    pool_want[q] = (a & 1) * 2 + (b & 1);
                }
            }
            auto pool_add = [&](uint4& v_, const unsigned long long packed, const uint4& gz, const unsigned want) {
                const unsigned* a = reinterpret_cast<const unsigned*>(&v_);
                const unsigned* b = reinterpret_cast<const unsigned*>(&gz);
                unsigned w[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float g0 = ((packed >> (16 * e)) & 0xff) == want ? __uint_as_float(b[e] << 16) : 0.f;
                    const float g1 = ((packed >> (16 * e + 8)) & 0xff) == want ? __uint_as_float(b[e] & 0xffff0000u) : 0.f;
                    w[e] = pack_bf2(__uint_as_float(a[e] << 16) + g0, __uint_as_float(a[e] & 0xffff0000u) + g1);
                }
                v_ = make_uint4(w[0], w[1], w[2], w[3]);
            };
            const int rc = lane & 3, rr = lane >> 2;                // store phase: 16-byte chunk rc of rows rr + 16 it
            const int rsw = (rc ^ ((lane >> 3) & 3)) * 8;
            const int wsw = (l31 >> 1) & 3;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // inference: folded BatchNorm + activation on the fp32 accumulators.  A lane's 16 values of a (pixel block i, quarter j) are channels
                // n0 + 32 j + 8 g4 + 4 h + e; the coefficients of the quarter's 32 channels are WAVE-UNIFORM (n0 comes from the workgroup id, j is
                // unrolled), so they come through the constant address space (s_load) — this kernel's LDS is full at K = 256 (no room for the per-tile
                // table the generic and halo-patch kernels have) and its vmcnt counts the ring's LDS-DMA pieces (a VMEM load here would wait for them).
                // r06 (second cut): the coefficients of HALF a quarter are requested in one batch, selected per half-wave into 16 VGPRs, and serve both
                // pixel blocks; the first cut fetched 16 scalars per (i, g4) quad behind two dependent s_waitcnt lgkmcnt(0) — 64 exposed scalar-memory
                // round trips per 64 x 128 tile, now 8, against 4 096 matrix cycles (ISA count; 256 -> 256 @200^2 x 64: 810 us against 595 for the raw store)
#pragma unroll
                for (int gp = 0; gp < 2; gp++) {                              // half a quarter at a time: 8 + 8 coefficient registers (16 + 16 spilled)
                    float qsc[affine ? 8 : 1], qsf[affine ? 8 : 1];
                    if constexpr (affine) {
                        typedef const __attribute__((address_space(4))) float cfloat_t;
#pragma unroll
                        for (int gg = 0; gg < 2; gg++) {
                            const int n8 = n0 + j * 32 + 8 * (2 * gp + gg);       // Nout % 8 == 0: the eight channels are inside or outside as a whole
                            const int n8c = n8 < p.Nout ? n8 : 0;                 // (columns >= Nout are never stored: any in-range coefficients do)
                            // (opaque per tile: the coefficients do not depend on the tile, and hipcc would hoist all 128 + 128 selected values out of
                            // the tile loop — 20 ... 35 VGPRs spilled to scratch in the first two cuts)
                            unsigned long long sca = (unsigned long long)(p.scale + n8c), sfa = (unsigned long long)(p.shift + n8c);
                            asm volatile("" : "+s"(sca), "+s"(sfa));
                            cfloat_t* scp = (cfloat_t*)sca;
                            cfloat_t* sfp = (cfloat_t*)sfa;
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const float sclo = scp[e], schi = scp[4 + e], sflo = sfp[e], sfhi = sfp[4 + e];
                                qsc[4 * gg + e] = h ? schi : sclo;
                                qsf[4 * gg + e] = h ? sfhi : sflo;
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) v[e] = acc[i][j][8 * gp + e];
                        if constexpr (affine) act_affine_vec<8>(v, qsc, qsf, p.act);       // the activation chain is walked once per 8 values
#pragma unroll
                        for (int gg = 0; gg < 2; gg++)
                            *reinterpret_cast<uint2*>(stage + (i * 32 + l31) * 32 + (((2 * gp + gg) ^ wsw) << 3) + 4 * h) =
                                make_uint2(pack_bf2(v[4 * gg], v[4 * gg + 1]), pack_bf2(v[4 * gg + 2], v[4 * gg + 3]));
                    }
                }
                // (same-wave LDS hand-off: the wave's own ds_write -> ds_read ordering)
                const int n = S2D ? rc * 8 : n0 + j * 32 + rc * 8;       // S2D: channel inside parity block j
                bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + (m0 + rr) * p.ldC + n;
                bf16_t* os[4];                                            // S2D: the four rows' pixels of parity block j
                if constexpr (S2D) {
#pragma unroll
                    for (int it = 0; it < 4; it++) os[it] = reinterpret_cast<bf16_t*>(p.out) + (pix[it] + (int64_t)(j >> 1) * p.OWf + (j & 1)) * p.ldC + n;
                }
                auto dst = [&](int it) -> bf16_t* {
                    if constexpr (S2D) return os[it];
                    else return o + (int64_t)it * 16 * p.ldC;
                };
                const bool n_ok = S2D || n < p.Nout;
                uint4 v[4];
#pragma unroll
                for (int it = 0; it < 4; it++) v[it] = *reinterpret_cast<const uint4*>(stage + (it * 16 + rr) * 32 + rsw);
                if (rows_left == WS_TM) {                           // wave-uniform: every tile but the last of the tensor
                    if (n_ok) {
                        if constexpr (POOL) {
                            unsigned long long pk[4];
                            uint4 gz[4];
#pragma unroll
                            for (int it = 0; it < 4; it++) {
                                pk[it] = *reinterpret_cast<const unsigned long long*>(p.pool_idx + pool_pp[it] * p.pool_ldi + n);
                                gz[it] = *reinterpret_cast<const uint4*>(p.pool_dz + pool_pp[it] * p.pool_ld + n);
                            }
#pragma unroll
                            for (int it = 0; it < 4; it++) pool_add(v[it], pk[it], gz[it], pool_want[it]);
                        }
                        if constexpr (accum) {
                            uint4 oldv[4];
#pragma unroll
                            for (int it = 0; it < 4; it++) oldv[it] = *reinterpret_cast<const uint4*>(dst(it));
#pragma unroll
                            for (int it = 0; it < 4; it++) {
                                const unsigned* a = reinterpret_cast<const unsigned*>(&v[it]);
                                const unsigned* b = reinterpret_cast<const unsigned*>(&oldv[it]);
                                unsigned w[4];
#pragma unroll
                                for (int e = 0; e < 4; e++)
                                    w[e] = pack_bf2(__uint_as_float(a[e] << 16) + __uint_as_float(b[e] << 16),
                                                    __uint_as_float(a[e] & 0xffff0000u) + __uint_as_float(b[e] & 0xffff0000u));
                                v[it] = make_uint4(w[0], w[1], w[2], w[3]);
                            }
                        }
#pragma unroll
                        for (int it = 0; it < 4; it++) *reinterpret_cast<uint4*>(dst(it)) = v[it];
                    }
                } else {                                            // the tensor's last tile (also this wave's last): row masks
#pragma unroll
                    for (int it = 0; it < 4; it++) {
                        if (it * 16 + rr >= rows_left || !n_ok) continue;
                        uint4 w = v[it];
                        if constexpr (POOL)
                            pool_add(w, *reinterpret_cast<const unsigned long long*>(p.pool_idx + pool_pp[it] * p.pool_ldi + n),
                                     *reinterpret_cast<const uint4*>(p.pool_dz + pool_pp[it] * p.pool_ld + n), pool_want[it]);
                        if constexpr (accum) {
                            const uint4 old = *reinterpret_cast<const uint4*>(dst(it));
                            const unsigned* a = reinterpret_cast<const unsigned*>(&w);
                            const unsigned* b = reinterpret_cast<const unsigned*>(&old);
                            unsigned x[4];
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                x[e] = pack_bf2(__uint_as_float(a[e] << 16) + __uint_as_float(b[e] << 16),
                                                __uint_as_float(a[e] & 0xffff0000u) + __uint_as_float(b[e] & 0xffff0000u));
                            w = make_uint4(x[0], x[1], x[2], x[3]);
                        }
                        *reinterpret_cast<uint4*>(dst(it)) = w;
                    }
                }
                if constexpr (stats) {
                    // column sums of the values actually stored (bf16-rounded); rows past M are exact zeros (zero input rows):
                    // lane -> (4-channel quad cq, row group rg), rows rg + 8 k
                    const int cq = lane & 7, rg = lane >> 3;
                    const int so = (((cq >> 1) ^ ((lane >> 4) & 3)) << 3) + (cq & 1) * 4;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const uint2 w = *reinterpret_cast<const uint2*>(stage + (rg + 8 * k) * 32 + so);
                        const float f0 = __uint_as_float(w.x << 16), f1 = __uint_as_float(w.x & 0xffff0000u);
                        const float f2 = __uint_as_float(w.y << 16), f3 = __uint_as_float(w.y & 0xffff0000u);
                        ssum[j][0] += f0; ssq[j][0] += f0 * f0;
                        ssum[j][1] += f1; ssq[j][1] += f1 * f1;
                        ssum[j][2] += f2; ssq[j][2] += f2 * f2;
                        ssum[j][3] += f3; ssq[j][3] += f3 * f3;
                    }
                }
            }
            if (!has_next) break;
            // the staged reads of the last quarter are done before the slot is refilled (step 0 of the next tile issues into it)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            first = false;
            t = tn;
            tp = tpn;
            rows_left = rows_next;
            vmask = vmask_next;
        }
    }
    if constexpr (stats) {
        // one partial row per wave: the eight row groups of a channel quad fold through the wave's ring, lane c then owns channels 2c, 2c + 1
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        float* part = reinterpret_cast<float*>(ring);              // [8 row groups][2][128] floats = 8 KiB of the wave's 12
        const int cq = lane & 7, rg = lane >> 3;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            *reinterpret_cast<float4*>(part + (rg * 2 + 0) * WS_BN + j * 32 + cq * 4) = make_float4(ssum[j][0], ssum[j][1], ssum[j][2], ssum[j][3]);
            *reinterpret_cast<float4*>(part + (rg * 2 + 1) * WS_BN + j * 32 + cq * 4) = make_float4(ssq[j][0], ssq[j][1], ssq[j][2], ssq[j][3]);
        }
        float* st = p.stats + (int64_t)(slot * WS_WAVES + wave) * 2 * p.Nout;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int col = lane * 2 + c;
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int r = 0; r < 8; r++) { sm += part[(r * 2 + 0) * WS_BN + col]; sq += part[(r * 2 + 1) * WS_BN + col]; }
            if (n0 + col < p.Nout) { st[n0 + col] = sm; st[p.Nout + n0 + col] = sq; }
        }
    }
}

// RYOLO_GEMM_WS: 0 off, 1 grids with >= 6 tiles per wave (DEFAULT since the end of r04), 2 every eligible grid (parity tests on small grids).
// It was off through r03 and most of r04 because of what the measurements said then: isolated on random data the kernel is 11-21 % faster than
// the generic one on every K <= 256 layer shape of yolov7 (tools/bench_conv.py), but inside the training plan those launches run at
// 4.3-5.2 TB/s on either kernel (HBM-bound; the sum over the 16 eligible launches is 3.84 vs 3.97 ms) and a workgroup that owns a whole CU
// (160 KiB LDS) cannot share it with the weight-gradient stream: the step is 0.4 % SLOWER (770.1 vs 773.4 img/s, alternating same-box
// runs); inference at batch 64 gains 1 %.
static int ws_mode()
{
    // on by size since the end of r04: with 512 / 256 weight-gradient workgroups on the side stream (conv.hip, conv3x3.hip) the step gains 0.25 %
    // from it (three alternating same-box comparisons); with 768 / 512 it lost 0.4 % (above)
    static const int m = getenv("RYOLO_GEMM_WS") ? atoi(getenv("RYOLO_GEMM_WS")) : 1;
    return m;
}

static bool ws1_geometry_1x1(const ConvGemmParams& p, Ws1Geom& g);
// the space-to-depth data gradient of a 32-input-channel stride-2 layer (see WsTaps): RYOLO_GEMM_WS_S2D = 0 keeps it on the generic kernel
static bool ws1_s2d_eligible(const ConvGemmParams& p)
{
    static const bool on = !(getenv("RYOLO_GEMM_WS_S2D") && atoi(getenv("RYOLO_GEMM_WS_S2D")) == 0);
    if (!on || p.s2d_cin != 32 || p.Nout != 128 || p.nclasses != 1 || p.oh_mul != 2 || p.ow_mul != 2 || p.cls[0].oh_add || p.cls[0].ow_add) return false;
    const int nt = p.cls[0].ntaps;
    if (nt < 1 || nt > 4 || p.Cin % 32 || nt * (p.Cin / 32) > 8 || nt * (p.Cin / 32) < 2) return false;
    if (p.sh != 1 || p.sw != 1 || p.IH != p.OH || p.IW != p.OW || p.OHf != 2 * p.OH || p.OWf != 2 * p.OW) return false;
    if (p.pool_idx || (p.epi != EPI_RAW && p.epi != EPI_ACCUM) || p.ldA % 8 || p.ldC % 8) return false;
    if ((int64_t)p.NB * p.OH * p.OW >= (1ll << 31) || (int64_t)(p.IW + 1) * p.ldA >= (1ll << 30)) return false;
    return true;
}

bool ws1_geometry(const ConvGemmParams& p, Ws1Geom& g)
{
    g.ok = 0;
    g.s2d = 0;
    if ((p.pipe & 0xff) == 1 && p.zeros && ws1_s2d_eligible(p)) {
        const int64_t tiles = ry_cdiv((int64_t)p.NB * p.OH * p.OW, WS_TM);
        static const int cus = getenv("RYOLO_GEMM_WS_WGS") ? atoi(getenv("RYOLO_GEMM_WS_WGS")) : 256;
        g.nk = p.cls[0].ntaps * (p.Cin / 32);
        g.gridN = 1;
        g.wgn = (int)(tiles < (int64_t)cus * WS_WAVES ? ry_cdiv(tiles, WS_WAVES) : cus);
        g.lds_bytes = (unsigned)((g.nk * WS_BN * 32 + WS_WAVES * WS_RING) * sizeof(bf16_t));
        g.stats_rows = g.wgn * WS_WAVES;
        g.s2d = 1;
        g.ok = 1;
        return true;
    }
    return ws1_geometry_1x1(p, g);
}

static bool ws1_geometry_1x1(const ConvGemmParams& p, Ws1Geom& g)
{
    g.ok = 0;
    g.pool = 0;
    // pointwise data gradients with a fused MaxPool gradient run here by default (RYOLO_GEMM_WS_POOL = 0: generic kernel); plain ones by RYOLO_GEMM_WS
    static const bool pool_on = !(getenv("RYOLO_GEMM_WS_POOL") && atoi(getenv("RYOLO_GEMM_WS_POOL")) == 0);
    const bool pool = p.pool_idx != nullptr;
    if (pool && (!pool_on || !p.pool_dz || p.pool_ld % 8 || p.pool_ldi % 8 || (p.OH & 1) || (p.OW & 1) || (p.epi != EPI_RAW && p.epi != EPI_ACCUM) ||
                 (int64_t)p.NB * p.OH * p.OW >= (1ll << 31)))
        return false;
    const int mode = pool ? (ws_mode() == 2 ? 2 : 1) : ws_mode();
    if (!mode || (p.pipe & 0xff) != 1 || !p.zeros) return false;
    if (p.nclasses != 1 || p.cls[0].ntaps != 1 || p.cls[0].dh[0] || p.cls[0].dw[0] || p.cls[0].widx[0] || p.cls[0].oh_add || p.cls[0].ow_add) return false;
    if (p.sh != 1 || p.sw != 1 || p.IH != p.OH || p.IW != p.OW || p.oh_mul != 1 || p.ow_mul != 1 || p.OHf != p.OH || p.OWf != p.OW) return false;
    if (p.s2d_cin) return false;
    if (p.epi != EPI_RAW && p.epi != EPI_STATS && p.epi != EPI_AFFINE_ACT && p.epi != EPI_ACCUM) return false;
    if (p.Cin % 32 || p.Cin > 256 || p.Cin < 64 || p.Nout % 8 || p.ldA % 8 || p.ldC % 8) return false;
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    const int64_t tiles = ry_cdiv(M, WS_TM);
    g.nk = p.Cin / 32;
    g.gridN = (int)ry_cdiv(p.Nout, WS_BN);
    static const int cus = getenv("RYOLO_GEMM_WS_WGS") ? atoi(getenv("RYOLO_GEMM_WS_WGS")) : 256;     // one workgroup per CU
    if (g.gridN > cus) return false;
    g.wgn = cus / g.gridN;
    if (tiles < (int64_t)g.wgn * WS_WAVES) {                       // fewer tiles than waves: shrink the grid (forced mode), else not eligible
        if (mode != 2) return false;
        g.wgn = (int)ry_cdiv(tiles, WS_WAVES);
    } else if (mode != 2 && tiles < 6ll * g.wgn * WS_WAVES) return false;      // < 6 tiles per wave: the tail quantisation costs more than persistence buys
    g.lds_bytes = (unsigned)((g.nk * WS_BN * 32 + WS_WAVES * WS_RING) * sizeof(bf16_t));
    g.stats_rows = g.wgn * WS_WAVES;
    g.pool = pool ? 1 : 0;
    g.ok = 1;
    return true;
}

int ws1_launch(const ConvGemmParams& p, const Ws1Geom& g, hipStream_t stream)
{
    static RyLdsAttr attr[8];
    WsTaps tk = {};
    auto magic = [](unsigned d, unsigned& m, unsigned& sh) {         // n / d == mulhi(n, m) >> sh for 0 <= n < 2^31 (m == 0: d == 1)
        if (d < 2) { m = 0; sh = 0; return; }
        unsigned l = 0;
        while ((1ull << l) < d) l++;
        m = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
        sh = l - 1;
    };
    if (g.s2d || g.pool) {
        magic((unsigned)(p.OH * p.OW), tk.m_img, tk.s_img);
        magic((unsigned)p.OW, tk.m_row, tk.s_row);
    }
    if (g.s2d) {
        const TapClass& tc = p.cls[0];
        const int cch = p.Cin / 32;
        tk.ntaps = tc.ntaps;
        for (int t = 0; t < tc.ntaps; t++) { tk.dh[t] = tc.dh[t]; tk.dw[t] = tc.dw[t]; }
        for (int k = 0; k < g.nk; k++) {
            const int t = k / cch, c = k - t * cch;
            tk.koff[k] = (tc.dh[t] * p.IW + tc.dw[t]) * p.ldA + c * 32;
            tk.wkoff[k] = tc.widx[t] * p.Cin + c * 32;
            tk.ktap[k] = t;
        }
    }
    auto go = [&](auto kern, RyLdsAttr& at) -> int {
        if (int rc = ry_max_dynamic_lds(at, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
        hipLaunchKernelGGL(kern, dim3(g.gridN * g.wgn), dim3(512), g.lds_bytes, stream, p, g.nk, g.gridN, g.wgn, tk);
        RY_CHECK_LAUNCH();
        return RY_OK;
    };
    if (g.pool) return p.epi == EPI_ACCUM ? go(&gemm1x1_ws_kernel<EPI_ACCUM, false, true>, attr[7]) : go(&gemm1x1_ws_kernel<EPI_RAW, false, true>, attr[6]);
    if (g.s2d) return p.epi == EPI_ACCUM ? go(&gemm1x1_ws_kernel<EPI_ACCUM, true>, attr[5]) : go(&gemm1x1_ws_kernel<EPI_RAW, true>, attr[4]);
    switch (p.epi) {
    case EPI_RAW: return go(&gemm1x1_ws_kernel<EPI_RAW, false>, attr[0]);
    case EPI_STATS: return go(&gemm1x1_ws_kernel<EPI_STATS, false>, attr[1]);
    case EPI_AFFINE_ACT: return go(&gemm1x1_ws_kernel<EPI_AFFINE_ACT, false>, attr[2]);
    case EPI_ACCUM: return go(&gemm1x1_ws_kernel<EPI_ACCUM, false>, attr[3]);
    }
    return RY_ERR_ARG;
}
