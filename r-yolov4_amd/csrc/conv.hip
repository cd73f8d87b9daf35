// Implicit-GEMM convolution for gfx950 (MI355X): forward, data-gradient and weight-gradient of every nn.Conv2d of
// the reference's conv stack (model/utils.py:6-32 `Conv`, :189-215 `RepConv`;
// 3x3 s1/s2 and 1x1 s1, plus 6x6 s2 / Cin=3 first layers through an explicit-im2col front end).
//
// MI355X-first design (not a cuDNN call, not NCHW):
//   * activations are NHWC bf16 with an explicit channel stride (`ld`), so a torch.cat along channels
//     (model/utils.py:64,95,118,143,160,241,261,280; neck.py) is just a channel offset into one buffer — producers write
//     their slice, consumers read theirs, no concat pass;
//   * im2col-free: the K loop walks (tap, 32-channel chunk); each A-tile row is one output pixel whose 64 contiguous
//     bytes are gathered (zero-filled at the border) straight from HBM into a swizzled LDS tile;
//   * 64-wide wavefronts on v_mfma_f32_32x32x16_bf16 (fp32 accumulate); 4 waves per workgroup, each owning a
//     (BM/WM)x(BN/WN) block of 32x32 MFMA tiles; LDS double buffered, one barrier per K step; 16-byte ds_read_b128
//     fragment reads are bank-conflict free through slot ^= (row>>2)&3;
//   * the same kernel serves the data gradient: stride-1 dgrad is a correlation with mirrored taps and [Cin][tap][Cout]
//     weights; stride-2 dgrad is decomposed into the 4 output-parity classes (1/2/2/4 live taps instead of 9, so no MFMA
//     is spent on structural zeros), selected by blockIdx.z;
//   * epilogues fuse what the reference runs as separate passes: training BatchNorm statistics (per-workgroup column
//     sums of y and y^2, deterministic two-level reduction, no atomics), eval-mode folded BN + activation, bias + fp32
//     head output, and gradient accumulation for tensors with several consumers;
//   * XCD-aware workgroup order: each XCD (private 4 MiB L2) gets a contiguous range of output tiles, n-tile fastest,
//     so the A tile gathered by one workgroup is re-used from L2 by its n-neighbours.
//   * weight gradient: K = output pixels (split over blockIdx.z, fp32 atomics into the torch-layout .grad),
//     both operands staged pixel-major and read transposed from LDS.
#include "conv_internal.h"
#include <type_traits>
#include <stdlib.h>

template <int N> __device__ __forceinline__ void gemm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

#define BK 32

#ifndef RY_STAGES
#define RY_STAGES 3
#endif

// T1: single-tap launches on the identity grid with input grid == output grid, stride 1, tap (0, 0), weight index 0 — every 1x1 forward and
// 1x1 stride-1 data gradient (the launcher checks it; requires ID and the LDS-DMA ring).  Row m of the GEMM IS input pixel m, so the tile
// needs no (image, row, column) decomposition, no tap table in LDS (and not the workgroup barrier behind it), no bounds tests and no
// per-stage scalar tap lookups: on an 8-32 step K loop that prologue was as long as the loop (tools/gemm_phases.py).
// NST: ring depth override (0 = RY_STAGES, 2 for 64-channel stages).  A workgroup ALONE on its CU (grids of <= 256 tiles: the low-resolution
// layers at 8 images, batch-1 inference) is latency-bound on a 3-stage ring — ~1.5 stages in flight per ~1.3 us round trip to L2 / Infinity
// Cache = 1600 cycles per K step — where three co-resident workgroups keep 4.5 in flight; with the LDS to itself it takes a 6-deep ring
// (4-deep for <= 512 tiles, two per CU).
template <int BM, int BN, int WM, int WN, int PIPE, int KB = 32, int EP = 0, bool ID = false, bool T1 = false, int NST = 0>
__global__ __launch_bounds__(256, NST >= 5 ? 1 : 2) void conv_gemm_kernel(const ConvGemmParams p)
{
    static_assert(!T1 || (ID && PIPE == 1), "T1 is an identity-grid LDS-DMA instantiation");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int PA = (BM * 4 + 255) / 256, PB = (BN * 4 + 255) / 256;       // 16-byte pieces per thread
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && BM % 64 == 0, "tile config");
    constexpr int WTM = BM / WM, WTN = BN / WN;               // rows x cols of one wave's output block
    constexpr int EP_LD = WTN + 8;                            // staging row stride (bf16), 16-byte aligned, breaks bank aliasing
    static_assert(KB == 32 || (KB == 64 && PIPE == 1), "64-channel stages exist for the flat LDS-DMA ring only");
    constexpr int NSTG = NST ? NST : (KB == 64 ? 2 : RY_STAGES);   // 64-channel stages are twice as large: 2-deep ring, same LDS
    constexpr int MAINLOOP_ELEMS = (PIPE ? NSTG : 2) * (BM + BN) * KB, EPI_STAGE = 4 * WTM * EP_LD;
    constexpr int EPI_ELEMS = EPI_STAGE + 4 * BN;             // output staging + the tile's per-column coefficients (2 x BN floats)
    constexpr int TAPTAB = 64;                                // 32 ints after the tiles: per-tap (dh, dw, widx) for the DMA loop
    __shared__ __attribute__((aligned(16))) bf16_t smem[(MAINLOOP_ELEMS > EPI_ELEMS ? MAINLOOP_ELEMS : EPI_ELEMS) + TAPTAB];
#define sA_(b) (smem + (b) * (BM + BN) * BK)
#define sB_(b) (smem + (b) * (BM + BN) * BK + BM * BK)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
#ifdef GEMM_TIMING
    const unsigned long long T0 = __builtin_readcyclecounter();
    unsigned long long T1 = 0, T2 = 0, T3 = 0;
#endif
    // Tap classes (the four output-parity classes of a stride-2 data gradient) read the SAME dY rows.  With the class on blockIdx.z the dispatcher
    // finishes class 0 over the whole tensor before class 1 begins and dY comes from HBM once per class (PMC r04: 3.97 GB for 1.97 GB of operands
    // on the 128 -> 64 layer).  r05, class-chunked order (host: launch_gemm puts the chunk length / 16 into bits 16-27 of `pipe` and launches a
    // 1-D grid): the linear block id walks chunk by chunk, inside a chunk class by class, inside a class tile by tile — thousands of
    // workgroups of ONE class run at a time (the classes have 1 / 2 / 2 / 4 taps: interleaving them tile by tile, tried in r04, left them
    // unbalanced and was slower), and a chunk's dY (tens of MB) is still in the 256 MiB Infinity Cache when the next class asks for it.
    int cls_i = blockIdx.z, tile_i;
    if (const int cht = (p.pipe >> 16) & 0xfff) {
        const int ncls = p.nclasses, T = (int)gridDim.x / ncls, CH = cht * 16;
        const int L = blockIdx.x;
        const int chunk = L / (CH * ncls);
        const int base = chunk * CH;
        const int nin = min(CH, T - base);
        const int within = L - chunk * CH * ncls;
        cls_i = within / nin;
        tile_i = base + xcd_remap(within - cls_i * nin, nin);
    } else {
        tile_i = xcd_remap(blockIdx.x, gridDim.x);
    }
    const TapClass& tc = p.cls[cls_i];
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    const int gridN = (p.Nout + BN - 1) / BN;
    const int tile = tile_i;
    const int mb = tile / gridN, nb = tile - mb * gridN;
    const int64_t m0 = (int64_t)mb * BM;
    const int n0 = nb * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const int cchunks = p.Cin / KB;
    const int nk = T1 ? cchunks : tc.ntaps * cchunks;
    if constexpr (PIPE == 0) {
        // ---- per-thread gather bookkeeping -----------------------------------------------------------------
        int a_ih0[PA], a_iw0[PA];
        int64_t a_base[PA];
        bool a_ok[PA];
    #pragma unroll
        for (int u = 0; u < PA; u++) {
            const int r = (tid >> 2) + u * 64;
            const int64_t m = m0 + r;
            a_ok[u] = m < M;
            const int64_t mm = a_ok[u] ? m : 0;
            const int img = (int)(mm / ((int64_t)p.OH * p.OW));
            const int rem = (int)(mm - (int64_t)img * p.OH * p.OW);
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_ih0[u] = oh * p.sh;
            a_iw0[u] = ow * p.sw;
            a_base[u] = (int64_t)img * p.IH * p.IW;
        }
        const int slot = tid & 3;

        uint4 ra[PA], rb[PB];
        auto gload = [&](int step) {
            const int t = step / cchunks;
            const int c0 = (step - t * cchunks) * BK + slot * 8;
            const int dh = tc.dh[t], dw = tc.dw[t], wi = tc.widx[t];
    #pragma unroll
            for (int u = 0; u < PA; u++) {
                const int ih = a_ih0[u] + dh, iw = a_iw0[u] + dw;
                const bool ok = a_ok[u] && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) v = *reinterpret_cast<const uint4*>(p.A + (a_base[u] + (int64_t)ih * p.IW + iw) * p.ldA + c0);
                ra[u] = v;
            }
    #pragma unroll
            for (int u = 0; u < PB; u++) {
                const int rr = (tid >> 2) + u * 64;
                const int n = n0 + rr;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (rr < BN && n < p.Nout) v = *reinterpret_cast<const uint4*>(p.W + ((int64_t)n * p.wtaps + wi) * p.Cin + c0);
                rb[u] = v;
            }
        };
        auto sstore = [&](int buf) {
    #pragma unroll
            for (int u = 0; u < PA; u++) {
                const int r = (tid >> 2) + u * 64;
                *reinterpret_cast<uint4*>(sA_(buf) + (r * 4 + (slot ^ ((r >> 2) & 3))) * 8) = ra[u];
            }
    #pragma unroll
            for (int u = 0; u < PB; u++) {
                const int r = (tid >> 2) + u * 64;
                if (r < BN) *reinterpret_cast<uint4*>(sB_(buf) + (r * 4 + (slot ^ ((r >> 2) & 3))) * 8) = rb[u];
            }
        };

        gload(0);
        sstore(0);
        __syncthreads();
        for (int k = 0; k < nk; k++) {
            const int buf = k & 1;
            if (k + 1 < nk) gload(k + 1);
    #pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                bf16x8 af[TM], bfr[TN];
                const int sl = ks * 2 + (lane >> 5);
    #pragma unroll
                for (int i = 0; i < TM; i++) {
                    const int r = wm * (BM / WM) + i * 32 + (lane & 31);
                    af[i] = *reinterpret_cast<const bf16x8*>(sA_(buf) + (r * 4 + (sl ^ ((r >> 2) & 3))) * 8);
                }
    #pragma unroll
                for (int j = 0; j < TN; j++) {
                    const int r = wn * (BN / WN) + j * 32 + (lane & 31);
                    bfr[j] = *reinterpret_cast<const bf16x8*>(sB_(buf) + (r * 4 + (sl ^ ((r >> 2) & 3))) * 8);
                }
    #pragma unroll
                for (int i = 0; i < TM; i++)
    #pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
            if (k + 1 < nk) sstore(buf ^ 1);
            __syncthreads();
        }


    } else if constexpr (PIPE == 1) {
        // ---- LDS-DMA ring: NSTG stages of (BM + BN) x KB bf16, filled by global_load_lds_dwordx4 (no VGPR staging) ----
        // One wave instruction moves 1 KiB = RPP rows x (2*KB) bytes into a lane-linear LDS image, so the bank swizzle of the
        // fragment reads is applied on the SOURCE side.  KB = 64 fetches whole 128-byte lines per pixel row (the KB = 32 gather
        // issues two 64-byte half-line requests per line and was L2/TA request-rate bound: ablation in profiles/ + DESIGN.md).
        //   KB = 32: 4 slots/row, phys = slot ^ ((row >> 2) & 3);   KB = 64: 8 slots/row, phys = slot ^ ((row >> 1) & 7)
        // the tap class of this launch: 40 kernarg bytes as ten dwords, requested first so that their round trip overlaps the index math
        const unsigned* tcw = reinterpret_cast<const unsigned*>(&tc);
        unsigned tw[10];
        if constexpr (!T1) {
#pragma unroll
            for (int i = 0; i < 10; i++) tw[i] = tcw[i];
        }
        constexpr int SPR = KB / 8;                             // 16-byte slots per row
        constexpr int RPP = 64 / SPR;                           // rows per 1-KiB piece
        constexpr int STG = (BM + BN) * KB;                     // elements per stage
        constexpr int PCS_A = BM / RPP, PCS_B = BN / RPP;       // pieces per stage
        constexpr int NPA = (PCS_A + 3) / 4, NPB = (PCS_B + 3) / 4;   // pieces per wave
        auto swz = [](int row) { return KB == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
        int a_ih0[NPA], a_iw0[NPA];
        const bf16_t* a_ptr[NPA];
        bool a_ok[NPA];
        // (img, oh, ow) of the tile's first row with ONE wave-uniform 64-bit division; rows inside the tile (< 256 further) by
        // small exact reciprocal divisions — the per-lane 64-bit divisions of the first version cost ~3k cycles of an 8-step K loop
        int t_img = 0, t_oh = 0, t_ow = 0;
        float rOW = 0.f, rOH = 0.f;
        if constexpr (!T1) {
            const int64_t HWo = (int64_t)p.OH * p.OW;
            t_img = (int)(m0 / HWo);
            const int t_rem = (int)(m0 - (int64_t)t_img * HWo);
            t_oh = t_rem / p.OW;
            t_ow = t_rem - t_oh * p.OW;
            rOW = 1.0f / (float)p.OW;
            rOH = 1.0f / (float)p.OH;
        }
#pragma unroll
        for (int u = 0; u < NPA; u++) {
            const int piece = wave + 4 * u;
            const int r = piece * RPP + lane / SPR;
            const int64_t m = m0 + r;
            a_ok[u] = piece < PCS_A && m < M;
            if constexpr (T1) {                                   // GEMM row m = input pixel m
                a_ih0[u] = 0;
                a_iw0[u] = 0;
                a_ptr[u] = p.A + (a_ok[u] ? m : 0) * p.ldA + ((lane % SPR) ^ swz(r)) * 8;
                continue;
            }
            const int o = t_ow + r;
            const int wr = small_div(o, p.OW, rOW);
            const int ow = o - wr * p.OW;
            const int orow = t_oh + wr;
            const int wi = small_div(orow, p.OH, rOH);
            const int oh = orow - wi * p.OH;
            const int img = a_ok[u] ? t_img + wi : 0;
            a_ih0[u] = oh * p.sh;
            a_iw0[u] = ow * p.sw;
            a_ptr[u] = p.A + ((int64_t)img * p.IH * p.IW + (int64_t)(a_ok[u] ? a_ih0[u] : 0) * p.IW + (a_ok[u] ? a_iw0[u] : 0)) * p.ldA + ((lane % SPR) ^ swz(r)) * 8;
        }
        const bf16_t* b_ptr[NPB];
        bool b_ok[NPB];
#pragma unroll
        for (int u = 0; u < NPB; u++) {
            const int piece = wave + 4 * u;
            const int r = piece * RPP + lane / SPR;
            b_ok[u] = piece < PCS_B && (n0 + r) < p.Nout;
            b_ptr[u] = p.W + (int64_t)(n0 + r) * p.wtaps * p.Cin + ((lane % SPR) ^ swz(r)) * 8;
        }
        // tap table -> LDS (ONE __shared__ object; an ordinary VMEM load inside the loop would make hipcc drain vmcnt(0))
        int* taptab = reinterpret_cast<int*>(smem + (MAINLOOP_ELEMS > EPI_ELEMS ? MAINLOOP_ELEMS : EPI_ELEMS));
        // constant tap index -> the bytes come from scalar dword loads of the kernarg segment; `tc.dh[tid]` made every thread fetch
        // its byte with a VMEM load and the workgroup wait a full memory round trip before its first DMA could be issued
        // ... and ALL of them are requested before the first LDS write: scalar loads and LDS writes share one counter (lgkmcnt), so
        // "load tap t, write tap t" in a loop made every tap wait for its own kernarg round trip — 5 000 of the 5 400 prologue cycles
        // of a workgroup (cycle counters, tools/gemm_phases.py).  The 40-byte TapClass is fetched as ten dwords, bytes are cut out below.
        static_assert(sizeof(TapClass) == 40 && offsetof(TapClass, dh) == 12 && offsetof(TapClass, dw) == 12 + RY_MAX_TAPS &&
                      offsetof(TapClass, widx) == 12 + 2 * RY_MAX_TAPS, "TapClass layout");
        if constexpr (!T1) {
#pragma unroll
            for (int t = 0; t < RY_MAX_TAPS; t++) {
                const unsigned bdh = (tw[(12 + t) >> 2] >> (((12 + t) & 3) * 8)) & 0xffu;
                const unsigned bdw = (tw[(21 + t) >> 2] >> (((21 + t) & 3) * 8)) & 0xffu;
                const unsigned bwi = (tw[(30 + t) >> 2] >> (((30 + t) & 3) * 8)) & 0xffu;
                if (tid == t && t < (int)tw[0]) taptab[t] = (int)(bdh | (bdw << 8) | (bwi << 16));
            }
            __syncthreads();
        }
        int is_t = 0, is_c0 = 0;                                  // (tap, channel chunk) of the next stage to issue
        int cur_dh = 0, cur_dw = 0;
        int64_t cur_a_off = 0, cur_b_off = 0;
        bf16_t* cur_stage = smem;
        auto issue_prep = [&](int step) {                         // scalar part of a stage issue
            if constexpr (T1) {                                   // one tap (0, 0), weight index 0: the stage is a channel offset
                cur_a_off = cur_b_off = (int64_t)step * KB;
                cur_stage = smem + (step % NSTG) * STG;
                return;
            }
            const int packed = __builtin_amdgcn_readfirstlane(taptab[is_t]);         // wave-uniform -> scalar registers
            cur_dh = (int)(signed char)(packed & 0xff);
            cur_dw = (int)(signed char)((packed >> 8) & 0xff);
            const int wi = (packed >> 16) & 0xff;
            const int c0 = is_c0;
            is_c0 += KB;
            if (is_c0 >= p.Cin) { is_c0 = 0; is_t++; }
            cur_a_off = ((int64_t)cur_dh * p.IW + cur_dw) * p.ldA + c0;
            cur_b_off = (int64_t)wi * p.Cin + c0;
            cur_stage = smem + (step % NSTG) * STG;
        };
        auto issue_item = [&](int it) {                           // one 1-KiB DMA piece: it < NPA -> A piece, else B piece
            if (it < NPA) {
                const int u = it;
                const int piece = wave + 4 * u;
                if (piece < PCS_A) {                                          // wave-uniform
                    const bool ok = T1 ? a_ok[u] : (a_ok[u] && (unsigned)(a_ih0[u] + cur_dh) < (unsigned)p.IH && (unsigned)(a_iw0[u] + cur_dw) < (unsigned)p.IW);
                    const bf16_t* src = ok ? a_ptr[u] + cur_a_off : p.zeros;
                    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(cur_stage + piece * 512), 16, 0, 0);
                }
            } else {
                const int u = it - NPA;
                const int piece = wave + 4 * u;
                if (piece < PCS_B) {
                    const bf16_t* src = b_ok[u] ? b_ptr[u] + cur_b_off : p.zeros;
                    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(cur_stage + BM * KB + piece * 512), 16, 0, 0);
                }
            }
        };
        constexpr int LPS = NPA + NPB;                                        // DMA instructions per wave per stage (upper bound)
        // vmcnt counts ISSUED instructions per wave.  When the piece count of a stage is not a multiple of 4 (BN = 32: two weight
        // pieces for four waves) some waves issue fewer DMA instructions per stage than others, and "at most LPS outstanding" would let
        // such a wave run ahead with one piece of the CURRENT stage still in flight (seen as a handful of garbage rows in 41 M: the
        // layer-2 data gradient at batch 64).  Each wave therefore waits on its OWN per-stage count.
        constexpr int LPS_LO = PCS_A / 4 + PCS_B / 4;                         // pieces every wave issues
        constexpr int XTR = LPS - LPS_LO;                                     // 0, 1 or 2 wave-dependent extra pieces
        const int extra = __builtin_amdgcn_readfirstlane((wave < PCS_A % 4 ? 1 : 0) + (wave < PCS_B % 4 ? 1 : 0));
        auto wait_inflight = [&](auto stages) {                               // allow `stages` later stages to stay in flight
            constexpr int S = decltype(stages)::value;
            if constexpr (XTR == 0) { gemm_wait_vm<S * LPS>(); }
            else {
                if (extra == XTR) gemm_wait_vm<S * LPS>();
                else if (XTR == 2 && extra == 1) gemm_wait_vm<S * (LPS_LO + 1)>();
                else gemm_wait_vm<S * LPS_LO>();
            }
        };
#pragma unroll
        for (int st = 0; st < NSTG - 1; st++)
            if (st < nk) {
                issue_prep(st);
#pragma unroll
                for (int it = 0; it < LPS; it++) issue_item(it);
            }
#ifdef GEMM_TIMING
        T1 = __builtin_readcyclecounter();
#endif
        constexpr int KS = KB / 16, NMF = KS * TM * TN;
        for (int k = 0; k < nk; k++) {
            // stages allowed to stay in flight while stage k is consumed
            const int pend = min(NSTG - 2, nk - 1 - k);
            if (NSTG >= 6 && pend >= 4) wait_inflight(std::integral_constant<int, 4>{});
            else if (NSTG >= 5 && pend == 3) wait_inflight(std::integral_constant<int, 3>{});
            else if (NSTG >= 4 && pend == 2) wait_inflight(std::integral_constant<int, 2>{});
            else if (NSTG >= 3 && pend == 1) wait_inflight(std::integral_constant<int, 1>{});
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                     // stage k visible to all waves; stage k-1 fully consumed
            const bool do_issue = k + NSTG - 1 < nk;
            if (do_issue) issue_prep(k + NSTG - 1);
            const bf16_t* sa = smem + (k % NSTG) * STG;
            const bf16_t* sb = sa + BM * KB;
            // every fragment of the stage is read up front into its own registers (hipcc otherwise re-uses one register set per
            // 16-channel sub-step and exposes an LDS round trip between the MFMA groups), then the MFMAs run with the next stage's
            // DMA instructions spread in their shadow (an LDS-DMA issue next to ds_reads costs 100-185 cycles, guide price table)
            bf16x8 af[KS][TM], bfr[KS][TN];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const int sl = ks * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    const int r = wm * (BM / WM) + i * 32 + (lane & 31);
                    af[ks][i] = *reinterpret_cast<const bf16x8*>(sa + (r * SPR + (sl ^ swz(r))) * 8);
                }
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const int r = wn * (BN / WN) + j * 32 + (lane & 31);
                    bfr[ks][j] = *reinterpret_cast<const bf16x8*>(sb + (r * SPR + (sl ^ swz(r))) * 8);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            int issued = 0;
#pragma unroll
            for (int q = 0; q < NMF; q++) {
                const int ks = q / (TM * TN), i = (q / TN) % TM, j = q % TN;
                // A operand = weights, B operand = pixels: the accumulator holds the transposed tile (see the epilogue)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int it = 0; it < LPS; it++)
                    if (it == issued && it * NMF < (q + 1) * LPS) {           // compile-time after unrolling: spread evenly
                        __builtin_amdgcn_sched_barrier(0);
                        if (do_issue) issue_item(it);
                        __builtin_amdgcn_sched_barrier(0);
                        issued++;
                    }
            }
        }
        __syncthreads();
    }

#ifdef GEMM_TIMING
    T2 = __builtin_readcyclecounter();
#endif
    // ---- epilogue ----------------------------------------------------------------------------------------
    // The MFMAs ran with A = weights, B = pixels, so acc[i][j] is the TRANSPOSED 32x32 tile: column = lane & 31 = pixel
    // i*32 + (lane & 31), row = channel j*32 + (e & 3) + 8*(e >> 2) + 4*(lane >> 5).  A lane therefore owns 4 consecutive
    // channels of ONE pixel per register quad: one 8-byte LDS store (or one 16-byte fp32 store) instead of four 2-byte ones.
    // ID: the launcher has checked that the output grid is the iteration grid and that neither the fused pool gradient nor the
    // depth-to-space store is asked for — the common case (every forward launch, every stride-1 data gradient).  Its store loops carry no
    // index arithmetic and none of the rare branches: in the one-size-fits-all loop every row iteration walked ~400 instructions of
    // uniform branches, 64-bit divisions and waits (3 300 cycles per workgroup for eight 16-byte stores per lane).
    const bool identity = ID || (p.oh_mul == 1 && p.ow_mul == 1 && p.OHf == p.OH && p.OWf == p.OW && tc.oh_add == 0 && tc.ow_add == 0);
    const int h = lane >> 5;
    // Output pixel of tile row rt (GEMM row m0 + rt) on a NON-identity grid (strided data gradients: parity classes, depth-to-space): the
    // tile's first row is decomposed once per workgroup (wave-uniform 64-bit division), rows inside the tile by small exact reciprocal
    // divisions — the per-row 64-bit divisions this replaces cost ~10 k cycles of a store loop (every stride-2 data gradient;
    // 64->128 taps4 @400^2: the slowest launch of the step).
    int e_img = 0, e_oh = 0, e_ow = 0;
    float e_rOW = 0.f, e_rOH = 0.f;
    if (!ID && !identity) {
        const int64_t HWo = (int64_t)p.OH * p.OW;
        e_img = (int)(m0 / HWo);
        const int rem = (int)(m0 - (int64_t)e_img * HWo);
        e_oh = rem / p.OW;
        e_ow = rem - e_oh * p.OW;
        e_rOW = 1.0f / (float)p.OW;
        e_rOH = 1.0f / (float)p.OH;
    }
    auto out_pixel = [&](int rt, int64_t m) -> int64_t {
        if constexpr (ID) return m;
        if (identity) return m;
        const int o = e_ow + rt;
        const int wr = small_div(o, p.OW, e_rOW);
        const int ow = o - wr * p.OW, orow = e_oh + wr;
        const int wi = small_div(orow, p.OH, e_rOH);
        const int oh = orow - wi * p.OH;
        return ((int64_t)(e_img + wi) * p.OHf + (oh * p.oh_mul + tc.oh_add)) * p.OWf + (ow * p.ow_mul + tc.ow_add);
    };

    if constexpr (EP == 3) {
        // Detection head written in its FINAL layout (r05, ConvGemmParams.head_attrs): column n = anchor a, attribute e of GEMM row (image b, cell)
        // goes to out[b][a][cell][e] — the view / permute of model/yololayer.py:25 — after the bias and ImplicitM (p.scale; model/neck.py:186), with
        // the objectness logit of every (anchor, cell) copied to the compact array p.stats for the fused loss.  Replaces the row-major fp32
        // intermediate and the pass that re-read it (ryolo_head_finish_fwd).  Each wave stages 32 pixels x its 64 columns in LDS (fp32) and writes
        // them out ANCHOR BY ANCHOR: for one anchor, consecutive cells are consecutive attrs-float rows, so an anchor that lies inside the wave's
        // columns is one contiguous run of 32 * attrs floats.  Measured on the 100^2 head (64 images, K = 256; tools/bench_head.py): row-major GEMM
        // 540 us + finish pass 470 us -> 583 us in one launch; direct stores from the accumulator layout (16 bytes per lane, 4 * attrs bytes apart)
        // 830 us; and with bias / ImplicitM fetched per element from global memory instead of from LDS 1 000 us.
        typedef float hf4 __attribute__((ext_vector_type(4), aligned(4)));
        static_assert(WTN == 64, "head epilogue: 64 columns per wave");
        constexpr int SLD = WTN + 4;                           // (16-byte aligned rows: the accumulator quads are staged as one ds_write_b128 each)
        const int attrs = p.head_attrs, och = p.head_och, na = p.Nout / attrs;
        const int cells = p.OH * p.OW;
        const float rattrs = 1.0f / (float)attrs, rcells = 1.0f / (float)cells;
        float* const outf = reinterpret_cast<float*>(p.out);
        __syncthreads();                                       // every wave is done reading the operand tiles
        float* const stg = reinterpret_cast<float*>(smem) + wave * (32 * SLD);
        // bias and ImplicitM of the tile's 128 columns: once per workgroup into LDS (read per element from global memory they were 128 VMEM
        // instructions per lane and 32-pixel group)
        float* const sbias = reinterpret_cast<float*>(smem) + 4 * 32 * SLD;
        float* const sscale = sbias + BN;
        if (tid < BN) {
            const int n = n0 + tid;
            sbias[tid] = (p.bias && n < p.Nout) ? p.bias[n] : 0.f;
            sscale[tid] = (p.scale && n < p.Nout) ? p.scale[n] : 1.f;
        }
        __syncthreads();
        const int nw0 = n0 + wn * WTN, nw1 = min(nw0 + WTN, p.Nout);
        if (nw0 < p.Nout) {
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int64_t mrow0 = m0 + wm * WTM + i * 32;      // (wave-uniform)
                if (mrow0 >= M) break;
                const int npix = (int)min((int64_t)32, M - mrow0);
#pragma unroll
                for (int jj = 0; jj < TN; jj++)
#pragma unroll
                    for (int g4 = 0; g4 < 4; g4++) {
                        const int c0 = jj * 32 + 8 * g4 + 4 * h;
                        const float4 bq = *reinterpret_cast<const float4*>(sbias + wn * WTN + c0);
                        const float4 sq = *reinterpret_cast<const float4*>(sscale + wn * WTN + c0);
                        float4 v;                                          // (acc + bias) * ImplicitM: the two fp32 operations of the two-pass form
                        v.x = (acc[i][jj][4 * g4 + 0] + bq.x) * sq.x;
                        v.y = (acc[i][jj][4 * g4 + 1] + bq.y) * sq.y;
                        v.z = (acc[i][jj][4 * g4 + 2] + bq.z) * sq.z;
                        v.w = (acc[i][jj][4 * g4 + 3] + bq.w) * sq.w;
                        *reinterpret_cast<float4*>(stg + (lane & 31) * SLD + c0) = v;
                    }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int b0 = (int)(mrow0 / cells), cell0 = (int)(mrow0 - (int64_t)b0 * cells);
                const bool one_img = cell0 + npix <= cells;
                const int a_lo = small_div(nw0, attrs, rattrs), a_hi = small_div(nw1 - 1, attrs, rattrs);
                for (int a = a_lo; a <= a_hi; a++) {
                    const int c_lo = max(a * attrs, nw0), c_hi = min((a + 1) * attrs, nw1);
                    const int len = c_hi - c_lo, e_lo = c_lo - a * attrs, scol = c_lo - nw0;
                    if (one_img && len == attrs) {
                        float* const dst = outf + (((int64_t)b0 * na + a) * cells + cell0) * attrs;
                        const int total = npix * attrs;
                        for (int f0 = lane * 4; f0 < total; f0 += 256) {
                            int pix = small_div(f0, attrs, rattrs), e = f0 - pix * attrs;
                            float v[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                v[k] = f0 + k < total ? stg[pix * SLD + scol + e] : 0.f;
                                if (++e == attrs) { e = 0; pix++; }
                            }
                            if (f0 + 3 < total) { const hf4 w = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<hf4*>(dst + f0) = w; }
                            else for (int k = 0; f0 + k < total; k++) dst[f0 + k] = v[k];
                        }
                    } else {
                        // part of an anchor (it straddles the wave's column range) or pixels of two images: rows of `len` floats; the pixel's
                        // (image, cell) from the group's first one (cell0 + pix < cells + 32: exact small division, no 64-bit arithmetic)
                        const float rlen = 1.0f / (float)len;
                        const int total = npix * len;
                        for (int f = lane; f < total; f += 64) {
                            const int pix = small_div(f, len, rlen), k = f - pix * len;
                            const int wr = small_div(cell0 + pix, cells, rcells);
                            const int b = b0 + wr, cell = cell0 + pix - wr * cells;
                            outf[(((int64_t)b * na + a) * cells + cell) * attrs + e_lo + k] = stg[pix * SLD + scol + k];
                        }
                    }
                    const int oc = a * attrs + och;
                    if (p.stats && oc >= c_lo && oc < c_hi && lane < npix) {
                        const int wr = small_div(cell0 + lane, cells, rcells);
                        const int b = b0 + wr, cell = cell0 + lane - wr * cells;
                        p.stats[((int64_t)b * na + a) * cells + cell] = stg[lane * SLD + oc - nw0];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else if (p.epi == EPI_F32_BIAS) {
        // small outputs (detection heads): direct fp32 stores, 4 consecutive channels per lane
        const bool vec4 = (p.ldC & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int64_t m = m0 + wm * WTM + i * 32 + (lane & 31);
            if (m >= M) continue;
            float* orow = reinterpret_cast<float*>(p.out) + out_pixel(wm * WTM + i * 32 + (lane & 31), m) * p.ldC;
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int n = n0 + wn * WTN + j * 32 + 8 * g4 + 4 * h;
                    if (vec4 && n + 3 < p.Nout) {                             // 16-byte store of the lane's 4 consecutive channels
                        float4 v = make_float4(acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]);
                        if (p.bias) { v.x += p.bias[n]; v.y += p.bias[n + 1]; v.z += p.bias[n + 2]; v.w += p.bias[n + 3]; }
                        *reinterpret_cast<float4*>(orow + n) = v;
                        continue;
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (n + q >= p.Nout) continue;
                        float v = acc[i][j][4 * g4 + q];
                        if (p.bias) v += p.bias[n + q];
                        orow[n + q] = v;
                    }
                }
        }
    } else {
        // bf16 outputs: stage the wave's WTM x WTN block in LDS (8-byte packed stores), then write whole 16-byte row segments
        __syncthreads();                                       // every wave is done reading the operand tiles
        bf16_t* stage = smem + wave * WTM * EP_LD;
        // inference (EPI_AFFINE_ACT): the folded BatchNorm coefficients of the tile's columns, once per workgroup into LDS — fetched per element
        // from global memory they were 8 VMEM instructions per stored quad (the head epilogue above: 1 000 -> 583 us from the same change)
        float* const cscale = reinterpret_cast<float*>(smem + EPI_STAGE);
        float* const cshift = cscale + BN;
        if (p.epi == EPI_AFFINE_ACT) {
            if (tid < BN) {
                const int n = n0 + tid;
                cscale[tid] = n < p.Nout ? p.scale[n] : 0.f;
                cshift[tid] = n < p.Nout ? p.shift[n] : 0.f;
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int c0 = j * 32 + 8 * g4 + 4 * h;
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = acc[i][j][4 * g4 + q];
                    if (p.epi == EPI_AFFINE_ACT) {
                        // folded BatchNorm (running statistics) + activation on the fp32 accumulator (columns >= Nout are never stored)
                        const float4 sc = *reinterpret_cast<const float4*>(cscale + wn * WTN + c0);
                        const float4 sf = *reinterpret_cast<const float4*>(cshift + wn * WTN + c0);
                        const float sc4[4] = {sc.x, sc.y, sc.z, sc.w}, sf4[4] = {sf.x, sf.y, sf.z, sf.w};
                        act_affine_quad(v, sc4, sf4, p.act);
                    }
                    *reinterpret_cast<uint2*>(stage + (i * 32 + (lane & 31)) * EP_LD + c0) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                }
        // (same-wave LDS hand-off: no workgroup barrier needed, only the wave's own ds_write -> ds_read ordering)
        constexpr int CH = WTN / 8;                            // 16-byte chunks per row
        constexpr int RPI = 64 / CH;                           // rows per iteration
        const int ch = lane % CH, r0 = lane / CH;
        const int n = n0 + wn * WTN + ch * 8;
        // EP: 0 plain store loop; 1 accumulate epilogue (EPI_ACCUM): the two-phase loop below, its own instantiation so that its registers do not
        // cost the plain kernels their third resident workgroup (164 -> 204 VGPRs when everything was one kernel).  (EP 2 — BatchNorm-backward sums
        // folded into this store loop, r02 — lost to the stand-alone reduce pass at every size since r04 and was retired in r06.)
        // fused MaxPool2d(2, 2) gradient: (image, row, column) of the tile's first pixel, rows inside the tile by small exact divisions
        int pl_img = 0, pl_oh = 0, pl_ow = 0;
        float pl_rOW = 0.f, pl_rOH = 0.f;
        if (!ID && p.pool_idx) {
            const int64_t HWo = (int64_t)p.OH * p.OW;
            pl_img = (int)(m0 / HWo);
            const int rem = (int)(m0 - (int64_t)pl_img * HWo);
            pl_oh = rem / p.OW;
            pl_ow = rem - pl_oh * p.OW;
            pl_rOW = 1.0f / (float)p.OW;
            pl_rOH = 1.0f / (float)p.OH;
        }
        if constexpr (EP != 0) {
            // Two phases: every global load of the store loop (the old value of an accumulate epilogue) is
            // issued before the first one is used — inside one loop with `continue` branches each iteration paid its own round trip.
            constexpr int NIT = WTM / RPI, GRP = NIT % 4 == 0 ? 4 : (NIT % 2 == 0 ? 2 : 1);
            const bool accum = p.epi == EPI_ACCUM;
            const int n_s = n < p.Nout ? n : 0;
#pragma unroll
            for (int g0 = 0; g0 < NIT; g0 += GRP) {
                bf16_t* ov[GRP];
                bool lv[GRP];
                uint4 oldv[GRP];
#pragma unroll
                for (int k = 0; k < GRP; k++) {
                    const int r = (g0 + k) * RPI + r0;
                    const int64_t m = m0 + wm * WTM + r;
                    lv[k] = m < M && n < p.Nout;
                    const int64_t pix = lv[k] ? out_pixel(wm * WTM + r, m) : 0;
                    bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + pix * p.ldC + n_s;
                    if (!ID && p.s2d_cin) {                            // depth-to-space: column block q = (ph, pw) -> pixel (+ph rows, +pw columns)
                        const int q = n_s / p.s2d_cin, ci = n_s - q * p.s2d_cin;
                        o = reinterpret_cast<bf16_t*>(p.out) + (pix + (int64_t)(q >> 1) * p.OWf + (q & 1)) * p.ldC + ci;
                    }
                    ov[k] = o;
                }
                if (accum) {
#pragma unroll
                    for (int k = 0; k < GRP; k++) oldv[k] = *reinterpret_cast<const uint4*>(ov[k]);
                }
#pragma unroll
                for (int k = 0; k < GRP; k++) {
                    if (!lv[k]) continue;
                    const int r = (g0 + k) * RPI + r0;
                    uint4 v = *reinterpret_cast<const uint4*>(stage + r * EP_LD + ch * 8);
                    if (!ID && p.pool_idx) {
                        const int o_ = pl_ow + wm * WTM + r;
                        const int wr_ = small_div(o_, p.OW, pl_rOW);
                        const int ow_ = o_ - wr_ * p.OW, orow_ = pl_oh + wr_;
                        const int wi_ = small_div(orow_, p.OH, pl_rOH);
                        const int oh_ = orow_ - wi_ * p.OH;
                        const int64_t pp = ((int64_t)(pl_img + wi_) * (p.OH >> 1) + (oh_ >> 1)) * (p.OW >> 1) + (ow_ >> 1);
                        const unsigned long long packed = *reinterpret_cast<const unsigned long long*>(p.pool_idx + pp * p.pool_ldi + n);
                        const uint4 gz = *reinterpret_cast<const uint4*>(p.pool_dz + pp * p.pool_ld + n);
                        const unsigned want = (unsigned)((oh_ & 1) * 2 + (ow_ & 1));
                        const unsigned* a = reinterpret_cast<const unsigned*>(&v);
                        const unsigned* b = reinterpret_cast<const unsigned*>(&gz);
                        unsigned w[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const float g0_ = ((packed >> (16 * q)) & 0xff) == want ? __uint_as_float(b[q] << 16) : 0.f;
                            const float g1_ = ((packed >> (16 * q + 8)) & 0xff) == want ? __uint_as_float(b[q] & 0xffff0000u) : 0.f;
                            w[q] = pack_bf2(__uint_as_float(a[q] << 16) + g0_, __uint_as_float(a[q] & 0xffff0000u) + g1_);
                        }
                        v = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                    if (accum) {
                        const unsigned* a = reinterpret_cast<const unsigned*>(&v);
                        const unsigned* b = reinterpret_cast<const unsigned*>(&oldv[k]);
                        unsigned w[4];
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            w[q] = pack_bf2(__uint_as_float(a[q] << 16) + __uint_as_float(b[q] << 16),
                                            __uint_as_float(a[q] & 0xffff0000u) + __uint_as_float(b[q] & 0xffff0000u));
                        v = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                    *reinterpret_cast<uint4*>(ov[k]) = v;
                }
                __builtin_amdgcn_sched_barrier(0);                     // keep the next group's loads behind this group's stores (registers)
            }
        } else {
#pragma unroll
            for (int it = 0; it < WTM / RPI; it++) {
                const int r = it * RPI + r0;
                const int64_t m = m0 + wm * WTM + r;
                if (m >= M || n >= p.Nout) continue;
                const int64_t pix = out_pixel(wm * WTM + r, m);
                uint4 v = *reinterpret_cast<const uint4*>(stage + r * EP_LD + ch * 8);
                bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + pix * p.ldC + n;
                if (!ID && p.pool_idx) {
                    const int o_ = pl_ow + wm * WTM + r;
                    const int wr_ = small_div(o_, p.OW, pl_rOW);
                    const int ow_ = o_ - wr_ * p.OW, orow_ = pl_oh + wr_;
                    const int wi_ = small_div(orow_, p.OH, pl_rOH);
                    const int oh_ = orow_ - wi_ * p.OH;
                    const int64_t pp = ((int64_t)(pl_img + wi_) * (p.OH >> 1) + (oh_ >> 1)) * (p.OW >> 1) + (ow_ >> 1);
                    const unsigned long long packed = *reinterpret_cast<const unsigned long long*>(p.pool_idx + pp * p.pool_ldi + n);
                    const uint4 gz = *reinterpret_cast<const uint4*>(p.pool_dz + pp * p.pool_ld + n);
                    const unsigned want = (unsigned)((oh_ & 1) * 2 + (ow_ & 1));
                    const unsigned* a = reinterpret_cast<const unsigned*>(&v);
                    const unsigned* b = reinterpret_cast<const unsigned*>(&gz);
                    unsigned w[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float g0 = ((packed >> (16 * q)) & 0xff) == want ? __uint_as_float(b[q] << 16) : 0.f;
                        const float g1 = ((packed >> (16 * q + 8)) & 0xff) == want ? __uint_as_float(b[q] & 0xffff0000u) : 0.f;
                        w[q] = pack_bf2(__uint_as_float(a[q] << 16) + g0, __uint_as_float(a[q] & 0xffff0000u) + g1);
                    }
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
                if (!ID && p.s2d_cin) {                                // depth-to-space: column block q = (ph, pw) -> pixel (+ph rows, +pw columns)
                    const int q = n / p.s2d_cin, ci = n - q * p.s2d_cin;
                    o = reinterpret_cast<bf16_t*>(p.out) + (pix + (int64_t)(q >> 1) * p.OWf + (q & 1)) * p.ldC + ci;
                }
                if (p.epi == EPI_ACCUM) {
                    const uint4 old = *reinterpret_cast<const uint4*>(o);
                    const unsigned* a = reinterpret_cast<const unsigned*>(&v);
                    const unsigned* b = reinterpret_cast<const unsigned*>(&old);
                    unsigned w[4];
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        w[q] = pack_bf2(__uint_as_float(a[q] << 16) + __uint_as_float(b[q] << 16),
                                        __uint_as_float(a[q] & 0xffff0000u) + __uint_as_float(b[q] & 0xffff0000u));
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
                *reinterpret_cast<uint4*>(o) = v;
            }
        }
#ifdef GEMM_TIMING
        T3 = __builtin_readcyclecounter();
#endif
        if (p.epi == EPI_STATS) {
            // BatchNorm batch statistics of the values actually stored (bf16-rounded), read back from the staged block:
            // lane -> (4-channel quad, row group); rows past M were zero-filled A rows and contribute exact zeros
            constexpr int NQ = WTN / 4, RG = 64 / NQ;
            const int cq = lane % NQ, rg = lane / NQ;
            float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < WTM / RG; k++) {
                const uint2 w = *reinterpret_cast<const uint2*>(stage + (rg + RG * k) * EP_LD + cq * 4);
                const float f0 = __uint_as_float(w.x << 16), f1 = __uint_as_float(w.x & 0xffff0000u);
                const float f2 = __uint_as_float(w.y << 16), f3 = __uint_as_float(w.y & 0xffff0000u);
                ssum[0] += f0; ssq[0] += f0 * f0;
                ssum[1] += f1; ssq[1] += f1 * f1;
                ssum[2] += f2; ssq[2] += f2 * f2;
                ssum[3] += f3; ssq[3] += f3 * f3;
            }
            __syncthreads();                                   // staging blocks dead: LDS becomes the reduction buffer
            // every lane parks its 8 partial sums in LDS and one thread per column folds the RG x WM partials: the cross-lane
            // shuffle tree this replaces (16 dependent ds_bpermute round trips) cost ~2 k cycles of a short-K workgroup
            float* part = reinterpret_cast<float*>(smem);      // [wave][RG][2][WTN]
            float* mine = part + ((wave * RG + rg) * 2) * WTN + cq * 4;
            *reinterpret_cast<float4*>(mine) = make_float4(ssum[0], ssum[1], ssum[2], ssum[3]);
            *reinterpret_cast<float4*>(mine + WTN) = make_float4(ssq[0], ssq[1], ssq[2], ssq[3]);
            __syncthreads();
            if (tid < BN && n0 + tid < p.Nout) {
                const int wn_c = tid / WTN, cc = tid % WTN;
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int w = 0; w < WM; w++)
#pragma unroll
                    for (int r = 0; r < RG; r++) {
                        const float* src = part + (((w * WN + wn_c) * RG + r) * 2) * WTN + cc;
                        sm += src[0];
                        sq += src[WTN];
                    }
                float* st = p.stats + (int64_t)mb * 2 * p.Nout;
                st[n0 + tid] = sm;
                st[p.Nout + n0 + tid] = sq;
            }
        }
    }
#ifdef GEMM_TIMING
    if (p.bias && tid == 0 && p.epi != EPI_F32_BIAS) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias)) + (size_t)blockIdx.x * 4;
        dbg[0] = T1 - T0; dbg[1] = T2 - T1; dbg[2] = T3 - T2; dbg[3] = __builtin_readcyclecounter() - T3;
    }
#endif
}

// ------------------------------------------------------------------------------------------------ weight gradient

// BM = 128 output channels x BN = 128 (tap,cin) columns; K = pixels.  LDS tiles are pixel-major [BK][128+PAD].
// pixel-major LDS rows of 128 channels + 32 pad: row stride 320 B = 64 B (mod 256 B), so the 4 rows x 32 B a 16-lane group of
// ds_read_b64_tr_b16 touches (and the neighbouring group's +32 B) fall on 8 disjoint bank ranges -> conflict free
#define WG_LD 160
// exact n / d for 0 <= n < 2^31 as mulhi(n, m) >> s (host: wg_magic): output pixels per image and per row — the X gather decomposes
// its pixel index without the loop-carried (image, row, column) walk of the first version, whose `while` carries compiled to a chain of
// divergent branches (5 per K step; ~270 instructions per K step for 8 MFMAs).
struct WgMagic { unsigned m_img, s_img, m_row, s_row; };
// P1: single tap (0, 0), stride 1, input grid == output grid (every 1x1 layer: 40 of yolov7's 46 generic weight gradients): the X row of
// output pixel m IS input pixel m — both operands walk their tensors with plain pointer increments.
template <int BM, bool P1>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p, const WgMagic mg)
{
    constexpr int BN = 128;
    constexpr int WM = BM == 128 ? 2 : 1, WN = 4 / WM;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 2 * BK * WG_LD];
#define wA_(b) (smem + (b) * 2 * BK * WG_LD)
#define wB_(b) (smem + (b) * 2 * BK * WG_LD + BK * WG_LD)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    // XCD-aware order (1-D grid): all (output-channel, column) tiles of one pixel range get consecutive remapped ids, i.e. run on
    // ONE XCD at about the same time, so the dY / X rows they all read are fetched into that XCD's L2 once.  With the plain 3-D
    // grid the sibling tiles were dealt round-robin to the 8 XCDs and every L2 fetched the same rows again: 9x the HBM /
    // Infinity-Cache traffic for a 3x3 layer — the kernel ran at the speed of its loads (ablation in DESIGN.md).
    const int cchunks = p.Cin / BK;
    const int nchunks = p.ntaps * cchunks;
    // chunks per column tile: 4, or 3 = ONE KERNEL ROW per tile for a 3x3 layer with 32 input channels and <= 64 output channels (the 32 -> 64
    // stride-2 layer at 800 -> 400; wgrad_row_tiles on the host).  With 4 the nine taps split 4 / 4 / 1: the three tiles of a pixel range ran at
    // different speeds, nothing they read was shared in L2, and the launch moved 9.1 GB for 3.9 GB of operands (PMC).  Row tiles do equal work on
    // the same dY rows at the same time, and a tile's three taps are three ADJACENT input pixels of one input row.
    const int CPT = (BM == 64 && p.ntaps == 9 && cchunks == 1) ? 3 : 4;
    const int gx = (p.Cout + BM - 1) / BM, gy = (nchunks + CPT - 1) / CPT;
    const int t_id = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = t_id % gx, by = (t_id / gx) % gy, bz = t_id / (gx * gy);
    const int i0 = bx * BM;                            // output-channel block
    const int q0 = by * CPT;                           // first 32-wide column chunk (tap-major, then cin)
    const int64_t kbeg = (int64_t)bz * p.kchunk;
    const int64_t kend = min(M, kbeg + p.kchunk);
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg + BK - 1) / BK);

    // A pieces: BK pixels x BM channels -> BM/8 16B pieces per pixel
    constexpr int APP = BM / 8;                        // pieces per pixel row
    constexpr int PA = BK * APP / 256;                 // 2 (BM=128) or 1 (BM=64)
    // B pieces: 4 chunks x BK pixels x 4 slots = 512 -> 2 per thread
    // register-staged global -> LDS pipeline, THREE K steps deep: a step's loads are issued three steps before they are stored
    // to LDS (one step of lead left every wave waiting a full HBM latency per 8 MFMAs: 16 % of the MFMA peak)
    struct RegSet { uint4 a[PA]; uint4 b[2]; unsigned ok; };   // ok: bit u = A piece u valid, bit 8+u = B piece u valid
    RegSet R0, R1, R2;
    int b_tap[2], b_c0[2];
    bool b_chunk_ok[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int id = tid + 256 * u;
        const int q = q0 + (id >> 7);
        b_chunk_ok[u] = (id >> 7) < CPT && q < nchunks;
        const int qq = b_chunk_ok[u] ? q : 0;
        b_tap[u] = qq / cchunks;
        b_c0[u] = (qq - b_tap[u] * cchunks) * BK + (id & 3) * 8;
    }
    int b_dh[2] = {0, 0}, b_dw[2] = {0, 0};
    if constexpr (!P1) {
#pragma unroll
        for (int u = 0; u < 2; u++) { b_dh[u] = p.dh[b_tap[u]]; b_dw[u] = p.dw[b_tap[u]]; }
    }
    // Loop-carried gather state (no division and no 64-bit multiply chain per K step — the first version of this loop
    // recomputed (img, oh, ow) from the pixel index with two integer divisions per piece per step):
    //   A pieces walk dY rows linearly: pointer += BK*ldY per step;
    //   B pieces share ONE pixel lane per thread ((tid & 127) >> 2): (img, oh, ow) advances by BK pixels per step.
    const bf16_t* a_ptr[PA];
    int a_px[PA];
    bool a_chan_ok[PA];
#pragma unroll
    for (int u = 0; u < PA; u++) {
        const int id = tid + 256 * u;
        const int px = id / APP, pc = id % APP;
        a_px[u] = px;
        a_chan_ok[u] = i0 + pc * 8 < p.CoutPad;
        a_ptr[u] = p.dY + (kbeg + px) * (int64_t)p.ldY + i0 + pc * 8;
    }
    const int b_px = (tid & 127) >> 2;
    const unsigned HWo = (unsigned)(p.OH * p.OW);
    const bf16_t* b_ptr[2] = {nullptr, nullptr};
    if constexpr (P1) {
#pragma unroll
        for (int u = 0; u < 2; u++) b_ptr[u] = p.X + (kbeg + b_px) * (int64_t)p.ldX + b_c0[u];
    }
    int64_t m_step = kbeg;                                   // first pixel of the step being loaded (M < 2^31: host check)
    auto gload = [&](RegSet& rs) {
        // loads are UNCONDITIONAL (invalid pieces read a safe address and are zeroed when stored): a branch around a load makes
        // hipcc fall back to s_waitcnt vmcnt(0) at the join, which would serialise the three-deep pipeline again
        unsigned ok = 0;
#pragma unroll
        for (int u = 0; u < PA; u++) {
            const bool v = m_step + a_px[u] < kend && a_chan_ok[u];
            rs.a[u] = *reinterpret_cast<const uint4*>(v ? a_ptr[u] : p.dY);
            ok |= v ? (1u << u) : 0u;
            a_ptr[u] += (int64_t)BK * p.ldY;
        }
        const bool pix_ok = m_step + b_px < kend;
        if constexpr (P1) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const bool v = pix_ok && b_chunk_ok[u];
                rs.b[u] = *reinterpret_cast<const uint4*>(v ? b_ptr[u] : p.X);
                ok |= v ? (0x100u << u) : 0u;
                b_ptr[u] += (int64_t)BK * p.ldX;
            }
        } else {
            // (image, row, column) of output pixel m_step + b_px: two exact multiply-high divisions, no loop-carried state
            const unsigned m = (unsigned)(m_step + b_px);
            const unsigned b_img = mg.m_img ? __umulhi(m, mg.m_img) >> mg.s_img : m;          // (m_* == 0: division by one)
            const unsigned rem = m - b_img * HWo;
            const unsigned b_oh = mg.m_row ? __umulhi(rem, mg.m_row) >> mg.s_row : rem;
            const unsigned b_ow = rem - b_oh * (unsigned)p.OW;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int ih = (int)b_oh * p.sh + b_dh[u], iw = (int)b_ow * p.sw + b_dw[u];
                const bool v = pix_ok && b_chunk_ok[u] && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
                const bf16_t* src = v ? p.X + (((int64_t)b_img * p.IH + ih) * p.IW + iw) * p.ldX + b_c0[u] : p.X;
                rs.b[u] = *reinterpret_cast<const uint4*>(src);
                ok |= v ? (0x100u << u) : 0u;
            }
        }
        rs.ok = ok;
        m_step += BK;
    };
    auto sstore = [&](int buf, const RegSet& rs) {
#pragma unroll
        for (int u = 0; u < PA; u++) {
            const int id = tid + 256 * u;
            const int px = id / APP, pc = id % APP;
            *reinterpret_cast<uint4*>(wA_(buf) + px * WG_LD + pc * 8) = (rs.ok >> u) & 1u ? rs.a[u] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int id = tid + 256 * u;
            const int ch = id >> 7, px = (id & 127) >> 2, sl = id & 3;
            *reinterpret_cast<uint4*>(wB_(buf) + px * WG_LD + ch * 32 + sl * 8) = (rs.ok >> (8 + u)) & 1u ? rs.b[u] : make_uint4(0, 0, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    auto compute = [&](int buf) {
        // Transposed fragment reads (gfx950 ds_read_b64_tr_b16): the tiles are pixel-major [pixel][channel] as they come
        // from HBM, the MFMA wants 8 consecutive pixels (K) per lane for ONE channel.  Within each 16-lane group the
        // instruction transposes a 4(pixel) x 16(channel) block: lane s supplies the address of pixel (s>>2), channels
        // 4*(s&3)..+3 and receives channel s for the 4 pixels.  Two reads (pixels +0..3, +4..7) build one operand.
        // All reads of the K step (both 16-pixel halves) are issued before the first MFMA, through the compiler builtin so
        // hipcc counts them (lgkmcnt) and overlaps the tail of the reads with the first MFMAs.
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const int s16 = lane & 15, grp = lane >> 4;
        bf16x8 af[2][TM], bfr[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int prow = ks * 16 + (grp >> 1) * 8 + (s16 >> 2);
            const int pcol = 16 * (grp & 1) + 4 * (s16 & 3);
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const bf16_t* a = wA_(buf) + prow * WG_LD + wm * (BM / WM) + i * 32 + pcol;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * WG_LD));
                af[ks][i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const bf16_t* b = wB_(buf) + prow * WG_LD + wn * (BN / WN) + j * 32 + pcol;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)b);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b + 4 * WG_LD));
                bfr[ks][j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    gload(R0);
    if (nk > 1) gload(R1);
    if (nk > 2) gload(R2);
    sstore(0, R0);
    __syncthreads();
    // step k: `cur` held step k (already in LDS) and is free -> receives step k+3; `nxt` holds step k+1 -> goes to the other buffer
    auto step = [&](auto steady, int k, RegSet& cur, const RegSet& nxt) {
        const int buf = k & 1;
        // the steady-state body is branch free: a conditional load or store is a control-flow join, and at joins hipcc's
        // waitcnt pass gives up the counted vmcnt(N) and drains to 0 (seen in the ISA) — which would undo the pipeline
#ifndef WG_NO_LOAD
        if (decltype(steady)::value || k + 3 < nk) gload(cur);
#endif
#ifndef WG_NO_COMPUTE
        compute(buf);
#endif
#ifndef WG_NO_STORE
        if (decltype(steady)::value || k + 1 < nk) sstore(buf ^ 1, nxt);
#endif
        __syncthreads();
    };
    int k = 0;
    for (; k + 5 < nk; k += 3) {
        step(std::true_type{}, k, R0, R1);
        step(std::true_type{}, k + 1, R1, R2);
        step(std::true_type{}, k + 2, R2, R0);
    }
    for (; k < nk; k += 3) {
        step(std::false_type{}, k, R0, R1);
        if (k + 1 < nk) step(std::false_type{}, k + 1, R1, R2);
        if (k + 2 < nk) step(std::false_type{}, k + 2, R2, R0);
    }

    // split-K partial tile -> workspace [z][Cout][ntaps*Cin] (GEMM layout; 128-byte row segments per store instruction).
    // No float atomics: the reduction over z is a separate deterministic pass.
    const int NK = p.ntaps * p.Cin;
    float* part = p.partial + (int64_t)bz * p.Cout * NK;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int col = wn * (BN / WN) + j * 32 + (lane & 31);
        const int q = q0 + (col >> 5);
        if ((col >> 5) >= CPT || q >= nchunks) continue;
        const int kc = q * BK + (col & 31);
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int co = i0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (co < p.Cout) part[(int64_t)co * NK + kc] = acc[i][j][e];
            }
        }
    }
}

// ---- 1x1 (pointwise) weight gradient on an LDS-DMA ring -------------------------------------------------------------------
// dW[co][ci] = sum_p dY[p][co] * X[p][ci] for single-tap stride-1 layers (40 of yolov7's 46 generic weight gradients).  The
// register-staged kernel above spends ~200 instructions per 32-pixel K step (global loads into VGPRs, validity selects, four
// ds_write_b128, the gather bookkeeping) around 8 MFMAs; with both operands plain [pixel][channel] rows the stage can be filled by
// global_load_lds straight into the layout the transposed fragment reads want — the design of conv3x3_wgrad_kernel without its halo
// ring: a stage = 4 + 4 channel quarters of [32 pixels][64 B] (lane-linear 1-KiB DMA pieces = 16 pixel rows x 64 B, conflict-free for
// ds_read_b64_tr_b16), 3 stages, counted vmcnt, one barrier per step.  Workgroup tile 128 co x 128 ci, wave (wm, wn) owns 64 x 64:
// per step and wave 4 DMA pieces + 16 transposed reads + 8 MFMAs.  Split-K slabs + the deterministic reduce as before.
// PX = pixels per K step: 32 (three 16-KiB stages, two in flight, three workgroups per CU) or 64 (two 32-KiB stages, one in flight, two
// workgroups per CU, 16 MFMAs per wave between barriers instead of 8 — the lesson of conv3x3_wgrad64_kernel).
template <int PX>
__global__ __launch_bounds__(256, PX == 64 ? 2 : 3) void wgrad1x1_dma_kernel(const WgradParams p)
{
    constexpr int NS = PX == 64 ? 2 : 3;                           // ring stages
    constexpr int OPB = PX * 256;                                  // bytes of one operand in a stage: [PX pixels][256 B]
    constexpr int STB = 2 * OPB;                                   // stage bytes
    constexpr int NP = PX / 8;                                     // DMA pieces per wave and step (PX / 4 per operand, 2 operands, 4 waves)
    __shared__ __attribute__((aligned(1024))) unsigned char w1_lds[NS * STB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    const int gx = (p.Cout + 127) / 128, gy = (p.Cin + 127) / 128;
    const int t_id = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = t_id % gx, by = (t_id / gx) % gy, bz = t_id / (gx * gy);
    const int i0 = bx * 128, j0 = by * 128;
    const int64_t kbeg = (int64_t)bz * p.kchunk;
    const int64_t kend = min(M, kbeg + p.kchunk);
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg + PX - 1) / PX);

    // DMA pieces: id = NP * wave + u; the first PX / 4 ids = dY, the rest = X; piece k of an operand = pixel rows 4k .. 4k + 3 of the stage,
    // WHOLE 256-byte rows (128 channels): lane -> (row = lane >> 4, 64-byte quarter position = (lane >> 2) & 3, 16-byte slot = lane & 3).
    // (A first version cut the stage into per-quarter [32 px][64 B] blocks like conv3x3_wgrad_kernel: its pieces fetched 16 rows x 64
    // bytes — four half-line requests per 256-byte row — and lost 8 % on the HBM-bound high-resolution layers against the
    // register-staged kernel, whose loads walk whole rows.)  The LDS image is lane-linear, i.e. plain [pixel][256 B] rows; four
    // consecutive rows of one quarter would sit on the same banks, so the quarter POSITION is swizzled with the row on the source
    // side: position qp of row r holds channel quarter qp ^ (r & 3); the reader applies the same XOR (its rows are 4-aligned groups).
    const int prow = lane >> 4, qpos = (lane >> 2) & 3, slot = lane & 3;
    const int qsrc = qpos ^ (prow & 3);                            // (piece rows start at multiples of 4: r & 3 == prow & 3)
    const bf16_t* src[NP];
    bool chan_ok[NP];
    int pix0[NP];                                                  // pixel (relative to kbeg) of this lane's row in stage 0
#pragma unroll
    for (int u = 0; u < NP; u++) {
        const int id = NP * wave + u, k = id % (PX / 4);
        pix0[u] = 4 * k + prow;
        if (id < PX / 4) {
            chan_ok[u] = i0 + 32 * qsrc + slot * 8 < p.CoutPad;
            src[u] = p.dY + (kbeg + pix0[u]) * (int64_t)p.ldY + i0 + 32 * qsrc + slot * 8;
        } else {
            chan_ok[u] = j0 + 32 * qsrc < p.Cin;
            src[u] = p.X + (kbeg + pix0[u]) * (int64_t)p.ldX + j0 + 32 * qsrc + slot * 8;
        }
    }
    const int64_t step_a = PX * (int64_t)p.ldY, step_b = PX * (int64_t)p.ldX;
    const int npix = (int)(kend - kbeg);
    int issued = 0;                                                // stages issued so far
    auto issue_stage = [&]() {
        unsigned char* st = w1_lds + (issued % NS) * STB;
#pragma unroll
        for (int u = 0; u < NP; u++) {
            const int id = NP * wave + u;
            const bool ok = chan_ok[u] && pix0[u] + PX * issued < npix;
            const bf16_t* s_ = ok ? src[u] : p.zeros;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)s_, (lds_void_t*)(st + id * 1024), 16, 0, 0);
            src[u] += id < PX / 4 ? step_a : step_b;
        }
        issued++;
    };
    issue_stage();
    if (NS == 3 && nk > 1) issue_stage();

    // transposed fragment reads (conv_internal.h): lane -> pixel row (grp >> 1) * 8 + (s16 >> 2) [+ 16 ks, + 4 for the second read],
    // channels 16 (grp & 1) + 4 (s16 & 3) .. + 3 of the quarter.  Row stride 256 B; quarter q of row r sits at position q ^ (r & 3)
    const int s16 = lane & 15, grp = lane >> 4;
    const unsigned fr_row = (unsigned)((grp >> 1) * 8 + (s16 >> 2));
    const unsigned fr_col = (unsigned)((16 * (grp & 1) + 4 * (s16 & 3)) * 2);
    unsigned fa[2], fbq[2];                                        // byte offsets inside an operand's stage half for ks = 0, first read
#pragma unroll
    for (int i = 0; i < 2; i++) fa[i] = fr_row * 256u + (unsigned)(((2 * wm + i) ^ (int)(fr_row & 3u)) * 64) + fr_col;
#pragma unroll
    for (int j = 0; j < 2; j++) fbq[j] = fr_row * 256u + (unsigned)(((2 * wn + j) ^ (int)(fr_row & 3u)) * 64) + fr_col;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    for (int s = 0; s < nk; s++) {
        if (NS == 2 || s + 1 >= nk) gemm_wait_vm<0>();             // NS == 3: the stage issued during step s - 1 (operands of step s + 1) may stay in flight
        else gemm_wait_vm<NP>();
        __builtin_amdgcn_s_barrier();                               // stage s visible to every wave; stage s - 1 fully consumed
        if (s + NS - 1 < nk) issue_stage();
#pragma unroll
        for (int hh = 0; hh < PX / 32; hh++) {
            const unsigned base = lds_addr(w1_lds + (s % NS) * STB) + (unsigned)(hh * 8192);
            // fragments through the asm reads (the builtin would make hipcc drain vmcnt(0), i.e. the DMA just issued: conv_internal.h); second
            // read of a fragment = 4 rows further (1024 B: same row & 3, same quarter position), ks = 1: 16 rows further (4096 B)
            bf16x8 af[2][2], bq[2][2];
            // (lgkmcnt is a 4-bit counter: never more than 12 of this wave's reads in flight)
#pragma unroll
            for (int i = 0; i < 2; i++) af[0][i] = lds_tr16x2(base + fa[i], 1024u);
#pragma unroll
            for (int j = 0; j < 2; j++) bq[0][j] = lds_tr16x2(base + (unsigned)OPB + fbq[j], 1024u);
#pragma unroll
            for (int i = 0; i < 2; i++) af[1][i] = lds_tr16x2(base + fa[i] + 4096u, 1024u);
            lds_wait2<4>(af[0][0], af[0][1]);                       // later reads: the four of af[1][*]
            lds_wait2<4>(bq[0][0], bq[0][1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; j++) bq[1][j] = lds_tr16x2(base + (unsigned)OPB + fbq[j] + 4096u, 1024u);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bq[0][j], acc[i][j], 0, 0, 0);
            lds_wait2<0>(af[1][0], af[1][1]);
            lds_wait2<0>(bq[1][0], bq[1][1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bq[1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // split-K partial tile -> workspace [z][Cout][Cin] fp32: acc[i][j][e] = (co = i0 + 64 wm + 32 i + (e & 3) + 8 (e >> 2) + 4 (lane >> 5), ci = j0 + 64 wn + 32 j + (lane & 31))
    float* part = p.partial + (int64_t)bz * p.Cout * p.Cin;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int ci = j0 + 64 * wn + 32 * j + (lane & 31);
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int co = i0 + 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (co < p.Cout) part[(int64_t)co * p.Cin + ci] = acc[i][j][e];
            }
    }
}

// ---- tapped / strided weight gradient on the same LDS-DMA ring (r04) -------------------------------------------------------
// dW[co][tap][ci] = sum_p dY[p][co] * X[in(p, tap)][ci] for the layers the halo-ring kernel (conv3x3.hip) does not take: the stride-2 3x3
// layers (6 launches of yolov7, 2.1 ms at ~490 TF/s on the register-staged kernel above).  Same stage layout, fragment reads and K loop as
// wgrad1x1_dma_kernel<32>; what changes is the SOURCE of an X piece: LDS-DMA takes a per-lane global address, so the gather (output pixel ->
// (image, row, column) by multiply-high, input pixel of the lane's tap, bounds -> the zero page) happens on the request side and the
// padding never exists in LDS as a select.  A 128-column tile = 4 chunks of 32 input channels; chunk q = (tap, channel block) = (q / (Cin/32),
// q % (Cin/32)) as in the generic kernel, so its split-K slabs and the reduce are unchanged.  A lane's quarter (and with it its tap) is
// the same for every piece it requests.  Every wave requests 2 dY + 2 X pieces per stage (the pointwise kernel gives waves 0-1 dY and 2-3 X:
// here that would put the whole gather on two waves); the X addresses of the NEXT request are computed behind the step's MFMAs.
__global__ __launch_bounds__(256, 3) void wgrad_taps_dma_kernel(const WgradParams p, const WgMagic mg)
{
    constexpr int PX = 32, NS = 3, OPB = PX * 256, STB = 2 * OPB, NP = 4;
    __shared__ __attribute__((aligned(1024))) unsigned char wt_lds[NS * STB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    const int cchunks = p.Cin / BK, nchunks = p.ntaps * cchunks;
    const int gx = (p.Cout + 127) / 128, gy = (nchunks + 3) / 4;
    const int t_id = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = t_id % gx, by = (t_id / gx) % gy, bz = t_id / (gx * gy);
    const int i0 = bx * 128, q0 = by * 4;
    const int64_t kbeg = (int64_t)bz * p.kchunk;
    const int64_t kend = min(M, kbeg + p.kchunk);
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg + PX - 1) / PX);
    const int npix = (int)(kend - kbeg);

    // piece k (0..7) of an operand = stage pixel rows 4k .. 4k + 3, whole 256-byte rows: lane -> (row = lane >> 4, quarter position =
    // (lane >> 2) & 3, 16-byte slot = lane & 3); position qp of row r holds quarter qp ^ (r & 3) (wgrad1x1_dma_kernel).  Wave w owns k = 2w, 2w + 1.
    const int prow = lane >> 4, qpos = (lane >> 2) & 3, slot = lane & 3;
    const int qsrc = qpos ^ (prow & 3);
    const bool a_ok = i0 + 32 * qsrc + slot * 8 < p.CoutPad;
    const bf16_t* a_src[2];
    int pix[2];                                                      // pixel (relative to kbeg) of this lane's row in the NEXT stage to request
#pragma unroll
    for (int u = 0; u < 2; u++) {
        pix[u] = 4 * (2 * wave + u) + prow;
        a_src[u] = p.dY + (kbeg + pix[u]) * (int64_t)p.ldY + i0 + 32 * qsrc + slot * 8;
    }
    const int64_t step_a = PX * (int64_t)p.ldY;
    const int q = q0 + qsrc;
    const bool b_ok = q < nchunks;
    const int tap = b_ok ? q / cchunks : 0;
    const int b_c0 = (b_ok ? q - tap * cchunks : 0) * BK + slot * 8;
    const int b_dh = p.dh[tap], b_dw = p.dw[tap];
    const unsigned HWo = (unsigned)(p.OH * p.OW);
    const bf16_t* b_src[2];                                          // X rows of the next stage to request (the zero page for padding / tails)
    auto gather = [&]() {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const unsigned m = (unsigned)((int)kbeg + pix[u]);
            const unsigned img = mg.m_img ? __umulhi(m, mg.m_img) >> mg.s_img : m;          // (m_* == 0: division by one)
            const unsigned rem = m - img * HWo;
            const unsigned oh = mg.m_row ? __umulhi(rem, mg.m_row) >> mg.s_row : rem;
            const unsigned ow = rem - oh * (unsigned)p.OW;
            const int ih = (int)oh * p.sh + b_dh, iw = (int)ow * p.sw + b_dw;
            const bool v = b_ok && pix[u] < npix && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
            b_src[u] = v ? p.X + (((int64_t)img * p.IH + ih) * p.IW + iw) * p.ldX + b_c0 : p.zeros;
        }
    };
    int issued = 0;
    auto issue_stage = [&]() {                                       // needs gather() for this stage; leaves pix[] at the following one
        unsigned char* st = wt_lds + (issued % NS) * STB;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const bf16_t* s_ = (a_ok && pix[u] < npix) ? a_src[u] : p.zeros;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)s_, (lds_void_t*)(st + (2 * wave + u) * 1024), 16, 0, 0);
            a_src[u] += step_a;
        }
#pragma unroll
        for (int u = 0; u < 2; u++)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)b_src[u], (lds_void_t*)(st + OPB + (2 * wave + u) * 1024), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < 2; u++) pix[u] += PX;
        issued++;
    };
    gather();
    issue_stage();
    if (nk > 1) { gather(); issue_stage(); }
    if (nk > 2) gather();

    const int s16 = lane & 15, grp = lane >> 4;
    const unsigned fr_row = (unsigned)((grp >> 1) * 8 + (s16 >> 2));
    const unsigned fr_col = (unsigned)((16 * (grp & 1) + 4 * (s16 & 3)) * 2);
    unsigned fa[2], fbq[2];
#pragma unroll
    for (int i = 0; i < 2; i++) fa[i] = fr_row * 256u + (unsigned)(((2 * wm + i) ^ (int)(fr_row & 3u)) * 64) + fr_col;
#pragma unroll
    for (int j = 0; j < 2; j++) fbq[j] = fr_row * 256u + (unsigned)(((2 * wn + j) ^ (int)(fr_row & 3u)) * 64) + fr_col;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    for (int s = 0; s < nk; s++) {
        if (s + 1 >= nk) gemm_wait_vm<0>();                          // the stage requested during step s - 1 (operands of step s + 1) may stay in flight
        else gemm_wait_vm<NP>();
        __builtin_amdgcn_s_barrier();
        if (s + NS - 1 < nk) issue_stage();
        const unsigned base = lds_addr(wt_lds + (s % NS) * STB);
        bf16x8 af[2][2], bq[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++) af[0][i] = lds_tr16x2(base + fa[i], 1024u);
#pragma unroll
        for (int j = 0; j < 2; j++) bq[0][j] = lds_tr16x2(base + (unsigned)OPB + fbq[j], 1024u);
#pragma unroll
        for (int i = 0; i < 2; i++) af[1][i] = lds_tr16x2(base + fa[i] + 4096u, 1024u);
        lds_wait2<4>(af[0][0], af[0][1]);
        lds_wait2<4>(bq[0][0], bq[0][1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; j++) bq[1][j] = lds_tr16x2(base + (unsigned)OPB + fbq[j] + 4096u, 1024u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bq[0][j], acc[i][j], 0, 0, 0);
        lds_wait2<0>(af[1][0], af[1][1]);
        lds_wait2<0>(bq[1][0], bq[1][1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bq[1][j], acc[i][j], 0, 0, 0);
        if (s + NS < nk) gather();                                   // addresses of the stage step s + 1 requests: VALU work in the MFMA shadow
        __builtin_amdgcn_sched_barrier(0);
    }
    const int NK = p.ntaps * p.Cin;
    float* part = p.partial + (int64_t)bz * p.Cout * NK;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int kc = by * 128 + 64 * wn + 32 * j + (lane & 31);
        if (kc >= NK) continue;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int co = i0 + 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (co < p.Cout) part[(int64_t)co * NK + kc] = acc[i][j][e];
            }
    }
}

// dW[co][cin][tap] += sum_z partial[z][co][tap*Cin + cin]   (torch weight layout; fixed summation order => deterministic).
// One workgroup per (co, CH-channel chunk): P = ntaps*CH/4 float4 positions x ZL split lanes stream the split-K slabs with
// 16-byte loads (the first version used 4-byte loads, 32 channel lanes x 32 split lanes: 1.9 TB/s over 5.5 GB of slabs per step,
// request-rate bound), an LDS pass folds the split lanes, and the [tap][cin] -> [cin][tap] transpose happens in LDS so the
// read-modify-write of the torch-layout gradient is one contiguous CH*ntaps-float run.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, int splitk, int Cout, int Cin, int ntaps,
                                                            int CH, int ZL, float* __restrict__ dW, float* __restrict__ dW2, int Cout1)
{
    extern __shared__ float red[];                   // [ZL][ntaps][CH + 1]
    const int c4n = CH >> 2, P = ntaps * c4n;
    const int pos = threadIdx.x % P, zl = threadIdx.x / P;
    const int t = pos / c4n, c4 = pos - t * c4n;
    const int co = blockIdx.x, cin0 = blockIdx.y * CH;
    const int NK = ntaps * Cin;
    const int64_t slab = (int64_t)Cout * NK;
    const int ldr = CH + 1;
    if (zl < ZL) {
        const float* base = partial + (int64_t)co * NK + (int64_t)t * Cin + cin0 + c4 * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int z = zl;
        for (; z + 3 * ZL < splitk; z += 4 * ZL) {
            const float4 v0 = *reinterpret_cast<const float4*>(base + (int64_t)z * slab);
            const float4 v1 = *reinterpret_cast<const float4*>(base + (int64_t)(z + ZL) * slab);
            const float4 v2 = *reinterpret_cast<const float4*>(base + (int64_t)(z + 2 * ZL) * slab);
            const float4 v3 = *reinterpret_cast<const float4*>(base + (int64_t)(z + 3 * ZL) * slab);
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; z < splitk; z += ZL) {
            const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)z * slab);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float* r = red + ((int64_t)zl * ntaps + t) * ldr + c4 * 4;
        r[0] = s.x; r[1] = s.y; r[2] = s.z; r[3] = s.w;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < CH * ntaps; j += 1024) {           // output element within the contiguous run: j = cl*ntaps + t
        const int cl = j / ntaps, tt = j - cl * ntaps;
        float sum = 0.f;
        for (int z = 0; z < ZL; z++) sum += red[((int64_t)z * ntaps + tt) * ldr + cl];
        if (dW2 && co >= Cout1) dW2[((int64_t)(co - Cout1) * Cin + cin0) * ntaps + j] += sum;      // second parameter tensor of a shared launch
        else dW[((int64_t)co * Cin + cin0) * ntaps + j] += sum;
    }
}

// chunk width / split lanes / LDS of wgrad_reduce_kernel for one layer
static void launch_wgrad_reduce(const WgradParams& p, int splitk, hipStream_t stream)
{
    int CH = 32;
    if (p.ntaps == 1) { while (CH < 256 && p.Cin % (CH * 2) == 0) CH *= 2; }
    else if (p.ntaps <= 4) { while (CH < 64 && p.Cin % (CH * 2) == 0) CH *= 2; }
    const int P = p.ntaps * (CH / 4);
    int ZL = 1024 / P;
    if (ZL > splitk) ZL = splitk;
    if (ZL > 32) ZL = 32;
    const size_t lds = (size_t)ZL * p.ntaps * (CH + 1) * sizeof(float);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(p.Cout, p.Cin / CH), dim3(1024), lds, stream, p.partial, splitk, p.Cout, p.Cin, p.ntaps,
                       CH, ZL, p.dW, p.dW2, p.Cout1);
}

// ------------------------------------------------------------------------------------------------ C ABI
static bool gemm_ident(const ConvGemmParams& p)
{
    return p.nclasses == 1 && p.oh_mul == 1 && p.ow_mul == 1 && p.OHf == p.OH && p.OWf == p.OW && p.cls[0].oh_add == 0 && p.cls[0].ow_add == 0 &&
           !p.pool_idx && !p.s2d_cin;
}
// single tap (0, 0) on the identity grid, input grid == output grid: the 1x1 instantiations (no tile decomposition, no tap table)
static bool gemm_is_t1(const ConvGemmParams& p)
{
    static const bool t1_on = !(getenv("RYOLO_GEMM_T1") && atoi(getenv("RYOLO_GEMM_T1")) == 0);      // A/B knob
    return (p.pipe & 0xff) == 1 && gemm_ident(p) && t1_on && p.cls[0].ntaps == 1 && p.cls[0].dh[0] == 0 && p.cls[0].dw[0] == 0 && p.cls[0].widx[0] == 0 &&
           p.sh == 1 && p.sw == 1 && p.IH == p.OH && p.IW == p.OW;
}

template <int BM, int BN, int WM, int WN, int PIPE, int KB = 32>
static int launch_gemm(const ConvGemmParams& p, hipStream_t stream)
{
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    const int64_t gm = ry_cdiv(M, BM), gn = ry_cdiv(p.Nout, BN);
    if (gm * gn > 0x7fffffff) return RY_ERR_UNSUPPORTED;
    dim3 grid((unsigned)(gm * gn), 1, p.nclasses);
    ConvGemmParams q = p;
    // several tap classes over one input: class-chunked 1-D order (conv_gemm_kernel); RYOLO_GEMM_CLS_CHUNK = tiles per chunk (0: classes on blockIdx.z)
    static const int cls_chunk = getenv("RYOLO_GEMM_CLS_CHUNK") ? atoi(getenv("RYOLO_GEMM_CLS_CHUNK")) : 1024;     // same-box img/s: 0 -> 934.0, 256 -> 936.3, 512 -> 936.0, 1024 -> 937.9 (three alternating runs each)
    if (p.nclasses > 1 && cls_chunk >= 16 && gm * gn * p.nclasses <= 0x7fffffff) {
        q.pipe = (p.pipe & 0xffff) | ((cls_chunk / 16 > 0xfff ? 0xfff : cls_chunk / 16) << 16);
        grid = dim3((unsigned)(gm * gn * p.nclasses), 1, 1);
    }
    const bool ident = gemm_ident(p);
    const bool t1 = PIPE == 1 && gemm_is_t1(p);
    if constexpr (PIPE == 1) {
        if (t1) {
            if (p.epi == EPI_ACCUM) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, PIPE, KB, 1, true, true>), grid, dim3(256), 0, stream, q);
            else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, PIPE, KB, 0, true, true>), grid, dim3(256), 0, stream, q);
            return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
        }
    }
    if (p.epi == EPI_ACCUM) {
        if (ident) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, PIPE, KB, 1, true>), grid, dim3(256), 0, stream, q);
        else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, PIPE, KB, 1, false>), grid, dim3(256), 0, stream, q);
    } else {
        if (ident) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, PIPE, KB, 0, true>), grid, dim3(256), 0, stream, q);
        else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, PIPE, KB, 0, false>), grid, dim3(256), 0, stream, q);
    }
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}

// deep-ring instantiations of the 128 x 128 tile for grids that leave a workgroup alone (or in a pair) on its CU
template <int NST>
static int launch_gemm_deep(const ConvGemmParams& p, hipStream_t stream)
{
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    const dim3 grid((unsigned)(ry_cdiv(M, 128) * ry_cdiv(p.Nout, 128)), 1, 1);
    const bool t1 = gemm_is_t1(p), acc = p.epi == EPI_ACCUM;
    if (t1 && acc) hipLaunchKernelGGL((conv_gemm_kernel<128, 128, 2, 2, 1, 32, 1, true, true, NST>), grid, dim3(256), 0, stream, p);
    else if (t1) hipLaunchKernelGGL((conv_gemm_kernel<128, 128, 2, 2, 1, 32, 0, true, true, NST>), grid, dim3(256), 0, stream, p);
    else if (acc) hipLaunchKernelGGL((conv_gemm_kernel<128, 128, 2, 2, 1, 32, 1, true, false, NST>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_gemm_kernel<128, 128, 2, 2, 1, 32, 0, true, false, NST>), grid, dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}
static int gemm_deep_stages(const ConvGemmParams& p)
{
    static const int mode = getenv("RYOLO_GEMM_DEEP") ? atoi(getenv("RYOLO_GEMM_DEEP")) : 1;     // 0 off; 1 by grid size; 4 / 6 force that depth on every eligible launch (tests)
    if (!mode || (p.pipe & 0xff) != 1 || !gemm_ident(p) || p.Nout <= 64 || (p.pipe & 0x800)) return 0;
    if (mode == 4 || mode == 6) return mode;
    const int64_t tiles = ry_cdiv((int64_t)p.NB * p.OH * p.OW, 128) * ry_cdiv(p.Nout, 128);
    const int nk = p.cls[0].ntaps * (p.Cin / BK);
    if (nk < 12) return 0;
    return tiles <= 256 ? 6 : (tiles <= 512 ? 4 : 0);
}

// rows of one generic-kernel tile for these parameters: the SAME decision ryolo_conv_gemm's dispatch makes (statistics rows = M tiles)
static bool gemm_k64(const ConvGemmParams& p) { return (p.Cin % 64 == 0) && p.cls[0].ntaps > 1 && p.Cin <= 256 && !(p.pipe & 0x100); }
static bool gemm_wide_n64(const ConvGemmParams& p)
{
    static const int n64_wide = getenv("RYOLO_GEMM_N64") ? atoi(getenv("RYOLO_GEMM_N64")) : 1;   // 0 off, 1 large grids, 2 every grid (parity tests on small grids)
    return (p.pipe & 0xff) && p.Nout > 32 && p.Nout <= 64 && n64_wide && !gemm_k64(p) && ((int64_t)p.NB * p.OH * p.OW >= 256ll * 1536 || n64_wide == 2);
}
static int gemm_tile_rows(const ConvGemmParams& p) { return (p.Nout <= 32 || gemm_wide_n64(p)) ? 256 : 128; }

extern "C" int ryolo_conv_gemm_stats_rows(int64_t M, int Nout, int pipe, int* rows)
{
    // upper bound of the [2][Nout] partial-statistics rows the EPI_STATS epilogue writes (exact: ryolo_conv_gemm_plan, which knows the tile)
    if (!rows) return RY_ERR_ARG;
    (void)pipe;
    *rows = (int)ry_cdiv(M, Nout <= 32 ? 256 : 128);               // (n tiles share the row; every generic tile is 128 pixels but the 256-pixel ones)
    return RY_OK;
}

// Stride-2 3x3 (pad 1) data gradient as ONE stride-1 GEMM over the dY grid (space-to-depth on the output): dx[2a + ph][2b + pw][ci] =
// sum over (da, db) in {0,1}^2 and co of dY[a + da][b + db][co] * W[co][ci][ph + 1 - 2 da][pw + 1 - 2 db] (taps outside the 3x3 kernel are
// structural zeros: 9 of 16 live).  out [4 * Cin][4 taps][CoutP] bf16, row n = (2 ph + pw) * Cin + ci, tap = 2 da + db.  For narrow layers
// (Cin <= 32) the four 32-column parity classes of the exact formulation ran the 256x32 tile at 6 % of the MFMA peak and wrote each
// 128-byte line in two 64-byte halves at different times; this form has a full 128-wide N tile and stores whole pixel pairs.
__global__ void pack_s2d_kernel(const float* __restrict__ w /*[Cout][Cin][3][3]*/, int Cout, int Cin, int CoutP, bf16_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = 4 * Cin * 4 * CoutP;
    if (i >= total) return;
    const int co = i % CoutP, tap = (i / CoutP) & 3, n = i / (4 * CoutP);
    const int q = n / Cin, ci = n - q * Cin;
    const int ph = q >> 1, pw = q & 1, da = tap >> 1, db = tap & 1;
    const int r = ph + 1 - 2 * da, s = pw + 1 - 2 * db;
    float v = 0.f;
    if (co < Cout && (unsigned)r < 3u && (unsigned)s < 3u) v = w[((co * Cin + ci) * 3 + r) * 3 + s];
    out[i] = f2bf(v);
}

extern "C" int ryolo_pack_s2d(const float* w, int Cout, int Cin, bf16_t* out, hipStream_t stream)
{
    if (!w || !out || Cout <= 0 || Cin <= 0) return RY_ERR_ARG;
    const int CoutP = (Cout + 31) / 32 * 32;
    const int total = 4 * Cin * 4 * CoutP;
    hipLaunchKernelGGL(pack_s2d_kernel, dim3((unsigned)ry_cdiv(total, 256)), dim3(256), 0, stream, w, Cout, Cin, CoutP, out);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

static int gemm_check(const ConvGemmParams& p)
{
    if (p.pool_idx && (!p.pool_dz || p.pool_ld % 8 || p.pool_ldi % 8 || p.Nout % 8 || (p.OH & 1) || (p.OW & 1) || p.OW >= 32768 || p.nclasses != 1 ||
                       p.oh_mul != 1 || p.ow_mul != 1 || p.OHf != p.OH || p.OWf != p.OW || p.cls[0].oh_add || p.cls[0].ow_add ||
                       (p.epi != EPI_RAW && p.epi != EPI_ACCUM) || p.s2d_cin))
        return RY_ERR_ARG;
    if (p.s2d_cin && (p.s2d_cin % 8 || p.Nout != 4 * p.s2d_cin || p.oh_mul != 2 || p.ow_mul != 2 || p.nclasses != 1 ||
                      (p.epi != EPI_RAW && p.epi != EPI_ACCUM)))
        return RY_ERR_ARG;
    if (!p.A || !p.W || !p.out || p.Cin <= 0 || p.Cin % BK || p.ldA % 8 || p.Nout <= 0 || p.nclasses < 1 || p.nclasses > 4)
        return RY_ERR_ARG;
    if (p.head_attrs) {
        if (p.head_attrs < 7 || p.Nout % p.head_attrs || p.head_och < 0 || p.head_och >= p.head_attrs || p.epi != EPI_F32_BIAS) return RY_ERR_ARG;
        if (!gemm_is_t1(p)) return RY_ERR_UNSUPPORTED;        // (the caller falls back to the row-major form + ryolo_head_finish_fwd)
    }
    return RY_OK;
}

// Which kernel ryolo_conv_gemm will run for these parameters and how many [2][Nout] partial-statistics rows its EPI_STATS
// epilogue writes (= number of M tiles).  kernel: 0 generic implicit GEMM (conv.hip), 1 3x3 halo-patch kernel (conv3x3.hip,
// selected by pipe bit 0x200 when the layer is eligible), 2 weight-stationary persistent 1x1 kernel (gemm1x1.hip; rows = waves),
// 3 persistent weight-stationary 3x3 kernel for 64 -> <= 64 channels (conv3x3_ws.hip; rows = workgroups), 4 the 256-wide pointwise GEMM for long
// reductions (gemm256.hip; bits 16-19 = tile columns / 32), 5 the streaming 3x3 stride-2 forward for 32 input channels (conv3x3s2_c32.hip;
// rows = workgroups), 6 the same layer's space-to-depth data gradient.
extern "C" int ryolo_conv_gemm_plan(const ConvGemmParams* pp, int* stats_rows, int* kernel)
{
    if (!pp || !stats_rows) return RY_ERR_ARG;
    const ConvGemmParams& p = *pp;
    if (p.Cin <= 0 || p.Cin % BK || p.Nout <= 0 || p.nclasses < 1 || p.nclasses > 4) return RY_ERR_ARG;
    if (p.head_attrs) {                                       // detection head in its final layout: the 128 x 128 1x1 instantiation or nothing
        if (p.head_attrs < 7 || p.Nout % p.head_attrs || p.epi != EPI_F32_BIAS) return RY_ERR_ARG;
        if (!gemm_is_t1(p) || (int64_t)p.NB * p.OH * p.OW > 0x7fffffff) return RY_ERR_UNSUPPORTED;
        *stats_rows = (int)ry_cdiv((int64_t)p.NB * p.OH * p.OW, 128);
        if (kernel) *kernel = 0 | 0x100 | (2 << 12) | (4 << 16);
        return RY_OK;
    }
    S2cGeom sg;
    if (s2c_geometry(p, sg)) {                                     // streaming 3x3 stride-2 forward, 32 input channels (conv3x3s2_c32.hip): one row per workgroup
        *stats_rows = sg.nwg;
        if (kernel) *kernel = 5;
        return RY_OK;
    }
    if (s2c_dgrad_geometry(p, sg)) {                               // ... and its data gradient in the space-to-depth form (no statistics epilogue)
        *stats_rows = sg.nwg;
        if (kernel) *kernel = 6;
        return RY_OK;
    }
    Ws3Geom w3;
    if ((p.pipe & 0x200) && ws3_geometry(p, w3)) {                 // persistent weight-stationary 3x3 (conv3x3_ws.hip): one statistics row per workgroup
        *stats_rows = w3.nwg;
        if (kernel) *kernel = 3;
        return RY_OK;
    }
    P3Geom g;
    if ((p.pipe & 0x200) && p3_geometry(p, g)) {
        *stats_rows = (int)g.gm;
        if (kernel) *kernel = 1 | ((g.BN / 32) << 16);            // bits 16-19 = tile columns / 32 (r06: 64-column tiles also for wide layers on small grids)
        return RY_OK;
    }
    G256Geom g2;
    if (g256_geometry(p, g2)) {                                     // 256-wide tiles for long-K pointwise layers (gemm256.hip): one statistics row per pixel tile
        *stats_rows = (int)g2.gm;
        if (kernel) *kernel = 4 | ((g2.BN / 32) << 16);
        return RY_OK;
    }
    Ws1Geom wg;
    if (p.nclasses == 1 && ws1_geometry(p, wg)) {
        *stats_rows = wg.stats_rows;
        if (kernel) *kernel = 2;
        return RY_OK;
    }
    // generic kernel: bits 8 = the 1x1 instantiation (T1), bits 12-15 = tile rows / 64, bits 16-19 = tile columns / 32 (what rocprofv3 lists
    // as separate kernels; tools and bench.py label their per-kernel tables with it)
    const int rows = gemm_tile_rows(p);
    const int cols = p.Nout <= 32 ? 32 : ((p.Nout <= 64 || ((p.pipe & 0xff) && (p.pipe & 0x800))) ? 64 : 128);
    if (kernel) *kernel = 0 | (gemm_is_t1(p) ? 0x100 : 0) | ((rows / 64) << 12) | ((cols / 32) << 16);
    *stats_rows = (int)ry_cdiv((int64_t)p.NB * p.OH * p.OW, rows);
    return RY_OK;
}

extern "C" int ryolo_conv_gemm(const ConvGemmParams* pp, hipStream_t stream)
{
    if (!pp) return RY_ERR_ARG;
    const ConvGemmParams& p = *pp;
    if (gemm_check(p)) return RY_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W)) & 15) return RY_ERR_ARG;
    for (int c = 0; c < p.nclasses; c++)
        if (p.cls[c].ntaps < 1 || p.cls[c].ntaps > RY_MAX_TAPS) return RY_ERR_ARG;
    if (p.epi == EPI_STATS && (!p.stats || p.nclasses != 1)) return RY_ERR_ARG;
    if ((int64_t)p.NB * p.OH * p.OW <= 0) return RY_OK;
    {
        S2cGeom sg;
        if (s2c_geometry(p, sg)) return s2c_launch(p, sg, stream);
        if (s2c_dgrad_geometry(p, sg)) return s2c_dgrad_launch(p, sg, stream);
    }
    if (p.pipe & 0x200) {
        Ws3Geom w3;
        if (ws3_geometry(p, w3)) return ws3_launch(p, w3, stream);
        P3Geom g;
        if (p3_geometry(p, g)) return p3_launch(p, g, stream);
    }
    if (p.head_attrs) {                                       // detection head in its final layout: the 1x1 instantiation of the generic kernel only
        if (!p.zeros) return RY_ERR_ARG;
        const int64_t M = (int64_t)p.NB * p.OH * p.OW;
        const int64_t gm = ry_cdiv(M, 128), gn = ry_cdiv(p.Nout, 128);
        if (gm * gn > 0x7fffffff || M > 0x7fffffff) return RY_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((conv_gemm_kernel<128, 128, 2, 2, 1, 32, 3, true, true>), dim3((unsigned)(gm * gn)), dim3(256), 0, stream, p);
        return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
    }
    if (p.pipe & 0xff) {
        if (!p.zeros) return RY_ERR_ARG;
        G256Geom g2;
        if (g256_geometry(p, g2)) return g256_launch(p, g2, stream);
        Ws1Geom wg;
        if (ws1_geometry(p, wg)) return ws1_launch(p, wg, stream);
        // 64-channel (full 128-byte line) stages: measured +1..5 % on 3x3 layers up to 256 channels, -4..-10 % on 1x1 / 512-channel
        // layers (tools/bench_conv.py matrix in DESIGN.md); 0x100 forces 32-channel stages for A/B runs
        const bool k64 = gemm_k64(p);
        if (p.Nout <= 32) return launch_gemm<256, 32, 4, 1, 1>(p, stream);
        // <= 64 output columns: 256 x 64 tiles (each wave 64 x 64: 8 MFMAs per K step; the 128 x 64 tile gives a wave 64 x 32 = 4 MFMAs per
        // step around the same barrier / DMA issue) when the grid still fills the chip; RYOLO_GEMM_N64 = 0 restores 128 x 64 (A/B)
        if (gemm_wide_n64(p)) return launch_gemm<256, 64, 4, 1, 1>(p, stream);
        if (const int deep = gemm_deep_stages(p)) return deep == 6 ? launch_gemm_deep<6>(p, stream) : launch_gemm_deep<4>(p, stream);
        if (p.Nout <= 64 || (p.pipe & 0x800)) return k64 ? launch_gemm<128, 64, 2, 2, 1, 64>(p, stream) : launch_gemm<128, 64, 2, 2, 1>(p, stream);   // 0x800: A/B, 64-wide N tiles everywhere
        return k64 ? launch_gemm<128, 128, 2, 2, 1, 64>(p, stream) : launch_gemm<128, 128, 2, 2, 1>(p, stream);
    }
    if (p.Nout <= 32) return launch_gemm<256, 32, 4, 1, 0>(p, stream);
    if (p.Nout <= 64) return launch_gemm<128, 64, 2, 2, 0>(p, stream);
    return launch_gemm<128, 128, 2, 2, 0>(p, stream);
}

static int wgrad_geometry(WgradParams& p, int& bm, int& gx, int& gy)
{
    if (p.Cin <= 0 || p.Cin % BK || p.ldX % 8 || p.ldY % 8 || p.Cout <= 0 || p.CoutPad % 8 || p.CoutPad < p.Cout || p.CoutPad > p.ldY ||
        p.ntaps < 1 || p.ntaps > RY_MAX_TAPS)
        return RY_ERR_ARG;
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    bm = p.Cout <= 64 ? 64 : 128;
    gx = (int)ry_cdiv(p.Cout, bm);
    const int cpt = (bm == 64 && p.ntaps == 9 && p.Cin == BK) ? 3 : 4;      // conv_wgrad_kernel<64>: one kernel row per column tile
    gy = (int)ry_cdiv((int64_t)p.ntaps * (p.Cin / BK), cpt);
    // 512 = 2 workgroups x 256 CUs (r04: +0.4 % step over 768, a third fewer split-K slabs; 384 equal, 256 -0.8 %; r01-r03 measured 768 best of
    // 768 / 1024 / 1536 / 2560 against the BatchNorm kernels of those rounds); env knob for A/B runs
    static const int target = getenv("RYOLO_WGRAD_BLOCKS") ? atoi(getenv("RYOLO_WGRAD_BLOCKS")) : 512;
    int64_t want = ry_cdiv(target, (int64_t)gx * gy);                // one full wave of resident workgroups by default
    int64_t maxsplit = ry_cdiv(M, 16 * BK);                          // at least 16 K-steps per split
    int64_t sk = want > maxsplit ? maxsplit : want;
    if (sk < 1) sk = 1;
    if (sk > 65535) sk = 65535;
    p.kchunk = ry_cdiv(ry_cdiv(M > 0 ? M : 1, sk), BK) * BK;
    p.splitk = (int)ry_cdiv(M > 0 ? M : 1, p.kchunk);
    return RY_OK;
}

extern "C" int ryolo_conv_wgrad_plan(const WgradParams* pp, int* splitk, size_t* workspace_bytes)
{
    if (!pp || !splitk || !workspace_bytes) return RY_ERR_ARG;
    WgradParams p = *pp;
    int bm, gx, gy;
    const int rc = wgrad_geometry(p, bm, gx, gy);
    if (rc) return rc;
    W3Geom g3;
    if (w3_geometry(p, g3)) p.splitk = g3.slabs;               // 3x3 stride-1 layers: halo-ring kernel (conv3x3.hip)
    else {
        int sk8, gx8, gy8;
        int64_t kc8;
        if (w1x8_geometry(p, &sk8, &kc8, &gx8, &gy8)) p.splitk = sk8;      // wide pointwise layers: 8-wave 256 x 256 tiles (wgrad1x1_8w.hip)
    }
    *splitk = p.splitk;
    *workspace_bytes = (size_t)p.splitk * p.Cout * p.ntaps * p.Cin * sizeof(float);
    return RY_OK;
}

static bool wgrad_pointwise(const WgradParams& p)
{
    return p.ntaps == 1 && p.dh[0] == 0 && p.dw[0] == 0 && p.sh == 1 && p.sw == 1 && p.IH == p.OH && p.IW == p.OW && p.OH * p.OW > 1 &&
           !(getenv("RYOLO_WGRAD_P1") && (atoi(getenv("RYOLO_WGRAD_P1")) & 1) == 0);
}
// tapped / strided layers with more than 64 output channels: the LDS-DMA ring with the gather on the request side (RYOLO_WGRAD_TAPS_DMA=0: A/B)
static bool wgrad_taps_dma(const WgradParams& p, int bm)
{
    static const int on = getenv("RYOLO_WGRAD_TAPS_DMA") ? atoi(getenv("RYOLO_WGRAD_TAPS_DMA")) : 1;
    return on && !wgrad_pointwise(p) && bm == 128 && p.zeros && ((reinterpret_cast<uintptr_t>(p.dY) | reinterpret_cast<uintptr_t>(p.X)) & 15) == 0 &&
           (int64_t)p.NB * p.OH * p.OW < (1ll << 31);
}

// 0: generic split-K kernels (conv.hip: register-staged, or the LDS-DMA pointwise form), 1: 3x3 stride-1 halo-ring kernel (conv3x3.hip),
// 2: tapped LDS-DMA kernel (conv.hip), 3: the 8-wave 256 x 256 pointwise kernel (wgrad1x1_8w.hip) — what ryolo_conv_wgrad will launch
// (4 was the parity-plane ring kernel for 3x3 stride-2 layers, r05: parity-green, step-neutral, retired in r06 — git history keeps it)
extern "C" int ryolo_conv_wgrad_kernel(const WgradParams* pp, int* kernel)
{
    if (!pp || !kernel) return RY_ERR_ARG;
    W3Geom g3;
    *kernel = 0;
    if (w3_geometry(*pp, g3)) { *kernel = 1; return RY_OK; }
    WgradParams p = *pp;
    {
        int sk8, gx8, gy8;
        int64_t kc8;
        if (w1x8_geometry(p, &sk8, &kc8, &gx8, &gy8)) { *kernel = 3; return RY_OK; }
    }
    int bm, gx, gy;
    if (wgrad_geometry(p, bm, gx, gy) == RY_OK && wgrad_taps_dma(p, bm)) *kernel = 2;
    return RY_OK;
}

// launch shape of the split-K kernel ryolo_conv_wgrad will use: workgroups and waves per workgroup.  The 8-wave kernels (conv3x3_wgrad8.hip,
// wgrad1x1_8w.hip) hold a CU exclusively (2 x ~200+ registers per SIMD lane, 76-150 KiB of LDS) and are sized to PART of the chip: a launch timed
// alone then occupies `workgroups` of the 256 CUs — bench.py prices such a launch against the CUs it holds as well as against the whole chip.
extern "C" int ryolo_conv_wgrad_grid(const WgradParams* pp, int* workgroups, int* waves)
{
    if (!pp || !workgroups || !waves) return RY_ERR_ARG;
    WgradParams p = *pp;
    int bm, gx, gy;
    const int rc = wgrad_geometry(p, bm, gx, gy);
    if (rc) return rc;
    W3Geom g3;
    if (w3_geometry(p, g3)) {
        *workgroups = g3.gx * g3.gc * g3.splitk;
        *waves = g3.v8 ? 8 : 4;
        return RY_OK;
    }
    int sk8, gx8, gy8;
    int64_t kc8;
    if (w1x8_geometry(p, &sk8, &kc8, &gx8, &gy8)) {
        *workgroups = gx8 * gy8 * sk8;
        *waves = 8;
        return RY_OK;
    }
    *workgroups = gx * gy * p.splitk;
    *waves = 4;
    return RY_OK;
}

extern "C" int ryolo_conv_wgrad(const WgradParams* pp, hipStream_t stream)
{
    if (!pp) return RY_ERR_ARG;
    WgradParams p = *pp;
    if (!p.dY || !p.X || !p.dW || !p.partial) return RY_ERR_ARG;
    int bm, gx, gy;
    const int rc = wgrad_geometry(p, bm, gx, gy);
    if (rc) return rc;
    if ((int64_t)p.NB * p.OH * p.OW <= 0) return RY_OK;
    W3Geom g3;
    if (w3_geometry(p, g3)) {
        const int rc3 = w3_launch(p, g3, stream);
        if (rc3) return rc3;
        launch_wgrad_reduce(p, g3.slabs, stream);
        RY_CHECK_LAUNCH();
        return RY_OK;
    }
    {
        int sk8, gx8, gy8;
        int64_t kc8;
        if (w1x8_geometry(p, &sk8, &kc8, &gx8, &gy8)) {
            const int rc8 = w1x8_launch(p, stream);
            if (rc8) return rc8;
            launch_wgrad_reduce(p, sk8, stream);
            RY_CHECK_LAUNCH();
            return RY_OK;
        }
    }
    if ((int64_t)p.NB * p.OH * p.OW >= (1ll << 31)) return RY_ERR_UNSUPPORTED;       // 32-bit pixel indices in the gather
    auto wg_magic = [](unsigned d, unsigned& m, unsigned& sh) {      // n / d == mulhi(n, m) >> sh for 0 <= n < 2^31
        if (d < 2) { m = 0; sh = 0; return; }                        // division by one: flagged with m == 0
        unsigned l = 0;
        while ((1ull << l) < d) l++;
        m = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
        sh = l - 1;
    };
    WgMagic mg;
    wg_magic((unsigned)(p.OH * p.OW), mg.m_img, mg.s_img);
    wg_magic((unsigned)p.OW, mg.m_row, mg.s_row);
    const bool p1 = wgrad_pointwise(p);
    const dim3 wgrid((unsigned)((int64_t)gx * gy * p.splitk));
    // pointwise layers wider than 64 output channels: the LDS-DMA ring kernel (same tiles, same slabs: bm == 128 gives gx = ceil(Cout / 128),
    // gy = ceil(Cin / 128) there too); 0x2 in RYOLO_WGRAD_P1 switches it off for A/B runs
    // bit 0 pointwise addressing, bit 1 the LDS-DMA kernel, bit 2 its 64-pixel steps (on since the end of r04: with two workgroups per CU on the side
    // stream — wgrad_geometry — the two-stage 64-pixel form is +0.45 % on the step, three alternating runs; at three per CU, r03, it was neutral)
    static const int p1_mode = getenv("RYOLO_WGRAD_P1") ? atoi(getenv("RYOLO_WGRAD_P1")) : 7;
    if (p1 && bm == 128 && (p1_mode & 2) && p.zeros && ((reinterpret_cast<uintptr_t>(p.dY) | reinterpret_cast<uintptr_t>(p.X)) & 15) == 0) {
        if (p1_mode & 4) hipLaunchKernelGGL(wgrad1x1_dma_kernel<64>, wgrid, dim3(256), 0, stream, p);      // 0x4: 64-pixel K steps (A/B)
        else hipLaunchKernelGGL(wgrad1x1_dma_kernel<32>, wgrid, dim3(256), 0, stream, p);
        launch_wgrad_reduce(p, p.splitk, stream);
        RY_CHECK_LAUNCH();
        return RY_OK;
    }
    if (wgrad_taps_dma(p, bm)) {
        hipLaunchKernelGGL(wgrad_taps_dma_kernel, wgrid, dim3(256), 0, stream, p, mg);
        launch_wgrad_reduce(p, p.splitk, stream);
        RY_CHECK_LAUNCH();
        return RY_OK;
    }
    if (bm == 64) {
        if (p1) hipLaunchKernelGGL((conv_wgrad_kernel<64, true>), wgrid, dim3(256), 0, stream, p, mg);
        else hipLaunchKernelGGL((conv_wgrad_kernel<64, false>), wgrid, dim3(256), 0, stream, p, mg);
    } else {
        if (p1) hipLaunchKernelGGL((conv_wgrad_kernel<128, true>), wgrid, dim3(256), 0, stream, p, mg);
        else hipLaunchKernelGGL((conv_wgrad_kernel<128, false>), wgrid, dim3(256), 0, stream, p, mg);
    }
    launch_wgrad_reduce(p, p.splitk, stream);
    RY_CHECK_LAUNCH();
    return RY_OK;
}
