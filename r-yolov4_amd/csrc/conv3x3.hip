// 3x3 stride-1 convolution (forward and data gradient) as an implicit GEMM over a HALO PATCH held in LDS — gfx950 only.
// Reference rows served: SURVEY.md §8a M1/M2 (`Conv`, model/utils.py:6-32, every 3x3 s1 conv of the ELAN / CSP / SPPCSPC
// blocks: 54 % of the conv FLOPs of yolov7 @800^2) and their data gradients.
//
// Why a second kernel: the generic conv_gemm loop (conv.hip) gathers a fresh (pixels x 32 channels) A tile for each of the
// 9 taps, so the same input bytes cross L2 -> TA -> LDS nine times and the loop is bound by the number of LDS-DMA
// instructions per MFMA (PMC + ablations in DESIGN.md §4.1).  Here a workgroup owns 256 output pixels and 64 / 128 output
// channels; per 32-channel chunk it brings the input pixels those outputs touch ONCE (the tile plus its halo: 324 ... 450
// rows instead of 9 x 256) and all 9 taps read shifted rows of that patch:
//   * tile = TH x TW output pixels of one image (2-D mode, patch (TH+2) x (TW+2)) or a flat run of 256 consecutive
//     output pixels (run mode, patch = run + one image row + 1 on both sides; zero tile waste on small maps);
//   * A-fragment row of output pixel m for tap (dh, dw) = patch row base(m) + (dh+1)*PW + (dw+1): one scalar add per tap;
//     padding (and, in run mode, wrap-around into the neighbouring row / image) is a per-lane 9-bit mask that redirects
//     the read to a 64-byte zero row — no zero-filled copies, no per-tap DMA;
//   * the patch of chunk c+1 streams in one 1-KiB piece per wave per tap step while chunk c is consumed (double buffer);
//     weights use a 3-slot ring, one (tap, chunk) tile of BN x 32 per step; counted `s_waitcnt vmcnt(N)` + ONE s_barrier
//     per step, nothing is drained to zero inside the loop;
//   * 4 waves, 128 x 64 per wave for BN = 128 (16 MFMA per 12 ds_read_b128 per step and per-lane DMA cost / MFMA 0.18 vs
//     0.5 in the generic loop), 64 x 64 for BN = 64; two workgroups per CU (<= 80 KiB LDS each);
//   * LDS images are lane-linear DMA targets; the bank swizzle (slot ^= (row >> 2) & 3, conflict-free ds_read_b128) is keyed
//     on the PATCH row and applied on the source address, the reader recomputes it per tap (3 VALU);
//   * epilogues as in conv.hip: raw bf16, training BatchNorm statistics, folded BN + activation (inference), accumulate.
#include "conv_internal.h"
#include <stdlib.h>
#include <type_traits>

#define P3_BM 256

extern __shared__ __attribute__((aligned(1024))) unsigned char p3_lds[];

template <int K> __device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
}

template <int U, int N> struct P3Unroll {
    template <class F> static __device__ __forceinline__ void run(F& f)
    {
        f(std::integral_constant<int, U>{});
        P3Unroll<U + 1, N>::run(f);
    }
};
template <int N> struct P3Unroll<N, N> {
    template <class F> static __device__ __forceinline__ void run(F&) {}
};

// EPI is a template parameter: with the epilogue selected by run-time branches the one kernel body carried every variant (17 k instructions
// behind the loop) and the training epilogues ran 10-13 k cycles per workgroup.  (r02-r05 also carried a `BS` parameter: BatchNorm-backward sums
// folded into the store loop — slower than the stand-alone reduce pass at every size since r04, retired in r06.)
template <int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_patch_kernel(const ConvGemmParams p, const P3Geom g)
{
    constexpr int BM = P3_BM, TM = BM / WM / 32, TN = BN / WN / 32, WTM = BM / WM, WTN = BN / WN;
    constexpr int NPB = BN / 64;                               // weight pieces (16 rows x 64 B) per wave per step
    constexpr int WSLOT = BN * 64;                             // bytes of one weight ring slot
    static_assert(WM * WN == 4 && TM % 2 == 0 && (BN == 64 || BN == 128), "tile config");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5;
#ifdef P3_TIMING
    const unsigned long long T0 = __builtin_readcyclecounter();
#endif
    const int H = p.OH, W = p.OW;
    const int64_t HW = (int64_t)H * W, M = (int64_t)p.NB * HW;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int mb = tile / g.gn, nb = tile - mb * g.gn;
    const int n0 = nb * BN;

    const unsigned patch_bytes = (unsigned)g.P * 1024u;
    const unsigned wring_off = 2u * patch_bytes;
    const unsigned zrow_off = wring_off + 3u * WSLOT;

    // ---- tile origin --------------------------------------------------------------------------------------------------
    int img0 = 0, oh0 = 0, ow0 = 0;
    int64_t p0 = 0;
    if (g.mode == 1) {
        img0 = mb / g.tilesPerImg;
        const int rem = mb - img0 * g.tilesPerImg;
        const int th = rem / g.tilesW;
        oh0 = th * g.TH;
        ow0 = (rem - th * g.tilesW) * g.TW;
    } else {
        p0 = (int64_t)mb * BM;
    }

    // ---- patch DMA sources -> LDS table ptab[u][tid]: piece (wave + 4u) = patch rows 16*piece ... +15, lane -> (row, slot) ---------
    // (element offset / 8 into A; ~0u -> zero page).  Kept in LDS, not in 9 VGPRs; the FIRST chunk's pieces are requested here, entry by
    // entry, so that the HBM round trip of the patch runs under the rest of the prologue (the address tables below are pure VALU work).
    unsigned* const ptab = reinterpret_cast<unsigned*>(p3_lds + zrow_off + 128);
    // output pixel of tile row m (-1: dead row), one entry per thread: the store loop of the epilogue reads it back instead of redoing the
    // tile -> image index arithmetic for each of its 16 rows per lane (35 VALU instructions per row, a quarter of the store loop)
    int* const rowpix = reinterpret_cast<int*>(p3_lds + zrow_off + 128 + (unsigned)g.TP * 1024u);
    {
        const int mrow = tid;
        int64_t pix;
        bool live;
        if (g.mode == 1) {
            const int rr = small_div(mrow, g.TW, g.rTW), c = mrow - rr * g.TW;
            live = mrow < g.TH * g.TW;
            pix = ((int64_t)img0 * H + oh0 + rr) * W + ow0 + c;
        } else {
            pix = p0 + mrow;
            live = pix < M;
        }
        rowpix[mrow] = live ? (int)pix : -1;
    }
    for (int u = 0; u < g.TP; u++) {
        const int pc = min(wave + 4 * u, g.P - 1);
        const int j = pc * 16 + (lane >> 2);
        bool ok = j < g.R;
        int64_t pix;
        if (g.mode == 1) {
            const int pr = small_div(j, g.PW, g.rPW), pcx = j - pr * g.PW;
            const int ih = oh0 - 1 + pr, iw = ow0 - 1 + pcx;
            ok = ok && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            pix = ((int64_t)img0 * H + ih) * W + iw;
        } else {
            pix = p0 - W - 1 + j;
            ok = ok && pix >= 0 && pix < M;
        }
        const int sl = (lane & 3) ^ ((j >> 2) & 3);
        const unsigned ofs = ok ? (unsigned)((pix * p.ldA + sl * 8) >> 3) : 0xffffffffu;
        ptab[u * 256 + tid] = ofs;
        const bf16_t* src = ofs != 0xffffffffu ? p.A + ((int64_t)ofs << 3) : p.zeros;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(p3_lds + pc * 1024), 16, 0, 0);
    }
    // ---- weight DMA sources; steps 0 and 1 requested now ----------------------------------------------------------------------
    unsigned b_ofs[NPB];                                        // element offset into W (< 2^31: checked on the host), ~0u -> zero page
#pragma unroll
    for (int u = 0; u < NPB; u++) {
        const int r = (wave + 4 * u) * 16 + (lane >> 2);
        b_ofs[u] = (n0 + r) < p.Nout ? (unsigned)((int64_t)(n0 + r) * p.wtaps * p.Cin + ((lane & 3) ^ ((r >> 2) & 3)) * 8) : 0xffffffffu;
    }
    auto issue_w = [&](int w_off, unsigned dst) {
#pragma unroll
        for (int u = 0; u < NPB; u++) {
            const bf16_t* src = b_ofs[u] != 0xffffffffu ? p.W + b_ofs[u] + w_off : p.zeros;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(p3_lds + dst + (wave + 4 * u) * 1024), 16, 0, 0);
        }
    };
    issue_w(g.twi[0] * p.Cin, wring_off);
    issue_w(g.twi[1] * p.Cin, wring_off + WSLOT);
    // ---- A fragment addresses.  atab[i][t]: byte offset INSIDE a patch buffer of the 16-byte fragment piece (16-channel half 0; half 1: ^ 32)
    // that output pixel (i, lane & 31) reads for tap t — patch row base(m) + (dh+1)*PW + (dw+1), bank swizzle applied; a tap that falls on
    // padding (or, for flat runs, wraps into the neighbouring row / image) points at the buffer's LAST row instead, which every chunk's DMA
    // fills from the zero page (the host sizes P so that 16 P > R).  36 registers, computed once: a tap step starts with its LDS reads — the
    // round-3 loop recomputed these addresses behind the barrier of every step (tap scalar from LDS -> wait -> 24 VALU -> first read).
    int run_oh = 0, run_ow = 0;
    const float rH = 1.0f / (float)H;
    if (g.mode != 1) {
        const int rem = (int)(p0 % HW);                           // wave-uniform: scalar division
        run_oh = rem / W;
        run_ow = rem - run_oh * W;
    }
    const unsigned zrel = (unsigned)(g.P * 16 - 1) << 6;
    // tap scalars first, all of them (kernarg dwords: one batch of s_loads, one wait — read inside the loops below they came back one at a time,
    // 41 waits of a scalar-memory round trip each in the first cut of this table)
    int toff[9], tbit[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        toff[t] = (g.tdh[t] + 1) * g.PW + (g.tdw[t] + 1);
        tbit[t] = (g.tdh[t] + 1) * 3 + (g.tdw[t] + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned atab[TM][9];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mrow = wm * WTM + i * 32 + (lane & 31);
        bool live;
        int oh, ow, abase;
        if (g.mode == 1) {
            const int r = small_div(mrow, g.TW, g.rTW), c = mrow - r * g.TW;
            live = mrow < g.TH * g.TW;
            oh = oh0 + r;
            ow = ow0 + c;
            abase = live ? r * g.PW + c : 0;
        } else {
            const int64_t pp = p0 + mrow;
            live = pp < M;
            // (oh, ow) of a flat pixel index: the run starts at (run_oh, run_ow) — computed once per workgroup with scalar math above
            int o = run_ow + mrow, orow = run_oh;
            const int wraps = small_div(o, W, g.rPW);               // PW == W for flat runs
            o -= wraps * W;
            orow += wraps;
            orow -= small_div(orow, H, rH) * H;                      // next image(s): row index modulo H
            oh = orow;
            ow = o;
            abase = mrow;
        }
        // tap (dh, dw) reads input (oh + dh, ow + dw): bit 3*(dh+1) + (dw+1) of m9 says whether that pixel exists (branch-free selects below:
        // hipcc turned `ok ? address : zero row` into 36 divergent branches)
        const unsigned colok = ((ow >= 1) ? 1u : 0u) | 2u | ((ow + 1 < W) ? 4u : 0u);
        unsigned m9 = (colok << 3) | ((oh >= 1) ? colok : 0u) | ((oh + 1 < H) ? (colok << 6) : 0u);
        if (!live) m9 = 0u;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const unsigned keep = 0u - ((m9 >> tbit[t]) & 1u);          // all ones / zero
            const unsigned prow = (unsigned)(abase + toff[t]);
            const unsigned a = (prow << 6) | ((((prow >> 2) & 3u) ^ (unsigned)h) << 4);
            atab[i][t] = (a & keep) | (zrel & ~keep);
        }
    }
    unsigned fb[TN];                                            // B fragment byte offset inside a ring slot for ks = 0 (ks = 1: ^ 32)
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int r = wn * WTN + j * 32 + (lane & 31);
        fb[j] = (unsigned)(r * 64 + ((h ^ ((r >> 2) & 3)) << 4));
    }
    // (no workgroup barrier here: ptab is read back by the thread that wrote it, rowpix only in the epilogue, and a __syncthreads would drain
    // the prologue's DMA — the first tap step waits for exactly what it needs)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const int cchunks = p.Cin >> 5;
    const int nsteps = cchunks * 9;
#ifdef P3_TIMING
    const unsigned long long T1 = __builtin_readcyclecounter();
#endif
    // Fragments of the COMING step that may be read before its barrier: its weights landed one barrier earlier (see the waits below) and
    // the patch of a chunk is complete two steps before the chunk starts (TP <= 7), so a step begins with its first MFMAs' operands already in
    // flight instead of barrier -> ds_read -> LDS latency -> first MFMA.
    constexpr int PF = 2;
    bf16x8 bnext[TN], anext[PF];
    auto pre_issue = [&](auto tcst, unsigned buf) {
        constexpr int t = decltype(tcst)::value;
#pragma unroll
        for (int j = 0; j < TN; j++) bnext[j] = *reinterpret_cast<const bf16x8*>(p3_lds + wring_off + (t % 3) * WSLOT + fb[j]);
#pragma unroll
        for (int u = 0; u < PF; u++) anext[u] = *reinterpret_cast<const bf16x8*>(p3_lds + (atab[u][t] + buf));
    };
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();                              // chunk 0 of the patch, weights of steps 0 and 1: landed, visible
    pre_issue(std::integral_constant<int, 0>{}, 0u);
    int s = 0;
    for (int cc = 0; cc < cchunks; cc++) {
        const bool more = cc + 1 < cchunks;
        const unsigned pbuf = (cc & 1) ? patch_bytes : 0u;
        const unsigned nbuf = (cc & 1) ? 0u : patch_bytes;
        // the 9 tap steps of a chunk are unrolled by hand (template recursion): tap index, ring slot and the atab column are compile-time
        auto step = [&](auto tcst) {
            constexpr int t = decltype(tcst)::value;
            constexpr int slot = t % 3;                        // ring slot of step (cc, t) (9 taps per chunk)
            // Barrier of step s.  Behind it: (a) every wave's DMA pieces of step s+1's weights have landed (each wave waits for its own:
            // they were requested in the FIRST units of step s-1, a step ago), so step s+1's fragments may be read any time after this
            // barrier; (b) every wave has finished reading step s-1's ring slot, which this step's weight requests overwrite.
            // The patch piece of step s-1 — requested last, streamed from HBM — may stay in flight (vmcnt counts in order).
            const bool prev_piece = t != 0 && (t - 1) < g.TP && more;
            if (prev_piece) wait_vm<1>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            const bool do_patch = more && t < g.TP;
            const bool do_w = s + 2 < nsteps;
            constexpr int t2 = (t + 2) % 9;
            const int w_off = g.twi[t2] * p.Cin + (cc + (t + 2 >= 9 ? 1 : 0)) * 32;
            const unsigned w_dst = wring_off + ((slot + 2) % 3) * WSLOT;                  // slot of step s + 2
            const unsigned wb = wring_off + slot * WSLOT;
            // this step's patch piece: its source offset is read together with the first fragments (one lgkmcnt wait covers both)
            const int pu = min(t, g.TP - 1);
            const unsigned pofs = ptab[pu * 256 + tid];
            const int ppc = min(wave + 4 * pu, g.P - 1);
            // Software pipeline over the 2*TM "units" (16-channel half ks, A fragment i): the A fragment of unit u+PF is read while
            // unit u's TN MFMAs run; B fragments of a half are read one unit before its first use.  (All-reads-first exposes the
            // LDS latency once per step; hipcc's own schedule funnels every A fragment through one register quad and waits on each.)
            // The step's DMA instructions (NPB weight pieces, then 1 patch piece per wave) follow the first units: an LDS-DMA
            // issue costs 100-185 cycles next to ds_reads but hides in the shadow of the matrix pipe (guide's price table).
            // Operands are swapped (A = weights, B = pixels): the accumulator holds the TRANSPOSED tile, so a lane owns 4
            // consecutive channels of one pixel — 8-byte packed stores in the epilogue.
            constexpr int NU = 2 * TM, NIT = NPB + 1;
            static_assert(NU >= NIT + 1 && NU - 2 >= TM, "DMA items and the pre-issue point fit in the unit sequence");
            bf16x8 af[NU], bfr[2][TN];
            auto read_a = [&](int u) { return *reinterpret_cast<const bf16x8*>(p3_lds + ((atab[u % TM][t] ^ (unsigned)((u / TM) << 5)) + pbuf)); };
#pragma unroll
            for (int j = 0; j < TN; j++) bfr[0][j] = bnext[j];
#pragma unroll
            for (int u = 0; u < PF; u++) af[u] = anext[u];
#pragma unroll
            for (int u = 0; u < NU; u++) {
                __builtin_amdgcn_sched_barrier(0);
                if (u + PF == TM) {
#pragma unroll
                    for (int j = 0; j < TN; j++) bfr[1][j] = *reinterpret_cast<const bf16x8*>(p3_lds + wb + (fb[j] ^ 32u));
                }
                if (u + PF < NU) af[u + PF] = read_a(u + PF);
                if (u == NU - 2) {                                       // bfr[0], af[0], af[1] are dead: the coming step's first fragments
                    if constexpr (t + 1 < 9) pre_issue(std::integral_constant<int, t + 1>{}, pbuf);
                    else pre_issue(std::integral_constant<int, 0>{}, nbuf);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[u % TM][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[u / TM][j], af[u], acc[u % TM][j], 0, 0, 0);
                // DMA item k after unit k: weights first, the patch piece last
                if (u < NIT) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (u < NPB) {
                        if (do_w) {
                            const bf16_t* src = b_ofs[u] != 0xffffffffu ? p.W + b_ofs[u] + w_off : p.zeros;
                            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(p3_lds + w_dst + (wave + 4 * u) * 1024), 16, 0, 0);
                        }
                    } else if (do_patch) {
                        const bf16_t* src = pofs != 0xffffffffu ? p.A + ((int64_t)pofs << 3) + (cc + 1) * 32 : p.zeros;
                        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(p3_lds + nbuf + ppc * 1024), 16, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            s++;
        };
        P3Unroll<0, 9>::run(step);
    }
#ifdef P3_TIMING
    const unsigned long long T2 = __builtin_readcyclecounter();
#endif
    __syncthreads();                                           // operand tiles dead: LDS is reused for the output staging
#ifdef P3_TIMING
    unsigned long long te[6] = {__builtin_readcyclecounter(), 0, 0, 0, 0, 0};    // sync | stage writes | statistics | store loop (both halves summed)
#endif

    // ---- epilogue.  acc[i][j] is the transposed 32x32 tile: column = lane & 31 = pixel i*32 + (lane & 31), row = channel
    // j*32 + (e & 3) + 8*(e >> 2) + 4*(lane >> 5): 4 consecutive channels per lane per register quad -> one 8-byte LDS store
    // each (32 per wave instead of the 128 two-byte stores of the untransposed layout), then 16-byte row segments to HBM.
    constexpr int EP_LD = WTN + 8;                             // staging row stride (bf16): 16-byte aligned, breaks bank aliasing
    bf16_t* const stage = reinterpret_cast<bf16_t*>(p3_lds) + wave * 64 * EP_LD;      // 64 pixel rows per pass
    constexpr int CH = WTN / 8;                                // 16-byte chunks per staged row
    constexpr int RPI = 64 / CH;                               // rows per store iteration
    const int ch = lane % CH, r0 = lane / CH;
    const int ncol = n0 + wn * WTN + ch * 8;
    const int cq = lane & 15, rg = lane >> 4;                  // statistics: 4 channels x every 4th row per lane
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    // inference: the folded BatchNorm coefficients of the tile's BN columns, once per workgroup into LDS behind the staging rows (inside the dead
    // patch buffers) instead of two 16-byte global loads per staged quad
    float* const cscale = reinterpret_cast<float*>(p3_lds + 4 * 64 * EP_LD * 2);
    float* const cshift = cscale + BN;
    if constexpr (EPI == EPI_AFFINE_ACT) {
        if (tid < BN) {
            const int n = n0 + tid;
            cscale[tid] = n < p.Nout ? p.scale[n] : 0.f;
            cshift[tid] = n < p.Nout ? p.shift[n] : 0.f;
        }
        __syncthreads();
    }
    // accumulate epilogue: the old values of the WHOLE tile are requested here, in front of the staging writes — one memory round trip per
    // workgroup, most of it under the staging of the first half (requested per group of 4 rows inside the store loop it was four round trips:
    // 12 k of the epilogue's 13.8 k cycles)
    constexpr int NITA = 64 / RPI;
    uint4 oldall[TM / 2][NITA];
    if constexpr (EPI == EPI_ACCUM) {
        const int ncol_a = ncol < p.Nout ? ncol : 0;
#pragma unroll
        for (int half = 0; half < TM / 2; half++)
#pragma unroll
            for (int it = 0; it < NITA; it++) {
                const int pi = rowpix[wm * WTM + half * 64 + it * RPI + r0];
                const int64_t pix = (pi >= 0 && ncol < p.Nout) ? (int64_t)pi : 0;      // dead rows read (and discard) pixel 0 of their own columns
                oldall[half][it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.out) + pix * p.ldC + ncol_a);
            }
    }
#pragma unroll
    for (int half = 0; half < TM / 2; half++) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int i = half * 2 + ii;
                    const int c0 = j * 32 + 8 * g4 + 4 * h;
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = acc[i][j][4 * g4 + q];
                    if constexpr (EPI == EPI_AFFINE_ACT) {
                        const float4 sc = *reinterpret_cast<const float4*>(cscale + wn * WTN + c0);     // (columns >= Nout are never stored)
                        const float4 sh = *reinterpret_cast<const float4*>(cshift + wn * WTN + c0);
                        const float sc4[4] = {sc.x, sc.y, sc.z, sc.w}, sf4[4] = {sh.x, sh.y, sh.z, sh.w};
                        act_affine_quad(v, sc4, sf4, p.act);
                    }
                    const uint2 w = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    *reinterpret_cast<uint2*>(stage + (ii * 32 + (lane & 31)) * EP_LD + c0) = w;
                }
#ifdef P3_TIMING
        { const unsigned long long t = __builtin_readcyclecounter(); te[1] += t - (half ? te[5] : te[0]); te[4] = t; }
#endif
        // (same-wave LDS hand-off: the wave's own ds_write -> ds_read ordering is enough, no workgroup barrier)
        if constexpr (EPI == EPI_STATS) {
            // BatchNorm batch statistics of the values actually stored (bf16-rounded); dead rows hold exact zeros
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint2 w = *reinterpret_cast<const uint2*>(stage + (rg + 4 * k) * EP_LD + cq * 4);
                const float f0 = __uint_as_float(w.x << 16), f1 = __uint_as_float(w.x & 0xffff0000u);
                const float f2 = __uint_as_float(w.y << 16), f3 = __uint_as_float(w.y & 0xffff0000u);
                ssum[0] += f0; ssq[0] += f0 * f0;
                ssum[1] += f1; ssq[1] += f1 * f1;
                ssum[2] += f2; ssq[2] += f2 * f2;
                ssum[3] += f3; ssq[3] += f3 * f3;
            }
        }
#ifdef P3_TIMING
        { const unsigned long long t = __builtin_readcyclecounter(); te[2] += t - te[4]; te[4] = t; }
#endif
        // Store loop in two phases: every global load of the half (the old value of an accumulate epilogue) is issued before the first one
        // is used.  With the loads inside one loop next to `continue` branches each of the 8 iterations paid its own memory round trip:
        // +24 us on a 77 us launch for EPI_ACCUM (128->128 @50^2).
        constexpr int NIT = 64 / RPI, GRP = 4;                          // 4 iterations in flight: more would cost a resident workgroup (VGPRs)
        static_assert(NIT % GRP == 0, "store loop grouping");
        constexpr bool accum = EPI == EPI_ACCUM;
#pragma unroll
        for (int g0 = 0; g0 < NIT; g0 += GRP) {
            int64_t pixv[GRP];
            bool lv[GRP];
            uint4 oldv[GRP];
#pragma unroll
            for (int k = 0; k < GRP; k++) {
                const int r = (g0 + k) * RPI + r0;
                const int pi = rowpix[wm * WTM + half * 64 + r];
                const bool live = pi >= 0 && ncol < p.Nout;
                lv[k] = live;
                pixv[k] = live ? (int64_t)pi : 0;                        // dead rows read (and discard) pixel 0 of their own columns
            }
            if constexpr (accum) {
#pragma unroll
                for (int k = 0; k < GRP; k++) oldv[k] = oldall[half][g0 + k];
            }
#pragma unroll
            for (int k = 0; k < GRP; k++) {
                if (!lv[k]) continue;
                const int r = (g0 + k) * RPI + r0;
                uint4 v = *reinterpret_cast<const uint4*>(stage + r * EP_LD + ch * 8);
                bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + pixv[k] * p.ldC + ncol;
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
                *reinterpret_cast<uint4*>(o) = v;
            }
        }
#ifdef P3_TIMING
        { const unsigned long long t = __builtin_readcyclecounter(); te[3] += t - te[4]; te[5] = t; }
#endif
    }
#ifdef P3_TIMING
    const unsigned long long T3 = __builtin_readcyclecounter();
#endif
    if constexpr (EPI == EPI_STATS) {
        // every lane parks its 8 partial sums in LDS, one thread per column folds the 4 row groups x WM waves
        // (a shuffle tree here is 16 dependent ds_bpermute round trips, ~2 k cycles)
        __syncthreads();
        float* part = reinterpret_cast<float*>(p3_lds);          // [wave][4][2][WTN]
        float* mine = part + ((wave * 4 + rg) * 2) * WTN + cq * 4;
        *reinterpret_cast<float4*>(mine) = make_float4(ssum[0], ssum[1], ssum[2], ssum[3]);
        *reinterpret_cast<float4*>(mine + WTN) = make_float4(ssq[0], ssq[1], ssq[2], ssq[3]);
        __syncthreads();
        if (tid < BN && n0 + tid < p.Nout) {
            const int wn_c = tid / WTN, cc = tid % WTN;
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < WM; w++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float* src = part + (((w * WN + wn_c) * 4 + r) * 2) * WTN + cc;
                    sm += src[0];
                    sq += src[WTN];
                }
            float* st = p.stats + (int64_t)mb * 2 * p.Nout;
            st[n0 + tid] = sm;
            st[p.Nout + n0 + tid] = sq;
        }
    }
#ifdef P3_TIMING
    if (p.bias && tid == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias)) + (size_t)blockIdx.x * 4;
        dbg[0] = T0; dbg[1] = T1; dbg[2] = T2; dbg[3] = __builtin_readcyclecounter();
        unsigned long long* d2 = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias)) + (size_t)(4 * 35000) + (size_t)blockIdx.x * 4;
        d2[0] = te[0] - T2; d2[1] = te[1]; d2[2] = te[2]; d2[3] = te[3];          // barrier | stage writes | statistics | stores (wave 0)
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------- host side
static unsigned p3_lds_bytes(int P, int BN) { return 2u * P * 1024u + 3u * BN * 64u + 128u + (unsigned)ry_cdiv(P, 4) * 1024u + P3_BM * 4u; }

static bool p3_geometry_bn(const ConvGemmParams& p, P3Geom& g, int bn_force)
{
    g = P3Geom{};
    const TapClass& tc = p.cls[0];
    if (p.nclasses != 1 || tc.ntaps != 9 || p.sh != 1 || p.sw != 1 || p.IH != p.OH || p.IW != p.OW) return false;
    if (p.pool_idx || p.s2d_cin) return false;                    // epilogue variants of the generic kernel only
    if (p.oh_mul != 1 || p.ow_mul != 1 || p.OHf != p.OH || p.OWf != p.OW || tc.oh_add || tc.ow_add) return false;
    if (p.Cin % 32 || p.Nout < 64 || p.Nout % 8 || p.ldA % 8 || p.ldC % 8 || !p.zeros) return false;
    if (p.epi != EPI_RAW && p.epi != EPI_STATS && p.epi != EPI_AFFINE_ACT && p.epi != EPI_ACCUM) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; t++) {
        if (tc.dh[t] < -1 || tc.dh[t] > 1 || tc.dw[t] < -1 || tc.dw[t] > 1 || tc.widx[t] < 0 || tc.widx[t] >= p.wtaps) return false;
        seen |= 1u << ((tc.dh[t] + 1) * 3 + tc.dw[t] + 1);
    }
    if (seen != 0x1ffu) return false;
    const int H = p.OH, W = p.OW;
    const int64_t M = (int64_t)p.NB * H * W;
    if (M * (int64_t)p.ldA >= (1ll << 34) || (int64_t)p.Nout * p.wtaps * p.Cin >= (1ll << 31)) return false;   // 32-bit DMA source offsets
    g.BN = (p.Nout <= 64 || bn_force == 64) ? 64 : 128;
    g.gn = (int)ry_cdiv(p.Nout, g.BN);
    const unsigned budget = 80u * 1024u;                          // two workgroups per CU
    // flat runs: no tile waste; the patch carries one image row of halo on both sides
    const int Rrun = P3_BM + 2 * W + 2;
    const int Prun = (int)ry_cdiv(Rrun + 1, 16);                   // 16 P > R: the last row of a patch buffer is the kernel's zero row
    int best_th = 0, best_tw = 0;
    for (int tw = 1; tw <= W && tw <= P3_BM; tw++) {
        if (W % tw) continue;
        for (int th = 1; th <= H && th * tw <= P3_BM; th++) {
            if (H % th) continue;
            const int area = th * tw, barea = best_th * best_tw;
            if (area > barea || (area == barea && (th + 2) * (tw + 2) < (best_th + 2) * (best_tw + 2))) { best_th = th; best_tw = tw; }
        }
    }
    const int R2 = (best_th + 2) * (best_tw + 2);
    const int P2 = (int)ry_cdiv(R2 + 1, 16);
    const bool ok2 = best_th * best_tw >= 224 && P2 <= 36 && p3_lds_bytes(P2, g.BN) <= budget;
    const bool okr = Prun <= 36 && p3_lds_bytes(Prun, g.BN) <= budget;
    if (!ok2 && !okr) return false;
    // cost per useful output pixel ~ (patch rows / 9 + weight rows) / live pixels; flat runs have no dead rows
    const double c2 = ok2 ? (R2 / 9.0 + g.BN) / (best_th * best_tw) : 1e30;
    const double cr = okr ? (Rrun / 9.0 + g.BN) / (double)P3_BM : 1e30;
    if (c2 < cr) {
        g.mode = 1;
        g.TH = best_th; g.TW = best_tw; g.PW = best_tw + 2; g.R = R2; g.P = P2;
        g.tilesW = W / best_tw;
        g.tilesPerImg = (H / best_th) * g.tilesW;
        g.gm = (int64_t)p.NB * g.tilesPerImg;
    } else {
        g.mode = 2;
        g.PW = W; g.R = Rrun; g.P = Prun;
        g.gm = ry_cdiv(M, P3_BM);
    }
    g.TP = (int)ry_cdiv(g.P, 4);
    if (g.TP > 7) { g.mode = 0; return false; }                   // the kernel reads a chunk's first fragments one barrier early: its last piece is requested by tap step 6
    for (int t = 0; t < 9; t++) { g.tdh[t] = tc.dh[t]; g.tdw[t] = tc.dw[t]; g.twi[t] = tc.widx[t]; }
    g.rPW = 1.0f / (float)g.PW;
    g.rTW = g.TW ? 1.0f / (float)g.TW : 0.f;
    g.lds_bytes = p3_lds_bytes(g.P, g.BN);
    if (g.gm * g.gn > 0x7fffffff || g.gm <= 0) { g.mode = 0; return false; }
    // Small grids.  Through r05 a launch with fewer than 512 workgroups (2 per CU) stayed on the generic kernel's 128-pixel tiles ("8 x 100^2 x
    // 128 -> 128 is 0.7x on this kernel": an r02 measurement of the r02 kernel).  Re-measured in r06 (tools/bench_conv.py, 8 images, same box,
    // statistics epilogue; generic [deep ring where it applies] vs this kernel): 100^2 128 -> 128 50.6 vs 39.0 us, 50^2 256 -> 256 55.8 vs 44.7,
    // 50^2 128 -> 128 26.2 vs 25.3, 25^2 256 -> 256 44.1 vs 39.9, 25^2 512 -> 512 83.3 vs 74.7, 25^2 1024 -> 512 157.0 vs 139.8, 32^2 512 -> 512
    // 86.9 vs 77.4 — it wins on every one-round grid of the 8-image step and of the batch-8 1024^2 inference tape, and by much more on 64-column
    // tiles (p3_geometry below: 26.6 / 56.6 / 39.3 / 18.6 / 60.7 us for the 25^2 256, 25^2 512, 50^2 256, 50^2 128, 32^2 512 cases); batch 1:
    // 25^2 256 -> 256 45.2 -> 27.4 us, 50^2 128 -> 128 27.2 -> 19.1, 25^2 512 -> 512 81.4 -> 46.6.  RYOLO_P3_MIN_WGS: A/B knob (512 ~ the r05
    // rule).  0x400 forces the kernel (tests).
    static const int min_wgs = getenv("RYOLO_P3_MIN_WGS") ? atoi(getenv("RYOLO_P3_MIN_WGS")) : 4;
    if (g.gm * ry_cdiv(p.Nout, 64) < min_wgs && !(p.pipe & 0x400)) { g.mode = 0; return false; }      // (counted in 64-column tiles: what a small grid runs)
    return true;
}

bool p3_geometry(const ConvGemmParams& p, P3Geom& g)
{
    if (!p3_geometry_bn(p, g, 0)) return false;
    // fewer 128-column workgroups than CUs: 64-column tiles double the grid (RYOLO_P3_SMALL_BN64 = the grid size below which it applies; 0: off)
    static const int small64 = getenv("RYOLO_P3_SMALL_BN64") ? atoi(getenv("RYOLO_P3_SMALL_BN64")) : 256;
    // (0x400 = the tests' "force this kernel" bit keeps the 128-column tile; 0x2000 forces the 64-column one)
    if (g.BN == 128 && ((g.gm * g.gn < small64 && !(p.pipe & 0x400)) || (p.pipe & 0x2000))) {
        P3Geom g64;
        if (p3_geometry_bn(p, g64, 64)) g = g64;
    }
    return true;
}

template <int BN, int WM, int WN, int EPI> static int p3_launch_t(const ConvGemmParams& p, const P3Geom& g, hipStream_t stream)
{
    static RyLdsAttr attr;
    if (ry_max_dynamic_lds(attr, reinterpret_cast<const void*>(&conv3x3_patch_kernel<BN, WM, WN, EPI>), 160 * 1024)) return RY_ERR_LAUNCH;
    static const unsigned ldspad = getenv("RYOLO_P3_LDSPAD") ? (unsigned)atoi(getenv("RYOLO_P3_LDSPAD")) : 0u;   // occupancy experiments
    hipLaunchKernelGGL((conv3x3_patch_kernel<BN, WM, WN, EPI>), dim3((unsigned)(g.gm * g.gn)), dim3(256), g.lds_bytes + ldspad, stream, p, g);
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}

template <int BN, int WM, int WN> static int p3_launch_e(const ConvGemmParams& p, const P3Geom& g, hipStream_t stream)
{
    switch (p.epi) {
        case EPI_RAW: return p3_launch_t<BN, WM, WN, EPI_RAW>(p, g, stream);
        case EPI_ACCUM: return p3_launch_t<BN, WM, WN, EPI_ACCUM>(p, g, stream);
        case EPI_STATS: return p3_launch_t<BN, WM, WN, EPI_STATS>(p, g, stream);
        case EPI_AFFINE_ACT: return p3_launch_t<BN, WM, WN, EPI_AFFINE_ACT>(p, g, stream);
    }
    return RY_ERR_ARG;
}

int p3_launch(const ConvGemmParams& p, const P3Geom& g, hipStream_t stream)
{
    if (g.BN == 64) return p3_launch_e<64, 4, 1>(p, g, stream);
    return p3_launch_e<128, 2, 2>(p, g, stream);
}

// =====================================================================================================================
// 3x3 stride-1 WEIGHT GRADIENT: dW[co][tap][ci] = sum_p dY[p][co] * X[p + tap][ci]
//
// The generic split-K kernel (conv.hip) gives every (tap, 32-channel chunk) column tile its own workgroup, so the dY rows
// are re-read 9*Cin/128 times and the X rows 9 times, from HBM / Infinity Cache (the sibling tiles drift apart and miss in
// L2): the kernel runs at the speed of its loads (~6.5 TB/s; ablation in DESIGN.md §4.2).  Here one workgroup owns
// 128 output channels x 32 input channels x ALL 9 taps for a range of pixels:
//   * K (pixels) runs over PADDED coordinates — each image framed by one zero pixel — so tap (dh, dw) is the constant row
//     offset dh*(W+2)+dw into ONE ring of input rows; padding rows are DMA'd from a zero page: no masks, no per-tap loads;
//   * per 32-pixel K step a workgroup brings 32 new ring rows (64 B each) and the 32 x 128 dY tile: 10 KiB for 72 MFMAs
//     (generic: 16 KiB for 32), X crosses L2 once per 128 output channels instead of 9 times;
//   * both operands sit in LDS as contiguous 64-byte pixel rows: lane-linear LDS-DMA targets AND conflict-free for the
//     transposed fragment reads (ds_read_b64_tr_b16: a 32-lane group reads 4 rows x 64 B = one full bank sweep);
//   * wave w owns output channels [32w, 32w+32): 9 accumulator tiles (one per tap), 18 MFMAs per 20 fragments per step;
//   * 3-stage dY ring + sliding X ring, counted vmcnt, one barrier per step; split-K slabs + the deterministic reduce of
//     conv.hip; XCD-aware order keeps the tiles that share a pixel range on one L2.
#define W3_NS 3
#ifndef W3_PF
#define W3_PF 6                                                  // fragments read ahead of the MFMA stream in conv3x3_wgrad64_kernel (A/B: -DW3_PF=4)
#endif

// CO64: layers with <= 64 output channels.  Two waves cover the channels (32 each) and the wave PAIRS split every K step into
// its two 16-pixel halves, each pair accumulating into its own split-K slab (the deterministic reduce adds them), so all four
// waves stay busy; the dY stage shrinks to 64 channels and the ring may be 1024 rows (maps up to 430 pixels wide).
template <bool CO64>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const WgradParams p, const W3Geom g)
{
    constexpr int DYS = CO64 ? 4096 : 8192;                       // bytes of one dY stage: [quarters][32 px][64 B]
    constexpr int NDY = CO64 ? 1 : 2;                             // dY pieces per wave per step
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cow = CO64 ? (wave & 1) : wave;                     // 32-channel quarter of this wave
    const int kh = CO64 ? (wave >> 1) : 0;                        // CO64: the 16-pixel half of each K step this wave multiplies
#ifdef W3_TIMING
    const unsigned long long T0 = __builtin_readcyclecounter();
#endif
    const int t_id = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = t_id % g.gx, bc = (t_id / g.gx) % g.gc, bz = t_id / (g.gx * g.gc);
    const int i0 = bx * 128, ci0 = bc * 32;
    const int64_t kbeg = (int64_t)bz * g.kchunk;
    const int64_t kend = min(g.Mp, kbeg + g.kchunk);
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg + 31) >> 5);
    const int H = p.OH, W = p.OW, PWp = g.PWp, HPp = g.HPp;
    const int HALO = PWp + 1;
    const unsigned rmask = (unsigned)g.RX - 1u;
    unsigned char* const dyst = p3_lds;                            // [W3_NS][quarters][32 px][64 B]
    unsigned char* const xring = p3_lds + W3_NS * DYS;              // [RX][64 B]

    // ---- DMA bookkeeping.  A 1-KiB piece = 16 pixel rows x 64 B; lane -> (row = lane >> 2, 16-byte slot = lane & 3).
    // dY: wave w stages the two 16-row halves of ITS OWN 32-channel quarter; X: waves 0/1 stage the two halves of the 32 new rows.
    // Every stream carries its padded coordinates (q, img, ihp, iwp) in 32-bit registers and advances them with carries; the source
    // address is rebuilt from them with two 32-bit mads and one 64-bit mad (the first version walked 64-bit indices through
    // while-loops and cost ~1600 cycles of DMA issue per K step — cycle counters).
    const int prow_l = lane >> 2, slot = lane & 3;
    const int kend32 = (int)kend, Mp32 = (int)g.Mp;               // Mp < 2^31 (host check)
    // Branch-free: a stream keeps only the padded index q of this lane's row; (image, padded row, padded column) come from two exact
    // multiply-high divisions by the launch constants (W3Geom.m_img / m_row) — the first version carried (img, ihp, iwp) through
    // while-loops, which compile to divergent branch chains: ~200 of the ~430 instructions of a K step were DMA address generation.
    const unsigned per = (unsigned)(HPp * PWp);
    auto locate = [&](int q, int limit, bool& ok) -> int {            // pixel index of padded position q (valid iff ok)
        const unsigned uq = (unsigned)q;
        const unsigned img = __umulhi(uq, g.m_img) >> g.s_img;
        const unsigned rem = uq - img * per;
        const unsigned ihp = __umulhi(rem, g.m_row) >> g.s_row;
        const unsigned iwp = rem - ihp * (unsigned)PWp;
        ok = uq < (unsigned)limit && (ihp - 1u) < (unsigned)H && (iwp - 1u) < (unsigned)W;     // (negative q: uq >= 2^31 > limit)
        return (int)((img * (unsigned)H + ihp - 1u) * (unsigned)W + iwp - 1u);                // < 2^31 when ok (host check)
    };
    // dY rows of step s, half u: padded pixel kbeg + 32 s + 16 u + prow_l
    int dq[NDY];
#pragma unroll
    for (int u = 0; u < NDY; u++) dq[u] = (int)kbeg + 16 * (CO64 ? kh : u) + prow_l;
    const bool d_chan_ok = (i0 + 32 * cow + slot * 8) < p.CoutPad;
    const bf16_t* const dy_base = p.dY + i0 + 32 * cow + slot * 8;
    const bf16_t* const x_base = p.X + ci0 + slot * 8;
    auto issue_dy1 = [&](int stage, int u) {
        bool ok;
        const int pix = locate(dq[u], kend32, ok);
        const bf16_t* src = (ok && d_chan_ok) ? dy_base + (int64_t)pix * p.ldY : p.zeros;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dyst + stage * DYS + cow * 2048 + (CO64 ? kh : u) * 1024), 16, 0, 0);
        dq[u] += 32;
    };
    auto issue_dy = [&](int stage) {
#pragma unroll
        for (int u = 0; u < NDY; u++) issue_dy1(stage, u);
    };
    // X ring: slot of padded row q = q mod RX (q may be negative before the first image: two's-complement masking).  Pieces are
    // 16 aligned rows.  Prologue: every wave stages 16 rows per iteration over [x0, xend) (a multiple of 64 rows covering the rows
    // of steps 0 and 1 with their halos); steady state: waves 0 and 1 append the next 32 rows each step (for step s + 2).
    const int x0 = (int)(((kbeg - HALO) >> 5) << 5);                 // aligned down to 32 (arithmetic shift: also for negatives)
    const int pro_iters = ((int)kbeg + 64 + HALO - x0 + 63) >> 6;
    int xq = x0 + 16 * wave + prow_l;                                // this lane's row in its current piece
    auto issue_x = [&]() {
        bool ok;
        const int pix = locate(xq, Mp32, ok);
        const bf16_t* src = ok ? x_base + (int64_t)pix * p.ldX : p.zeros;
        const unsigned row0 = (unsigned)(xq - prow_l) & rmask;        // wave-uniform, 16-aligned
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(xring + row0 * 64), 16, 0, 0);
    };
    auto step_x = [&](int by) { xq += by; };
    for (int it = 0; it < pro_iters; it++) {
        issue_x();
        step_x(64);
    }
    // now xq = x0 + 64 * pro_iters + 16 * wave + prow_l: exactly where waves 0 and 1 continue
    issue_dy(0);
    if (nk > 1) issue_dy(1);

    // ---- fragment addressing: transposed reads, lane -> (pixel row, channel) inside a 16-lane group (see conv.hip)
    const int s16 = lane & 15, grp = lane >> 4;
    const int fr_row = (grp >> 1) * 8 + (s16 >> 2);                  // + ks*16 (+4 for the second half)
    const int fr_col = (16 * (grp & 1) + 4 * (s16 & 3)) * 2;         // byte offset inside the 64-byte row
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

#ifdef W3_TIMING
    const unsigned long long T1 = __builtin_readcyclecounter();
#endif
    const unsigned kb32 = (unsigned)(int)kbeg;                        // low bits are all the ring index needs
    for (int s = 0; s < nk; s++) {
        // DMA this wave issued during step s-1 (operands of step s+1) may stay in flight: 2 dY pieces (+1 ring piece on waves 0, 1);
        // step 0 follows the prologue, whose last two instructions are dY(1)
#if defined(W3_NO_XDMA) || defined(W3_NO_YDMA)
        wait_vm<0>();                                                 // ablation builds: counts differ, drain
#else
        if (s + 1 >= nk) wait_vm<0>();
        else if (s == 0 || wave >= 2) wait_vm<NDY>();
        else wait_vm<NDY + 1>();
#endif
        __builtin_amdgcn_s_barrier();                                 // step s operands visible; step s-1 fully consumed
        const bool do_dma = s + 2 < nk;
        const int nstage = (s + 2) % W3_NS;
#if !defined(W3_DMA_SHADOW) && !defined(W3_NO_DMA)
        if (do_dma) {
#ifndef W3_NO_XDMA
            if (wave < 2) { issue_x(); step_x(32); }
#endif
#ifndef W3_NO_YDMA
            issue_dy(nstage);
#endif
        }
#endif
        const unsigned char* da = dyst + (s % W3_NS) * DYS + cow * 2048;
        const unsigned q0 = kb32 + 32u * (unsigned)s;                 // padded index (mod 2^32) of the step's first pixel
        // 18 (16-pixel half, tap) MFMAs per step; the B fragment of MFMA i+PF is read while MFMA i runs (a software pipeline PF
        // fragments deep: all-reads-first exposes the LDS latency once per half and needs 36 live registers, the compiler's own
        // schedule double-buffers by one fragment and waits on every MFMA)
        // (transposed reads through inline asm + counted lgkmcnt: the builtin makes hipcc drain vmcnt(0) — this step's freshly issued
        // DMA for step s + 2 — before the first read of every step, see conv_internal.h)
        const unsigned da_a = lds_addr(da) + (unsigned)(fr_row * 64 + fr_col);
        const unsigned xr_a = lds_addr(xring) + (unsigned)fr_col;
        auto read_a = [&](int ks) { return lds_tr16x2(da_a + (unsigned)(ks * 16 * 64), 256u); };
        auto read_b = [&](int i) {
            const int ks = i / 9, t = i % 9;
            const unsigned qq = q0 + (unsigned)(g.toff[t] + ks * 16 + fr_row);
            const ry_s16x4 lo = lds_tr16(xr_a + (qq & rmask) * 64u), hi = lds_tr16(xr_a + ((qq + 4u) & rmask) * 64u);
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        constexpr int PF = 4;
        if constexpr (!CO64) {
            // issue order: a0, b0..b3, a1, then b(i+4) in front of MFMA i.  Reads (two per fragment) issued AFTER the fragment MFMA i
            // needs: i < 4: the rest of b0..b3, a1 and b4..b(i+4) = 10; 4 <= i < 14: four fragments = 8; then 6, 4, 2, 0.
            bf16x8 af[2], bq[18];
            af[0] = read_a(0);
#pragma unroll
            for (int i = 0; i < PF; i++) bq[i] = read_b(i);
            af[1] = read_a(1);
#pragma unroll
            for (int i = 0; i < 18; i++) {
                __builtin_amdgcn_sched_barrier(0);
                if (i + PF < 18) bq[i + PF] = read_b(i + PF);
                if (i < 4) lds_wait2<10>(af[0], bq[i]);
                else if (i < 14) lds_wait2<8>(af[1], bq[i]);
                else if (i == 14) lds_wait<6>(bq[i]);
                else if (i == 15) lds_wait<4>(bq[i]);
                else if (i == 16) lds_wait<2>(bq[i]);
                else lds_wait<0>(bq[i]);
                __builtin_amdgcn_sched_barrier(0);
                acc[i % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i / 9], bq[i], acc[i % 9], 0, 0, 0);
                // -DW3_DMA_SHADOW: the step's DMA instructions between the MFMAs instead of at the top of the step — measured
                // SLOWER here (470 vs 570 TF/s at 128->128 @100^2): their address VALU starves the in-order MFMA issue of a 2-wave SIMD
                if (i == 3 || i == 8 || i == 13) {
                    __builtin_amdgcn_sched_barrier(0);
#if !defined(W3_NO_DMA) && defined(W3_DMA_SHADOW)
                    if (do_dma) {
                        if (i == 3) { if (wave < 2) { issue_x(); step_x(32); } }
                        else issue_dy1(nstage, i == 8 ? 0 : 1);
                    }
#endif
                }
            }
        } else {
            // (<= 64 output channels: 9 MFMAs per step and wave — here the compiler's own just-in-time lgkmcnt schedule around the
            // builtin reads measured FASTER than the counted asm reads, 1.80 vs 2.02 ms at 64->64 @400^2, so this variant keeps it)
            typedef __attribute__((address_space(3))) ry_s16x4 lds_s16x4;
            auto read_a_b = [&](int ks) {
                const unsigned char* a = da + (ks * 16 + fr_row) * 64 + fr_col;
                const ry_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
                const ry_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 256));
                return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            };
            auto read_b_b = [&](int i) {
                const int ks = i / 9, t = i % 9;
                const unsigned qq = q0 + (unsigned)(g.toff[t] + ks * 16 + fr_row);
                const ry_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xring + (qq & rmask) * 64 + fr_col));
                const ry_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xring + ((qq + 4u) & rmask) * 64 + fr_col));
                return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            };
            bf16x8 bq[9];
            const bf16x8 af = read_a_b(kh);
#pragma unroll
            for (int i = 0; i < PF; i++) bq[i] = read_b_b(kh * 9 + i);
#pragma unroll
            for (int i = 0; i < 9; i++) {
                __builtin_amdgcn_sched_barrier(0);
                if (i + PF < 9) bq[i + PF] = read_b_b(kh * 9 + i + PF);
                __builtin_amdgcn_sched_barrier(0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[i], acc[i], 0, 0, 0);
                if (i == 2 || i == 5) {
                    __builtin_amdgcn_sched_barrier(0);
#if !defined(W3_NO_DMA) && defined(W3_DMA_SHADOW)
                    if (do_dma) {
                        if (i == 2) { if (wave < 2) { issue_x(); step_x(32); } }
                        else issue_dy1(nstage, 0);
                    }
#endif
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

#ifdef W3_TIMING
    const unsigned long long T2 = __builtin_readcyclecounter();
#endif
    // ---- split-K partial tile -> workspace [z][Cout][9*Cin] (GEMM layout; 128-byte row segments per store)
    const int NK = 9 * p.Cin;
    float* part = p.partial + ((int64_t)bz * (CO64 ? 2 : 1) + kh) * p.Cout * NK;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int kc = t * p.Cin + ci0 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int co = i0 + 32 * cow + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (co < p.Cout) part[(int64_t)co * NK + kc] = acc[t][e];
        }
    }
#ifdef W3_TIMING
    if (tid == 0) {   // debug build only: timestamps into the tail of the slab workspace (tools/bench_wgrad.py reads them)
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.partial + (size_t)g.slabs * p.Cout * NK) + (size_t)blockIdx.x * 4;
        dbg[0] = T1 - T0; dbg[1] = T2 - T1; dbg[2] = __builtin_readcyclecounter() - T2; dbg[3] = nk;
    }
#endif
}

// ---- 64-pixel K steps -------------------------------------------------------------------------------------------------------------------
// The first ring kernels take 32 padded pixels per step.  For <= 64 output channels the wave PAIRS of conv3x3_wgrad_kernel<true> split such a
// step: 9 MFMAs per wave between two barriers — 64->64 @400^2 ran at 415 TF/s (1.82 ms where its bytes allow 0.5): the step's fixed cost
// (barrier, DMA wait, fragment addressing, DMA issue) was larger than its matrix work.  Here a step covers 64 pixels:
//   CO64 (<= 64 output channels): wave w multiplies channel quarter w & 1 with pixel half w >> 1 (32 pixels): 18 MFMAs per wave and step;
//   128-channel tiles: wave w owns channel quarter w and all 64 pixels: 36 MFMAs per wave and step (two passes of the 18-MFMA schedule).
// DMA pieces per wave and step: its own dY rows (2 / 4 pieces) + ONE of the four 16-row pieces of the 64 new ring rows — balanced over
// the waves (the 32-pixel kernels give the ring to waves 0 and 1).  Both streams run ONE step ahead (two dY stages, a step is long enough
// to cover the HBM latency): CO64 2 x 8 KiB + the 1024-row ring = 80 KiB; 128 channels 2 x 16 KiB + a 512-row ring = 64 KiB: two workgroups
// per CU.  Ring rows live = 2 halos + this step + the next + alignment slack (<= 1015 for a 400-pixel map).
// Measured (kernel + reduce, batch 64): 64->64 @400^2 1822 -> 1092 us, @200^2 503 -> 317; the training step 746 -> 777 img/s.
// Round 4: the loop above spent ~400 VALU instructions per step on fragment addresses ((row & ring mask) per read, twice per fragment) and on
// copying read results into operand quads — for 36 MFMAs; it ran at 49 % of the matrix rate, bound by VALU issue (cycle stamps: 4725 per step).
// Now the taps are walked ROW BY ROW: the three taps of a kernel row read ring rows r - 1, r, r + 1, and the second transposed read of a fragment
// sits 4 rows further, so ONE wrapped address per (kernel row, 16-pixel half) serves six reads through ds_read's immediate offset.  An
// immediate cannot wrap: the ring is followed by a copy of its first 16 rows (the piece that lands on ring row 0 is requested twice) when the
// LDS budget has the 1 KiB (g.mirror); otherwise every read wraps its own address (2 VALU instructions instead of 3 + 2 copies).
template <bool CO64, bool MIRROR>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad64_kernel(const WgradParams p, const W3Geom g)
{
    constexpr int DYS = CO64 ? 8192 : 16384;                       // one dY stage: [2 | 4 quarters][64 px][64 B]
    constexpr int NDY = CO64 ? 2 : 4;                              // dY pieces per wave and step
    constexpr int NH = CO64 ? 1 : 2;                               // 32-pixel halves a wave multiplies per step
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cow = CO64 ? (wave & 1) : wave, kh = CO64 ? (wave >> 1) : 0;
#ifdef W3_TIMING
    const unsigned long long T0 = __builtin_readcyclecounter();
#endif
    const int t_id = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = t_id % g.gx, bc = (t_id / g.gx) % g.gc, bz = t_id / (g.gx * g.gc);
    const int i0 = bx * 128, ci0 = bc * 32;
    const int64_t kbeg = (int64_t)bz * g.kchunk;
    const int64_t kend = min(g.Mp, kbeg + g.kchunk);
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg + 63) >> 6);
    const int H = p.OH, W = p.OW, PWp = g.PWp, HPp = g.HPp;
    const int HALO = PWp + 1;
    const unsigned rmask = (unsigned)g.RX - 1u;
    // LDS: [ring RX x 64 B][mirror 1 KiB if MIRROR][dY stage 0][dY stage 1] — the ring first, so that a wrapped byte offset IS the address
    unsigned char* const xring = p3_lds;
    unsigned char* const dyst = p3_lds + (unsigned)g.RX * 64u + (MIRROR ? 1024u : 0u);
    const int prow_l = lane >> 2, slot = lane & 3;
    const int kend32 = (int)kend, Mp32 = (int)g.Mp;
    const unsigned per = (unsigned)(HPp * PWp);
    auto locate = [&](int q, int limit, bool& ok) -> int {
        const unsigned uq = (unsigned)q;
        const unsigned img = __umulhi(uq, g.m_img) >> g.s_img;
        const unsigned rem = uq - img * per;
        const unsigned ihp = __umulhi(rem, g.m_row) >> g.s_row;
        const unsigned iwp = rem - ihp * (unsigned)PWp;
        ok = uq < (unsigned)limit && (ihp - 1u) < (unsigned)H && (iwp - 1u) < (unsigned)W;
        return (int)((img * (unsigned)H + ihp - 1u) * (unsigned)W + iwp - 1u);
    };
    // DMA requests of a wave per step: ONE 16-row piece of the 64 new ring rows (lane -> row lane >> 2, 16-byte slot lane & 3) and the dY rows
    // of ITS quarter that it reads itself, RPW = 64 (CO64: 32) rows x 32 channels.  The dY stage is SLOT-MAJOR — [4 slots of 8 channels][RPW
    // rows][16 B] — so that a lane owns one pixel row (lane % RPW) and instruction u moves slot u (CO64: 2 u + lane / 32) of all rows: ONE
    // padded-pixel decomposition per lane and step serves all NDY instructions.  (Row-major stages took one per instruction: with the ring
    // piece five `locate`s per step — two mulhi, three mullo, a 64-bit mad each, quarter-rate — and the step ran 3590 cycles with them, 2395
    // without.)  The pointers of the NEXT request are formed inside the MFMA stream (prep_*), the block behind the barrier is NDY + 1 DMAs.
    constexpr int RPW = CO64 ? 32 : 64, SPI = 64 / RPW;              // rows per wave, slots per instruction
    const int drow = lane % RPW, dsub = lane / RPW;
    const bf16_t* const dy_base = p.dY + i0 + 32 * cow;
    const bf16_t* const x_base = p.X + ci0 + slot * 8;
    const int x0 = (int)(((kbeg - HALO) >> 6) << 6);                 // aligned down to 64 (arithmetic shift: also for negatives)
    const int pro_iters = ((int)kbeg + 64 + HALO + 16 - x0 + 63) >> 6;
    int xq = x0 + 16 * wave + prow_l;                                 // this lane's ring row of the next request
    int dq = (int)kbeg + 32 * kh + drow;                              // this lane's dY pixel of the next request
    const bf16_t *xsrc, *dsrc;                                        // nullptr: padding / out of range -> the zero page
    auto prep_x = [&]() {
        bool ok;
        const int pix = locate(xq, Mp32, ok);
        xsrc = ok ? x_base + (int64_t)pix * p.ldX : nullptr;
    };
    auto prep_dy = [&]() {
        bool ok;
        const int pix = locate(dq, kend32, ok);
        dsrc = ok ? dy_base + (int64_t)pix * p.ldY : nullptr;
    };
    auto issue_x = [&]() {
        const bf16_t* src = xsrc ? xsrc : p.zeros;
        const unsigned row0 = (unsigned)__builtin_amdgcn_readfirstlane(xq - prow_l) & rmask;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(xring + row0 * 64), 16, 0, 0);
        if (MIRROR && row0 == 0u)                                    // wave-uniform: once per lap of the ring
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(xring + (unsigned)g.RX * 64u), 16, 0, 0);
        xq += 64;
    };
    auto issue_dy1 = [&](int stage, int u) {
        const int sl = u * SPI + dsub;                                // 8-channel slot this lane moves
        const bf16_t* src = (dsrc && (i0 + 32 * cow + sl * 8) < p.CoutPad) ? dsrc + sl * 8 : p.zeros;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dyst + stage * DYS + cow * 4096 + kh * 2048 + u * 1024), 16, 0, 0);
        if (u == NDY - 1) dq += 64;
    };
    auto issue_dy = [&](int stage) {
#pragma unroll
        for (int u = 0; u < NDY; u++) issue_dy1(stage, u);
    };
    for (int it = 0; it < pro_iters; it++) { prep_x(); issue_x(); }   // rows of step 0 with both halos
    prep_dy();
    issue_dy(0);
    prep_x();                                                         // the request of step 0 (operands of step 1)
    prep_dy();

    const int s16 = lane & 15, grp = lane >> 4;
    const int fr_row = (grp >> 1) * 8 + (s16 >> 2);
    const int fr_col = (16 * (grp & 1) + 4 * (s16 & 3)) * 2;
    f32x16 acc[9];                                                   // acc[3 * (dh + 1) + (dw + 1)]
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
    const unsigned kb32 = (unsigned)(int)kbeg;
    const unsigned xr_a = lds_addr(xring) + (unsigned)fr_col;
    const unsigned lane_b = (unsigned)(fr_row * 64);                  // this lane's row inside a fragment, in ring bytes
    const unsigned bmask = ((unsigned)g.RX << 6) - 1u;
#ifdef W3_TIMING
    const unsigned long long T1 = __builtin_readcyclecounter();
#endif
#ifdef W3_TIMING
    unsigned long long t_wait = 0, t_bar = 0;
#endif
    for (int s = 0; s < nk; s++) {
#ifdef W3_TIMING
        const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
        wait_vm<0>();                                                 // everything this wave issued one step ago has landed
#ifdef W3_TIMING
        const unsigned long long tw1 = __builtin_readcyclecounter();
#endif
        __builtin_amdgcn_s_barrier();                                 // ... and everybody else's; step s - 1 fully consumed
#ifdef W3_TIMING
        t_wait += tw1 - tw0;
        t_bar += __builtin_readcyclecounter() - tw1;
#endif
        const bool more = s + 1 < nk;                                 // this step requests the operands of the next one
        // ONE fragment stream per step over its NH 32-pixel halves: fragment I = 18 hh + 9 ks + 3 (dh + 1) + (dw + 1) reads ring row
        // q0(hh) + 16 ks + dh * PWp + dw + fr_row (+ 4 for its second read); A fragment k = 2 hh + ks (16 pixels of dY) is read right in front of
        // B fragment 9 k.  Reads run W3_PF fragments ahead of the MFMAs, across the halves (the round-3 loop restarted its pipeline per half).
        constexpr int NF = 18 * NH, PF = W3_PF;
        const unsigned qs = kb32 + 64u * (unsigned)s + (CO64 ? 32u * (unsigned)kh : 0u);
        // dY fragment (slot-major stage): channel fr_col / 2 = 8 * slot + c, row fr_row (+ 4 for the second read)
        const unsigned da_s = lds_addr(dyst + (s & 1) * DYS + cow * 4096 + kh * 2048) + (unsigned)((fr_col >> 4) * (RPW * 16) + fr_row * 16 + (fr_col & 15));
        ry_s16x4 al[2 * NH], ah[2 * NH], bl[NF], bh[NF];
        unsigned gaddr[NF / 3];                                       // MIRROR: wrapped address of the dw = -1 fragment of a (half, ks, dh) group
        auto read_b = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int hh = I / 18, i = I % 18, ks = i / 9, dhi = (i % 9) / 3, dwi = i % 3;
            if constexpr (i % 9 == 0) {                               // A fragment 2 hh + ks first
                constexpr unsigned ao = (unsigned)((32 * hh + 16 * ks) * 16);
                al[I / 9] = lds_tr16_off<ao>(da_s);
                ah[I / 9] = lds_tr16_off<ao + 64>(da_s);
            }
            if constexpr (MIRROR) {
                if constexpr (dwi == 0) {
                    const unsigned sb = (qs + (unsigned)(32 * hh + 16 * ks + (dhi - 1) * PWp - 1)) << 6;     // scalar
                    gaddr[I / 3] = xr_a + ((lane_b + sb) & bmask);
                }
                bl[I] = lds_tr16_off<dwi * 64>(gaddr[I / 3]);
                bh[I] = lds_tr16_off<dwi * 64 + 256>(gaddr[I / 3]);
            } else {
                const unsigned sb = (qs + (unsigned)(32 * hh + 16 * ks + (dhi - 1) * PWp - 1 + dwi)) << 6;      // scalar
                bl[I] = lds_tr16(xr_a + ((lane_b + sb) & bmask));
                bh[I] = lds_tr16(xr_a + ((lane_b + sb + 256u) & bmask));
            }
        };
        P3Unroll<0, PF>::run(read_b);
        bf16x8 af[2 * NH];
        auto mma = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            __builtin_amdgcn_sched_barrier(0);
#if !defined(W3_ABL) || W3_ABL != 2
            if constexpr (I + PF < NF) read_b(std::integral_constant<int, I + PF>{});
#else
            if constexpr (I + PF < NF) { bl[I + PF] = bl[(I + PF) % PF]; bh[I + PF] = bh[(I + PF) % PF]; if constexpr ((I + PF) % 9 == 0) { al[(I + PF) / 9] = al[0]; ah[(I + PF) / 9] = ah[0]; } }
#endif
            // LDS returns in order: "at most N later reads in flight" = fragment I has landed, and with it everything read before it (its A
            // fragment included: tied to the wait only at its first use — tying it again would re-define it and cost two copies per MFMA).
            // N = 2 per B fragment read after I, + 2 if an A fragment was read among them.
            constexpr int ahead = (I + PF < NF ? I + PF : NF - 1);
            constexpr int N = 2 * (ahead - I) + ((ahead / 9 > I / 9) ? 2 : 0);
            static_assert(N <= 15, "lgkmcnt is a 4-bit counter");
#if defined(W3_ABL) && W3_ABL == 2
            constexpr int NW = I < PF ? N : 0;
            if constexpr (I % 9 == 0) { lds_wait_h2<NW>(al[I / 9], ah[I / 9], bl[I], bh[I]); af[I / 9] = join_halves(al[I / 9], ah[I / 9]); }
            else lds_wait_h<NW>(bl[I], bh[I]);
#else
            if constexpr (I % 9 == 0) { lds_wait_h2<N>(al[I / 9], ah[I / 9], bl[I], bh[I]); af[I / 9] = join_halves(al[I / 9], ah[I / 9]); }
            else lds_wait_h<N>(bl[I], bh[I]);
#endif
            __builtin_amdgcn_sched_barrier(0);
#if defined(W3_ABL) && W3_ABL == 3
            if constexpr (I < 9) acc[I % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[I / 9], join_halves(bl[I], bh[I]), acc[I % 9], 0, 0, 0);
            else asm volatile("" :: "v"(bl[I]), "v"(bh[I]));
#else
            acc[I % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[I / 9], join_halves(bl[I], bh[I]), acc[I % 9], 0, 0, 0);
#endif
            // source pointers of the next step's request, right behind an MFMA
            // The step's NDY + 1 DMA instructions follow MFMAs 0, 2, 4, ... (a DMA issue stalls the wave for ~100+ cycles; issued as one block
            // behind the barrier the five of them cost a third of the step — 3590 cycles against 2395 without — behind an MFMA they wait in
            // its shadow), then the pointers of the next request are formed behind MFMAs in the second half of the stream.
            if constexpr (I % 2 == 0 && I / 2 <= NDY) {
                __builtin_amdgcn_sched_barrier(0);
#if defined(W3_ABL) && W3_ABL == 1
                if (false) {
#elif defined(W3_ABL) && W3_ABL == 4
                if (more) { xsrc = nullptr; dsrc = nullptr;                  // every request from the zero page
#elif defined(W3_ABL) && W3_ABL == 5
                if (more && I == 0) {                                        // ring rows only
#elif defined(W3_ABL) && W3_ABL == 6
                if (more && I != 0) {                                        // dY only
#else
                if (more) {
#endif
                    if constexpr (I == 0) issue_x();
                    else issue_dy1((s + 1) & 1, I / 2 - 1);
                }
            }
            if constexpr (I == NF / 2 + 1 || I == NF / 2 + 5) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (I == NF / 2 + 1) prep_x();
                else prep_dy();
            }
        };
        P3Unroll<0, NF>::run(mma);
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef W3_TIMING
    const unsigned long long T2 = __builtin_readcyclecounter();
#endif
    // split-K partial tile -> workspace (CO64: two slabs per K range, one per pixel half, summed by the deterministic reduce)
    const int NK = 9 * p.Cin;
    float* part = p.partial + ((int64_t)bz * (CO64 ? 2 : 1) + kh) * p.Cout * NK;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int kc = g.tap_of[j] * p.Cin + ci0 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int co = i0 + 32 * cow + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (co < p.Cout) part[(int64_t)co * NK + kc] = acc[j][e];
        }
    }
#ifdef W3_TIMING
    if (tid == 0) {   // debug build only: timestamps into the tail of the slab workspace (tools/bench_wgrad.py reads them)
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.partial + (size_t)g.slabs * p.Cout * NK) + (size_t)blockIdx.x * 4;
        dbg[0] = t_wait; dbg[1] = T2 - T1; dbg[2] = t_bar; dbg[3] = nk;              // (wave 0's DMA wait and barrier wait, summed over the steps)
    }
#endif
}

bool w3_geometry(const WgradParams& p, W3Geom& g)
{
    g = W3Geom{};
    if (!p.zeros || p.ntaps != 9 || p.sh != 1 || p.sw != 1 || p.IH != p.OH || p.IW != p.OW) return false;
    if (p.Cin % 32 || p.ldX % 8 || p.ldY % 8 || p.CoutPad % 8 || p.CoutPad < p.Cout || p.CoutPad > p.ldY) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; t++) {
        if (p.dh[t] < -1 || p.dh[t] > 1 || p.dw[t] < -1 || p.dw[t] > 1) return false;
        seen |= 1u << ((p.dh[t] + 1) * 3 + p.dw[t] + 1);
    }
    if (seen != 0x1ffu) return false;
    g.PWp = p.OW + 2;
    g.HPp = p.OH + 2;
    g.Mp = (int64_t)p.NB * g.HPp * g.PWp;
    if (g.Mp >= (1ll << 31)) return false;                        // 32-bit stream coordinates
    for (int t = 0; t < 9; t++) g.toff[t] = p.dh[t] * g.PWp + p.dw[t];
    for (int t = 0; t < 9; t++) g.tap_of[(p.dh[t] + 1) * 3 + p.dw[t] + 1] = t;
    auto magic = [](unsigned d, unsigned& m, unsigned& sh) {         // n / d == mulhi(n, m) >> sh for 0 <= n < 2^31 (d >= 3 here: padded sizes)
        unsigned l = 0;
        while ((1ull << l) < d) l++;
        m = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
        sh = l - 1;
    };
    magic((unsigned)(g.HPp * g.PWp), g.m_img, g.s_img);
    magic((unsigned)g.PWp, g.m_row, g.s_row);
    // round 5: the 8-wave form (conv3x3_wgrad8.hip: 2 x (5 | 4) accumulator blocks per wave, rings of any length) where it applies
    if (w8_geometry(p, g)) { g.ok = 1; return true; }
    g.co64 = p.Cout <= 64 ? 1 : 0;
    // RYOLO_W3_STEP64 (A/B knob): bit 0 = 64-pixel K steps for <= 64 output channels, bit 1 = for the 128-channel tiles (conv3x3_wgrad64_kernel)
    static const int step64 = getenv("RYOLO_W3_STEP64") ? atoi(getenv("RYOLO_W3_STEP64")) : 3;
    bool s64 = (step64 >> (g.co64 ? 0 : 1)) & 1;
    int need = 2 * (g.PWp + 1) + (s64 ? 209 : 160);
    int rx = 256;
    while (rx < need) rx <<= 1;
    if (s64 && rx > (g.co64 ? 1024 : 512)) {                      // the 64-pixel form would not leave two workgroups per CU: 32-pixel steps
        s64 = false;
        need = 2 * (g.PWp + 1) + 160;
        rx = 256;
        while (rx < need) rx <<= 1;
    }
    if (rx > (g.co64 ? 1024 : 512)) return false;                 // <= 80 KiB LDS: two workgroups per CU; wider maps stay on the generic kernel
    g.step64 = s64 ? 1 : 0;
    g.RX = rx;
    g.gx = (int)ry_cdiv(p.Cout, 128);
    g.gc = p.Cin / 32;
    // one workgroup per CU (r04; 512 = two per CU until then): with the BatchNorm passes at 5-8 waves per SIMD on the main stream the side stream
    // does better with fewer, longer workgroups (half the split-K slabs, prologue amortised over twice the steps): same-box step 863 -> 874 img/s
    // at 256, 868 at 128 / 192, 860 at 768 (A/B knob)
    static const int w3_target = getenv("RYOLO_W3_BLOCKS") ? atoi(getenv("RYOLO_W3_BLOCKS")) : 256;
    int64_t sk = ry_cdiv(w3_target, (int64_t)g.gx * g.gc);
    static const int minsteps = getenv("RYOLO_W3_MINSTEPS") ? atoi(getenv("RYOLO_W3_MINSTEPS")) : 24;   // measured 24 / 48 / 128: shorter splits fill the chip, the two-halo prologue still amortises
    const int64_t maxsplit = g.Mp / ((int64_t)minsteps * 32);      // K-steps per split: the ring prologue (2 halos) must amortise
    if (sk > maxsplit) sk = maxsplit;
    // small problems stay on the generic kernel, whose finer tiles fill the chip better (measured: 16 x 100^2 x 32 -> 64: 0.6x here)
    static const bool force = getenv("RYOLO_W3_FORCE") != nullptr;   // A/B runs: ignore the size heuristic
    if (sk < 1) sk = 1;
    if ((int64_t)g.gx * g.gc * sk < 128 && !force) return false;
    const int kstep = g.step64 ? 64 : 32;
    g.kchunk = ry_cdiv(ry_cdiv(g.Mp, sk), kstep) * kstep;
    g.splitk = (int)ry_cdiv(g.Mp, g.kchunk);
    g.lds_bytes = (g.step64 ? 2u * (g.co64 ? 8192u : 16384u) : W3_NS * (g.co64 ? 4096u : 8192u)) + (unsigned)g.RX * 64u;
    static const int w3_mirror = getenv("RYOLO_W3_MIRROR") ? atoi(getenv("RYOLO_W3_MIRROR")) : 1;      // A/B knob
    g.mirror = (g.step64 && w3_mirror && g.lds_bytes + 1024u <= 80u * 1024u) ? 1 : 0;
    if (g.mirror) g.lds_bytes += 1024u;
    g.slabs = g.splitk * (g.co64 ? 2 : 1);
    g.ok = 1;
    return true;
}

int w3_launch(const WgradParams& p, const W3Geom& g, hipStream_t stream)
{
    if (g.v8) return w8_launch(p, g, stream);
    static RyLdsAttr attr_f, attr_t, attr_64f, attr_64t, attr_64fm, attr_64tm;
    if (ry_max_dynamic_lds(attr_f, reinterpret_cast<const void*>(&conv3x3_wgrad_kernel<false>), 160 * 1024) ||
        ry_max_dynamic_lds(attr_t, reinterpret_cast<const void*>(&conv3x3_wgrad_kernel<true>), 160 * 1024) ||
        ry_max_dynamic_lds(attr_64f, reinterpret_cast<const void*>(&conv3x3_wgrad64_kernel<false, false>), 160 * 1024) ||
        ry_max_dynamic_lds(attr_64t, reinterpret_cast<const void*>(&conv3x3_wgrad64_kernel<true, false>), 160 * 1024) ||
        ry_max_dynamic_lds(attr_64fm, reinterpret_cast<const void*>(&conv3x3_wgrad64_kernel<false, true>), 160 * 1024) ||
        ry_max_dynamic_lds(attr_64tm, reinterpret_cast<const void*>(&conv3x3_wgrad64_kernel<true, true>), 160 * 1024))
        return RY_ERR_LAUNCH;
    const dim3 grid((unsigned)((int64_t)g.gx * g.gc * g.splitk));
    if (g.step64 && g.co64 && g.mirror)
        hipLaunchKernelGGL((conv3x3_wgrad64_kernel<true, true>), grid, dim3(256), g.lds_bytes, stream, p, g);
    else if (g.step64 && g.co64)
        hipLaunchKernelGGL((conv3x3_wgrad64_kernel<true, false>), grid, dim3(256), g.lds_bytes, stream, p, g);
    else if (g.step64 && g.mirror)
        hipLaunchKernelGGL((conv3x3_wgrad64_kernel<false, true>), grid, dim3(256), g.lds_bytes, stream, p, g);
    else if (g.step64)
        hipLaunchKernelGGL((conv3x3_wgrad64_kernel<false, false>), grid, dim3(256), g.lds_bytes, stream, p, g);
    else if (g.co64)
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<true>), grid, dim3(256), g.lds_bytes, stream, p, g);
    else
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<false>), grid, dim3(256), g.lds_bytes, stream, p, g);
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}
