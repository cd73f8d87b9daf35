// 3x3 stride-1 convolution with 64 input and <= 64 output channels as a PERSISTENT, WEIGHT-STATIONARY kernel — gfx950 only.
// Reference rows served: SURVEY.md §8a M1/M2 (`Conv`, model/utils.py:6-32): the 64 -> 64 3x3 layers of the first ELAN blocks (400^2, 200^2,
// 100^2 maps at 800^2 input) and their data gradients — 15 launches, 5.6 ms of the batch-64 step on conv3x3_patch_kernel<64, 4, 1>
// (600 TF/s = 23 % of the MFMA peak, 2.7x above their HBM byte floor).
//
// Why the halo-patch kernel is slow here: K = 9 x 64 = 576 is 18 steps; every workgroup re-streams the 72 KiB weight tensor through its
// LDS ring for 256 output pixels, pays a barrier + a counted DMA wait per 8 MFMAs of a wave, and spends 43 % of its life in prologue
// and epilogue.  These layers are HBM-side (2 x M x 64 channels x 2 B against 2 x M x 64 x 576 FLOP: 0.47 ms of bytes for the 400^2 layer,
// 0.30 ms of MFMA), so what matters is a steady stream: loads a full tile ahead, stores that never stop the matrix pipe for long.
//
// This kernel:
//   * ONE 4-wave workgroup per CU for the whole launch (grid = CUs), walking tiles of TH x TW output pixels (10 x 25 on the 100 / 200 /
//     400-wide maps) in an XCD-local order (the workgroups of an XCD work on neighbouring tiles at the same time: halo rows hit in L2);
//   * the WEIGHTS LIVE IN REGISTERS: wave (wm, wn) owns 128 pixels x 32 output channels; its 9 taps x 4 K-steps of MFMA A operands
//     (32 channels x 16 inputs each) are 36 x 4 = 144 VGPRs, loaded once per launch (one wave per SIMD, so a wave may use the whole
//     512-entry register file: 144 weights + 64 accumulators + addresses).  No weight traffic, no weight LDS, in the tile loop;
//   * the input patch of a tile ((TH + 2) x (TW + 2) pixels x 64 channels, 41.5 KiB) comes in by LDS-DMA, double buffered, TWO tiles
//     ahead of its use is requested right after the barrier that frees the buffer (one whole tile of compute covers the HBM latency);
//     halo pixels outside the image are DMA'd from the zero page, so the compute loop has no masks at all;
//   * a tile is 144 MFMAs per wave (v_mfma_f32_32x32x16_bf16, operands swapped: the accumulator holds 4 consecutive channels of one pixel
//     per lane) fed by 144 ds_read_b128 whose addresses are per-wave constants (row base per (pixel block, tap), XOR per K-step): the
//     128-byte patch rows are bank-swizzled on the DMA source side (16-byte slot ^= (row >> 1) & 7: conflict-free for 32 consecutive rows);
//   * ONE barrier per tile; the epilogue is wave-local (accumulators -> 10 KiB staging block of the wave -> 16-byte row segments), its
//     stores are fire-and-forget under the next tile's MFMAs;
//   * BatchNorm batch statistics are accumulated in 8 registers per lane over ALL tiles of the workgroup and folded once at the end:
//     one partial row per workgroup (256 rows per launch instead of 41 000: no fold pass in front of bn_finalize).
#include "conv_internal.h"
#include <type_traits>

#define WS3_STG_LD 40                                          // staging row stride in bf16 (32 channels + 8: 80 bytes, 16-byte aligned)
#define WS3_STG_BYTES (128 * WS3_STG_LD * 2)                   // one wave's staging block: 128 pixel rows
#define WS3_WAVES 4

extern __shared__ __attribute__((aligned(1024))) unsigned char ws3_lds[];

// compile-time loop: hipcc does not unroll a loop whose body holds inline asm (convergent), and a rolled loop would index the register arrays
// dynamically (they would go to scratch memory)
template <int U, int N> struct Ws3Unroll {
    template <class F> static __device__ __forceinline__ void run(F& f)
    {
        f(std::integral_constant<int, U>{});
        Ws3Unroll<U + 1, N>::run(f);
    }
};
template <int N> struct Ws3Unroll<N, N> {
    template <class F> static __device__ __forceinline__ void run(F&) {}
};

template <int N> __device__ __forceinline__ void ws3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS reads next to LDS-DMA in flight go through inline asm with hand-counted lgkmcnt: for a plain load hipcc's waitcnt pass cannot tell the
// staging / patch reads from the DMA's LDS writes and drains vmcnt(0) in front of them — i.e. it would wait for the patch requested two tiles
// ahead right after requesting it (seen in the ISA of the first cut).  LDS operations of a wave return in order; no scalar load may sit
// between a read and its wait (scalar loads share lgkmcnt and return out of order): the regions below touch registers only.
__device__ __forceinline__ bf16x8 ws3_rd128(unsigned addr)
{
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
typedef unsigned ws3_u4 __attribute__((ext_vector_type(4)));
typedef unsigned ws3_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ws3_u4 ws3_rd128u(unsigned addr)
{
    ws3_u4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ ws3_u2 ws3_rd64u(unsigned addr)
{
    ws3_u2 r;
    asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
template <int N> __device__ __forceinline__ void ws3_wait_lds(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
template <int N> __device__ __forceinline__ void ws3_wait_lds(ws3_u4& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
template <int N> __device__ __forceinline__ void ws3_wait_lds(ws3_u2& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_ws64_kernel(const ConvGemmParams p, const Ws3Geom g)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // scalar: uniform branches, scalar LDS bases
    const int wm = wave >> 1, wn = wave & 1;                   // wave = pixel half of the tile (128 pixels) x output-channel half (32)
    const int h = lane >> 5, l31 = lane & 31;
    const int H = p.OH, W = p.OW;

    // ---- my tiles: XCD x owns the contiguous range [x * T8, (x + 1) * T8); its workgroups take them round robin -----------------------------
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int nloc = gridDim.x >> 3;                           // grid is a multiple of 8
    const int T8 = (int)((g.ntiles + 7) >> 3);
    const int tbeg = xcd * T8 + loc;
    const int tend = min((xcd + 1) * T8, (int)g.ntiles);
    const int ntl = tbeg < tend ? (tend - tbeg + nloc - 1) / nloc : 0;

    // ---- weights -> registers (A operands: row = output channel nb * 32 + l31, K = 16 input channels of step ks, 8 per lane half) -----------
    bf16x8 wreg[9][4];
    {
        const int co = wn * 32 + l31;
        const bool ok = co < p.Nout;
        const bf16_t* wrow = p.W + (int64_t)(ok ? co : 0) * p.wtaps * 64 + 8 * h;
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                uint4 v = *reinterpret_cast<const uint4*>(wrow + g.twi[t] * 64 + ks * 16);
                if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                wreg[t][ks] = __builtin_bit_cast(bf16x8, v);
            }
    }

    // ---- patch DMA: piece j = wave + 4 u covers patch rows 8 j ... 8 j + 7; lane -> (row, 16-byte slot); (patch row, column) are recomputed per
    //      tile (two multiplications per piece) rather than kept in 12 registers
    constexpr int MAXU = 12;                                   // <= 48 pieces = 384 patch rows
    // ---- B-operand (pixel) fragment addresses: pixel block b of this wave, tap t -> swizzled LDS byte offset of its patch row (buffer 0) -----
    unsigned aaddr[4][9];
    bool live[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int m = wm * 128 + b * 32 + l31;
        live[b] = m < g.TH * g.TW;
        const int r = small_div(live[b] ? m : 0, g.TW, g.rTW), c = (live[b] ? m : 0) - r * g.TW;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const unsigned row = (unsigned)((r + g.tdh[t] + 1) * g.PW + c + g.tdw[t] + 1);
            aaddr[b][t] = (row << 7) + ((((row >> 1) & 7u) ^ (unsigned)h) << 4);       // K-step ks: ^ (ks << 5)
        }
    }
    // ---- store side: lane -> 16-byte chunk ch of staged row it * 16 + r0 -----------------------------------------------------------------------
    const int ch = lane & 3, r0 = lane >> 2;
    const int ncol = wn * 32 + ch * 8;
    const bool col_ok = ncol < p.Nout;
    unsigned char* const stg = ws3_lds + 2u * g.patch_bytes + (unsigned)wave * WS3_STG_BYTES;
    // statistics: lane -> channel quad cq (4 channels), rows rg + 8 k
    const int cq = lane & 7, rg = lane >> 3;
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};

    auto tile_origin = [&](int tile, int& pix0, int& oh0, int& ow0) {
        const int img = tile / g.tilesPerImg;
        const int rem = tile - img * g.tilesPerImg;
        const int th = rem / g.tilesW;
        oh0 = th * g.TH;
        ow0 = (rem - th * g.tilesW) * g.TW;
        pix0 = (img * H + oh0) * W + ow0;
    };
    auto issue_patch = [&](int tile, unsigned buf_off) {
        int pix0, oh0, ow0;
        tile_origin(tile, pix0, oh0, ow0);
#pragma unroll
        for (int u = 0; u < MAXU; u++) {
            const int j = wave + WS3_WAVES * u;
            if (j < g.NP) {                                    // wave-uniform
                const int row = 8 * j + (lane >> 3);
                const int pr = small_div(row, g.PW, g.rPW), pc = row - pr * g.PW;
                const int ih = oh0 + pr - 1, iw = ow0 + pc - 1;
                // branch-free: every lane forms its address, dead lanes (halo outside the image, rows past the patch) select the zero page
                const bool ok = (row < g.R) & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
                const int64_t pix = (int64_t)pix0 + (pr - 1) * W + (pc - 1);
                const unsigned ofs = (unsigned)((pix * p.ldA + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) >> 3);
                const bf16_t* live_src = p.A + ((int64_t)ofs << 3);
                const bf16_t* src = ok ? live_src : p.zeros;
                __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(ws3_lds + buf_off + (unsigned)j * 1024u), 16, 0, 0);
            }
        }
    };

    if (ntl > 0) issue_patch(tbeg, 0u);
    if (ntl > 1) issue_patch(tbeg + nloc, g.patch_bytes);
    ws3_wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    for (int k = 0; k < ntl; k++) {
        const unsigned pbuf = (k & 1) ? g.patch_bytes : 0u;
        // (the buffer offset goes through an opaque VGPR: with a loop-invariant expression hipcc hoists all 72 fragment addresses out of the tile
        // loop — 72 registers of a kernel that has none to spare; patch_bytes is a multiple of 1024, so (a + p) ^ (ks << 5) == (a ^ (ks << 5)) + p)
        unsigned pbv = pbuf;
        asm volatile("" : "+v"(pbv));
        // accumulate epilogue: the 8 old 16-byte row segments of this lane are requested NOW and consumed after the tile's MFMAs (loading them
        // in the store phase exposed two HBM round trips per tile: as long as the tile itself)
        int pix0, oh0, ow0;
        tile_origin(tbeg + k * nloc, pix0, oh0, ow0);
        uint4 oldv[8];
        if (EPI == EPI_ACCUM) {
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int m = wm * 128 + it * 16 + r0;
                const int r = small_div(m, g.TW, g.rTW), c = m - r * g.TW;
                const bool lv = (m < g.TH * g.TW) & col_ok;
                oldv[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.out) + ((int64_t)pix0 + (lv ? r * W + c : 0)) * p.ldC +
                                                            (col_ok ? ncol : 0));
            }
        }
        f32x16 acc[4];
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[b][e] = 0.f;
        // ---- 144 MFMAs over the resident patch: unit u = (tap, K-step, pixel block); the fragment of unit u + PF is read while unit u multiplies --
        constexpr int NU = 144, PF = 4;
        bf16x8 fr[PF];
        const unsigned lbase = lds_addr(ws3_lds);
        auto rd = [&](int u) {
            const int t = u >> 4, ks = (u >> 2) & 3, b = u & 3;
            return ws3_rd128(lbase + ((aaddr[b][t] + pbv) ^ (unsigned)(ks << 5)));
        };
        auto pre = [&](auto uc) { constexpr int u = decltype(uc)::value; fr[u] = rd(u); };
        Ws3Unroll<0, PF>::run(pre);
        auto unit = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            __builtin_amdgcn_sched_barrier(0);
            // reads u + 1 ... u + PF - 1 were issued after read u: at most that many may still be in flight
            constexpr int later = NU - 1 - u < PF - 1 ? NU - 1 - u : PF - 1;
            ws3_wait_lds<later>(fr[u % PF]);
            const bf16x8 cur = fr[u % PF];
            if constexpr (u + PF < NU) fr[u % PF] = rd(u + PF);
            __builtin_amdgcn_sched_barrier(0);
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[u >> 4][(u >> 2) & 3], cur, acc[u & 3], 0, 0, 0);
        };
        Ws3Unroll<0, NU>::run(unit);
        __builtin_amdgcn_sched_barrier(0);
        // ---- accumulators -> the wave's staging block: lane owns pixel (b, l31), channels 8 g4 + 4 h + (0..3) of the wave's 32; dead pixels store zeros --
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                float v0 = acc[b][4 * g4], v1 = acc[b][4 * g4 + 1], v2 = acc[b][4 * g4 + 2], v3 = acc[b][4 * g4 + 3];
                if (!live[b]) v0 = v1 = v2 = v3 = 0.f;
                *reinterpret_cast<uint2*>(stg + ((b * 32 + l31) * WS3_STG_LD + 8 * g4 + 4 * h) * 2) = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
            }
        // ---- everyone is done with patch k (its buffer is free) and every piece of patch k + 1 has landed -------------------------------------
        ws3_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        // ---- store phase of tile k (wave-local: own ds_write -> ds_read ordering) -----------------------------------------------------------
        const unsigned sbase = lds_addr(stg);
        if (EPI == EPI_STATS) {
#pragma unroll
            for (int k4 = 0; k4 < 16; k4 += 4) {                  // 4 reads in flight
                ws3_u2 w[4];
#pragma unroll
                for (int q = 0; q < 4; q++) w[q] = ws3_rd64u(sbase + (unsigned)(((rg + 8 * (k4 + q)) * WS3_STG_LD + cq * 4) * 2));
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (q == 0) ws3_wait_lds<3>(w[q]);
                    else if (q == 1) ws3_wait_lds<2>(w[q]);
                    else if (q == 2) ws3_wait_lds<1>(w[q]);
                    else ws3_wait_lds<0>(w[q]);
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
        for (int g0 = 0; g0 < 8; g0 += 4) {                       // two groups of 4 rows: the loads of an accumulate epilogue are issued together
            bf16_t* optr[4];
            ws3_u4 sv[4];
            bool lvq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int m = wm * 128 + (g0 + q) * 16 + r0;               // tile-local pixel of staged row (g0 + q) * 16 + r0
                const int r = small_div(m, g.TW, g.rTW), c = m - r * g.TW;
                lvq[q] = m < g.TH * g.TW && col_ok;
                optr[q] = reinterpret_cast<bf16_t*>(p.out) + ((int64_t)pix0 + (lvq[q] ? r * W + c : 0)) * p.ldC + (col_ok ? ncol : 0);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) sv[q] = ws3_rd128u(sbase + (unsigned)((((g0 + q) * 16 + r0) * WS3_STG_LD + ch * 8) * 2));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q == 0) ws3_wait_lds<3>(sv[q]);
                else if (q == 1) ws3_wait_lds<2>(sv[q]);
                else if (q == 2) ws3_wait_lds<1>(sv[q]);
                else ws3_wait_lds<0>(sv[q]);
                uint4 v = make_uint4(sv[q].x, sv[q].y, sv[q].z, sv[q].w);
                if (EPI == EPI_ACCUM) {
                    const unsigned* a = reinterpret_cast<const unsigned*>(&v);
                    const unsigned* bb = reinterpret_cast<const unsigned*>(&oldv[g0 + q]);
                    unsigned w[4];
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        w[e] = pack_bf2(__uint_as_float(a[e] << 16) + __uint_as_float(bb[e] << 16),
                                        __uint_as_float(a[e] & 0xffff0000u) + __uint_as_float(bb[e] & 0xffff0000u));
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
                if (lvq[q]) *reinterpret_cast<uint4*>(optr[q]) = v;
            }
        }
        // ---- the patch two tiles ahead goes into the buffer this tile just released (requested LAST in the iteration: every LDS read above is
        //      already done, the next ones are the asm reads of the compute loop) ------------------------------------------------------------------
        if (k + 2 < ntl) issue_patch(tbeg + (k + 2) * nloc, pbuf);
    }
    if (EPI == EPI_STATS) {
        // one partial row per workgroup: lanes park their 8 sums in LDS, one thread per channel folds 2 pixel halves x 8 row groups in a fixed order
        __syncthreads();
        float* part = reinterpret_cast<float*>(ws3_lds);         // [wave][rg][2][32]
        float* mine = part + ((wave * 8 + rg) * 2) * 32 + cq * 4;
        *reinterpret_cast<float4*>(mine) = make_float4(ssum[0], ssum[1], ssum[2], ssum[3]);
        *reinterpret_cast<float4*>(mine + 32) = make_float4(ssq[0], ssq[1], ssq[2], ssq[3]);
        __syncthreads();
        if (tid < 64 && tid < p.Nout) {
            const int wnc = tid >> 5, cc = tid & 31;
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < 2; w++)
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float* src = part + (((w * 2 + wnc) * 8 + r) * 2) * 32 + cc;
                    sm += src[0];
                    sq += src[32];
                }
            float* st = p.stats + (int64_t)blockIdx.x * 2 * p.Nout;
            st[tid] = sm;
            st[p.Nout + tid] = sq;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- host side
static int ws3_enabled()
{
    static const int v = [] { const char* e = getenv("RYOLO_P3_WS64"); return e ? atoi(e) : 1; }();
    return v;
}
static int ws3_cus()
{
    static const int v = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t pr;
            if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount;
        }
        return n & ~7;
    }();
    return v;
}

bool ws3_geometry(const ConvGemmParams& p, Ws3Geom& g)
{
    g = Ws3Geom{};
    if (!ws3_enabled()) return false;
    const TapClass& tc = p.cls[0];
    if (p.nclasses != 1 || tc.ntaps != 9 || p.sh != 1 || p.sw != 1 || p.IH != p.OH || p.IW != p.OW) return false;
    if (p.pool_idx || p.s2d_cin || p.nbstat) return false;
    if (p.oh_mul != 1 || p.ow_mul != 1 || p.OHf != p.OH || p.OWf != p.OW || tc.oh_add || tc.ow_add) return false;
    if (p.Cin != 64 || p.Nout > 64 || p.Nout < 8 || p.Nout % 8 || p.ldA % 8 || p.ldC % 8 || !p.zeros) return false;
    if (p.epi != EPI_RAW && p.epi != EPI_STATS && p.epi != EPI_ACCUM) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; t++) {
        if (tc.dh[t] < -1 || tc.dh[t] > 1 || tc.dw[t] < -1 || tc.dw[t] > 1 || tc.widx[t] < 0 || tc.widx[t] >= p.wtaps) return false;
        seen |= 1u << ((tc.dh[t] + 1) * 3 + tc.dw[t] + 1);
    }
    if (seen != 0x1ffu) return false;
    const int H = p.OH, W = p.OW;
    const int64_t M = (int64_t)p.NB * H * W;
    if (M * (int64_t)p.ldA >= (1ll << 34) || M >= (1ll << 31) - 2 * W - 4) return false;        // 32-bit DMA source offsets / pixel indices
    int best_th = 0, best_tw = 0;
    for (int tw = 1; tw <= W && tw <= 256; tw++) {
        if (W % tw) continue;
        for (int th = 1; th <= H && th * tw <= 256; th++) {
            if (H % th) continue;
            const int area = th * tw, barea = best_th * best_tw;
            if (area > barea || (area == barea && (th + 2) * (tw + 2) < (best_th + 2) * (best_tw + 2))) { best_th = th; best_tw = tw; }
        }
    }
    if (best_th * best_tw < 224) return false;
    g.TH = best_th; g.TW = best_tw; g.PW = best_tw + 2;
    g.R = (best_th + 2) * (best_tw + 2);
    g.NP = (int)ry_cdiv(g.R, 8);
    if (g.NP > 48) return false;
    g.patch_bytes = (unsigned)g.NP * 1024u;
    g.lds_bytes = 2u * g.patch_bytes + (unsigned)WS3_WAVES * WS3_STG_BYTES;
    if (g.lds_bytes > 160u * 1024u) return false;
    g.tilesW = W / best_tw;
    g.tilesPerImg = (H / best_th) * g.tilesW;
    g.ntiles = (int64_t)p.NB * g.tilesPerImg;
    g.nwg = ws3_cus();
    // a persistent workgroup amortises its weight load and the pipeline fill over its tiles: small problems stay on the halo-patch kernel
    if (g.ntiles < 4ll * g.nwg && ws3_enabled() < 2) return false;
    if (g.ntiles < g.nwg) g.nwg = (int)((g.ntiles + 7) & ~7ll);
    for (int t = 0; t < 9; t++) { g.tdh[t] = tc.dh[t]; g.tdw[t] = tc.dw[t]; g.twi[t] = tc.widx[t]; }
    g.rPW = 1.0f / (float)g.PW;
    g.rTW = 1.0f / (float)g.TW;
    return true;
}

template <int EPI> static int ws3_launch_t(const ConvGemmParams& p, const Ws3Geom& g, hipStream_t stream)
{
    static RyLdsAttr attr;
    if (ry_max_dynamic_lds(attr, reinterpret_cast<const void*>(&conv3x3_ws64_kernel<EPI>), 160 * 1024)) return RY_ERR_LAUNCH;
    hipLaunchKernelGGL((conv3x3_ws64_kernel<EPI>), dim3((unsigned)g.nwg), dim3(64 * WS3_WAVES), g.lds_bytes, stream, p, g);
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}

int ws3_launch(const ConvGemmParams& p, const Ws3Geom& g, hipStream_t stream)
{
    if (p.epi == EPI_STATS) return ws3_launch_t<EPI_STATS>(p, g, stream);
    if (p.epi == EPI_ACCUM) return ws3_launch_t<EPI_ACCUM>(p, g, stream);
    return ws3_launch_t<EPI_RAW>(p, g, stream);
}
