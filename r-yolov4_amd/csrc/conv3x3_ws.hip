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

#define WS3_STG_BYTES (128 * 64)                               // one wave's staging block: 128 pixel rows x 32 channels (64-byte rows, 16-byte chunks
                                                               // XOR-swizzled with (row >> 1) & 3)
#define WS3_WAVES 4
#define WS3_PF 6                                               // pixel fragments in flight per wave (one wave per SIMD: nobody else hides the LDS latency)
#define WS3_NU 144                                             // MFMA units per tile and wave: 9 taps x 4 K-steps x 4 pixel blocks
#define WS3_NPW 12                                             // DMA pieces per wave and tile (pieces past the patch go to a dummy KiB)

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
// ahead right after requesting it (seen in the ISA of the first cut).  LDS operations of a wave return in order, so "at most N of the
// operations issued after X are still in flight" means X is done; scalar loads share the counter and return out of order, which can only make
// such a wait more conservative (X cannot be outstanding while fewer than N + 1 LDS operations are).
typedef unsigned ws3_u4 __attribute__((ext_vector_type(4)));
typedef unsigned ws3_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 ws3_rd128(unsigned addr)
{
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
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
// (LDS WRITES too: in front of a plain ds_write hipcc waits for every LDS-DMA in flight — here the pieces of the patch two tiles ahead, requested
// a few hundred cycles earlier)
__device__ __forceinline__ void ws3_wr64(unsigned addr, unsigned lo, unsigned hi)
{
    const ws3_u2 v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int N> __device__ __forceinline__ void ws3_wait_lds(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
// ... and the registers of EARLIER extra reads that this wait also covers (tied, so that their consumers cannot be scheduled above it)
template <int N> __device__ __forceinline__ void ws3_wait_lds(bf16x8& f, ws3_u2& a, ws3_u2& b, ws3_u2& c, ws3_u2& d)
{
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f), "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N> __device__ __forceinline__ void ws3_wait_lds(bf16x8& f, ws3_u4& a, ws3_u4& b, ws3_u4& c, ws3_u4& d)
{
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f), "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N> __device__ __forceinline__ void ws3_wait_lds(ws3_u2& a, ws3_u2& b, ws3_u2& c, ws3_u2& d)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N> __device__ __forceinline__ void ws3_wait_lds(ws3_u4& a, ws3_u4& b, ws3_u4& c, ws3_u4& d)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

// ---- what else rides in the MFMA shadow of a tile's 144 units (the work of the PREVIOUS tile's epilogue and of the patch two tiles ahead) ----
//   unit 50                     accumulate epilogue: the 8 old 16-byte row segments of THIS tile are requested (consumed a tile later)
//   units 1, 9, 17, 25          statistics: 4 staged 8-byte reads each, consumed WS3_PF units later
//   units 33, 41                4 staged 16-byte row segments each, stored to HBM WS3_PF units later
//   units 56 + 7 i, i = 0..11   DMA piece i of the patch two tiles ahead (LAST among the tile's memory operations: "all but the newest 12
//                               have completed" is then exactly "the patch of the NEXT tile has landed", whatever the stores did)
__host__ __device__ constexpr int ws3_extra_reads(int u, bool first, bool stats)
{
    if (first) return 0;
    if (stats && (u == 1 || u == 9 || u == 17 || u == 25)) return 4;
    if (u == 33 || u == 41) return 4;
    return 0;
}
// LDS operations issued after fragment read v and before unit v waits for it (unit w: wait, consume, extra reads, fragment read w + PF, MFMA)
__host__ __device__ constexpr int ws3_later(int v, bool first, bool stats)
{
    int n = v < WS3_PF - 1 ? WS3_PF - 1 - v : 0;
    for (int w = (v - WS3_PF + 1 > 0 ? v - WS3_PF + 1 : 0); w < v; w++) n += ws3_extra_reads(w, first, stats) + (w + WS3_PF < WS3_NU ? 1 : 0);
    return n;
}

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_ws64_kernel(const ConvGemmParams p, const Ws3Geom g)
{
    constexpr bool STATS = EPI == EPI_STATS, ACCUM = EPI == EPI_ACCUM, AFFINE = EPI == EPI_AFFINE_ACT;
    constexpr int PF = WS3_PF, NU = WS3_NU;
#ifdef WS3_TIMING
    const unsigned long long TT0 = __builtin_readcyclecounter();
    unsigned long long t_loop = 0, t_tail = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // scalar: uniform branches, scalar LDS bases
    const int wm = wave >> 1, wn = wave & 1;                   // wave = pixel half of the tile (128 pixels) x output-channel half (32)
    const int h = lane >> 5, l31 = lane & 31;
    const int H = p.OH, W = p.OW;
    const unsigned PB = g.patch_bytes;
    const unsigned lbase = lds_addr(ws3_lds);
    const unsigned sbase = lbase + 3u * PB + (unsigned)wave * WS3_STG_BYTES;          // this wave's staging block
    const unsigned dummy_off = 3u * PB + WS3_WAVES * WS3_STG_BYTES;                  // 1 KiB nobody reads

    // ---- my tiles: XCD x owns the contiguous range [x * T8, (x + 1) * T8); its workgroups take them round robin -----------------------------
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int nloc = gridDim.x >> 3;                           // grid is a multiple of 8
    const int T8 = (int)((g.ntiles + 7) >> 3);
    const int tbeg = xcd * T8 + loc;
    const int tend = min((xcd + 1) * T8, (int)g.ntiles);
    const int ntl = tbeg < tend ? (tend - tbeg + nloc - 1) / nloc : 0;

    // ---- weights -> registers (A operands: row = output channel wn * 32 + l31, K = 16 input channels of step ks, 8 per lane half) -----------
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

    // ---- patch DMA: piece j = wave + 4 u covers patch rows 8 j ... 8 j + 7, lane -> (row, 16-byte slot).  Per piece and lane, once per launch:
    //      the element offset of its 16 bytes relative to the tile's first pixel, and 5 flag bits — the row belongs to the top / bottom / left /
    //      right halo, or does not exist; a tile ANDs them with "which of its edges lie on the image border" (dead lanes read the zero page)
    int d_rel[WS3_NPW];
    unsigned dfl0 = 0, dfl1 = 0;
#pragma unroll
    for (int u = 0; u < WS3_NPW; u++) {
        const int j = wave + WS3_WAVES * u;
        const int row = 8 * j + (lane >> 3);
        const int pr = small_div(row, g.PW, g.rPW), pc = row - pr * g.PW;
        d_rel[u] = ((pr - 1) * W + (pc - 1)) * p.ldA + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
        const unsigned fl = (pr == 0 ? 1u : 0u) | (pr == g.TH + 1 ? 2u : 0u) | (pc == 0 ? 4u : 0u) | (pc == g.TW + 1 ? 8u : 0u) |
                            ((j >= g.NP || row >= g.R) ? 16u : 0u);
        if (u < 6) dfl0 |= fl << (5 * u);
        else dfl1 |= fl << (5 * (u - 6));
    }
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
    unsigned abase_now = 0u;                                   // what the addresses above are currently relative to (LDS address of the patch buffer in use)
    // ---- inference (EPI_AFFINE_ACT): the folded BatchNorm coefficients of this lane's 16 output channels (accumulator element 4 g4 + q of every
    //      pixel block = channel wn * 32 + 8 g4 + 4 h + q), in registers for the whole launch — the eval tape ran these layers on the halo-patch
    //      kernel's 256 x 64 tile until r06 (656 against 880 TF/s on the 400^2 map) -------------------------------------------------------------------
    float asc[AFFINE ? 4 : 1][4], asf[AFFINE ? 4 : 1][4];
    if constexpr (AFFINE) {
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = wn * 32 + 8 * g4 + 4 * h + q;
                asc[g4][q] = c < p.Nout ? p.scale[c] : 0.f;
                asf[g4][q] = c < p.Nout ? p.shift[c] : 0.f;
            }
    }
    // ---- store side: lane -> 16-byte chunk ch of staged row it * 16 + r0 (chunk position ^ (row >> 1) & 3) ------------------------------------
    const int ch = lane & 3, r0 = lane >> 2;
    const int ncol = wn * 32 + ch * 8;
    const bool col_ok = ncol < p.Nout;
    // statistics: lane -> channel quad cq (4 channels = half a chunk), rows rg + 8 k
    const int cq = lane & 7, rg = lane >> 3;
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    auto stg_addr16 = [&](int row) { return sbase + (unsigned)(row * 64 + ((ch ^ ((row >> 1) & 3)) << 4)); };
    auto stg_addr8 = [&](int row) { return sbase + (unsigned)(row * 64 + ((((cq >> 1) ^ ((row >> 1) & 3)) << 4) | ((cq & 1) << 3))); };

    struct TileAt { int pix0; unsigned em; };                   // first pixel, edges on the image border (flag bits as above; 16 always set)
    auto tile_at = [&](int tile) {
        const int img = tile / g.tilesPerImg;
        const int rem = tile - img * g.tilesPerImg;
        const int th = rem / g.tilesW;
        const int oh0 = th * g.TH, ow0 = (rem - th * g.tilesW) * g.TW;
        TileAt t;
        t.pix0 = (img * H + oh0) * W + ow0;
        t.em = 16u | (oh0 == 0 ? 1u : 0u) | (oh0 + g.TH == H ? 2u : 0u) | (ow0 == 0 ? 4u : 0u) | (ow0 + g.TW == W ? 8u : 0u);
        return t;
    };
    // one DMA piece; have == false (no such tile): every lane reads the zero page into the dummy KiB — the instruction is ALWAYS issued, so the
    // per-tile count of memory operations is a constant (the end-of-tile wait counts them)
    auto issue_piece = [&](int u, bool have, const TileAt& t, unsigned buf_off) {
        const unsigned fl = ((u < 6 ? dfl0 >> (5 * u) : dfl1 >> (5 * (u - 6))) & 31u);
        const bool dead = (fl & (have ? t.em : 31u)) != 0u;
        const bf16_t* src = dead ? p.zeros : p.A + ((int64_t)t.pix0 * p.ldA + d_rel[u]);
        const int j = wave + WS3_WAVES * u;
        const unsigned dst = (have && j < g.NP) ? buf_off + (unsigned)j * 1024u : dummy_off;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(ws3_lds + dst), 16, 0, 0);
    };
    // output row segment of staged row `row` of the tile at pix0 (dead rows: pixel 0 of the tile, never stored)
    auto out_ptr = [&](int pix0, int row, bool& lv) {
        const int m = wm * 128 + row;
        const int r = small_div(m, g.TW, g.rTW), c = m - r * g.TW;
        lv = (m < g.TH * g.TW) & col_ok;
        return reinterpret_cast<bf16_t*>(p.out) + ((int64_t)pix0 + (lv ? r * W + c : 0)) * p.ldC + (col_ok ? ncol : 0);
    };
    auto stat_add = [&](const ws3_u2& w) {
        const float f0 = __uint_as_float(w.x << 16), f1 = __uint_as_float(w.x & 0xffff0000u);
        const float f2 = __uint_as_float(w.y << 16), f3 = __uint_as_float(w.y & 0xffff0000u);
        ssum[0] += f0; ssq[0] += f0 * f0;
        ssum[1] += f1; ssq[1] += f1 * f1;
        ssum[2] += f2; ssq[2] += f2 * f2;
        ssum[3] += f3; ssq[3] += f3 * f3;
    };
    auto store_seg = [&](bf16_t* o, bool lv, const ws3_u4& sv, const uint4& old) {
        uint4 v = make_uint4(sv.x, sv.y, sv.z, sv.w);
        if (ACCUM) {
            const unsigned* a = reinterpret_cast<const unsigned*>(&v);
            const unsigned* bb = reinterpret_cast<const unsigned*>(&old);
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
                w[e] = pack_bf2(__uint_as_float(a[e] << 16) + __uint_as_float(bb[e] << 16),
                                __uint_as_float(a[e] & 0xffff0000u) + __uint_as_float(bb[e] & 0xffff0000u));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        if (lv) *reinterpret_cast<uint4*>(o) = v;
    };

    uint4 oldv[8];                                             // accumulate epilogue: old row segments of the tile whose store phase comes next
    // ---- one tile: 144 MFMAs from patch buffer `bcur`; in their shadow the epilogue of the previous tile (at prev_pix0) and the DMA of the tile
    //      two ahead (into `bfar`, the buffer the previous tile used) -------------------------------------------------------------------------------
    auto compute = [&](auto first_tag, unsigned bcur, unsigned bfar, int far_tile, int prev_pix0, int cur_pix0) {
        constexpr bool FIRST = decltype(first_tag)::value;
        // the 36 fragment row addresses follow the patch buffer (one add each per tile; per unit only the K-step XOR is left: patch_bytes is a
        // multiple of 1024, so (a + p) ^ (ks << 5) == (a ^ (ks << 5)) + p)
        {
            const unsigned delta = lbase + bcur - abase_now;
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int t = 0; t < 9; t++) aaddr[b][t] += delta;
            abase_now = lbase + bcur;
        }
        const bool have_far = far_tile >= 0;
        TileAt far;
        far.pix0 = 0;
        far.em = 31u;
        f32x16 acc[4];
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[b][e] = 0.f;
        bf16x8 fr[PF];
        ws3_u2 sw[4];
        ws3_u4 sv[4];
        bf16_t* optr[4];
        bool olv[4];
        auto rd = [&](int u) {
            const int t = u >> 4, ks = (u >> 2) & 3, b = u & 3;
            return ws3_rd128(ks ? aaddr[b][t] ^ (unsigned)(ks << 5) : aaddr[b][t]);
        };
#ifdef WS3_TIMING
        const unsigned long long tc0 = __builtin_readcyclecounter();
#endif
        auto pre = [&](auto uc) { constexpr int u = decltype(uc)::value; fr[u] = rd(u); };
        Ws3Unroll<0, PF>::run(pre);
        auto unit = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int later = ws3_later(u, FIRST, STATS);
            static_assert(later <= 15, "lgkmcnt is a 4-bit counter");
            __builtin_amdgcn_sched_barrier(0);
            // ---- wait for fragment u (and for the extra reads issued PF units ago, which precede fragment u in the LDS queue) ---------------------
            constexpr bool use_stats = !FIRST && STATS && (u == 1 + PF || u == 9 + PF || u == 17 + PF || u == 25 + PF);
            constexpr bool use_store = !FIRST && (u == 33 + PF || u == 41 + PF);
            if constexpr (use_stats) ws3_wait_lds<later>(fr[u % PF], sw[0], sw[1], sw[2], sw[3]);
            else if constexpr (use_store) ws3_wait_lds<later>(fr[u % PF], sv[0], sv[1], sv[2], sv[3]);
            else ws3_wait_lds<later>(fr[u % PF]);
            const bf16x8 cur = fr[u % PF];
            if constexpr (use_stats) {
#pragma unroll
                for (int q = 0; q < 4; q++) stat_add(sw[q]);
            }
            if constexpr (use_store) {
                constexpr int g0 = u == 33 + PF ? 0 : 4;
#pragma unroll
                for (int q = 0; q < 4; q++) store_seg(optr[q], olv[q], sv[q], oldv[g0 + q]);
            }
            // ---- extra requests of this unit --------------------------------------------------------------------------------------------------
            if constexpr (ACCUM && u == 50) {
                // the old values of THIS tile, for the store phase that runs under the NEXT tile's MFMAs (requested most of a tile ahead: loading
                // them inside that store phase exposed an HBM round trip per tile; the previous tile's were consumed at units 39 / 47)
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    bool lv;
                    const bf16_t* o = out_ptr(cur_pix0, it * 16 + r0, lv);
                    oldv[it] = *reinterpret_cast<const uint4*>(o);
                }
            }
            if constexpr (u == 52) {
                // scalar index arithmetic (two divisions) in the shadow of the MFMAs; branch-free: a join in this stream would make hipcc drain
                // its counters (no tile two ahead: tile 0's origin with every lane dead)
                far = tile_at(have_far ? far_tile : 0);
                far.em = have_far ? far.em : 31u;
            }
            if constexpr (!FIRST && STATS && (u == 1 || u == 9 || u == 17 || u == 25)) {
                constexpr int k4 = (u - 1) / 2;                   // rows rg + 8 (k4 + q): groups 0, 4, 8, 12
#pragma unroll
                for (int q = 0; q < 4; q++) sw[q] = ws3_rd64u(stg_addr8(rg + 8 * (k4 + q)));
            }
            if constexpr (!FIRST && (u == 33 || u == 41)) {
                constexpr int g0 = u == 33 ? 0 : 4;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    optr[q] = out_ptr(prev_pix0, (g0 + q) * 16 + r0, olv[q]);
                    sv[q] = ws3_rd128u(stg_addr16((g0 + q) * 16 + r0));
                }
            }
            if constexpr (u >= 56 && (u - 56) % 7 == 0 && (u - 56) / 7 < WS3_NPW) issue_piece((u - 56) / 7, have_far, far, bfar);
            if constexpr (u + PF < NU) fr[u % PF] = rd(u + PF);
            __builtin_amdgcn_sched_barrier(0);
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[u >> 4][(u >> 2) & 3], cur, acc[u & 3], 0, 0, 0);
        };
        Ws3Unroll<0, NU>::run(unit);
        __builtin_amdgcn_sched_barrier(0);
#ifdef WS3_TIMING
        const unsigned long long tc1 = __builtin_readcyclecounter();
        t_loop += tc1 - tc0;
#endif
        // ---- accumulators -> the wave's staging block: lane owns pixel (b, l31), channels 8 g4 + 4 h + (0..3) of the wave's 32; dead pixels store
        //      zeros.  (The previous tile's staged rows were all read above: same wave, program order.) ------------------------------------------
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                float v0 = acc[b][4 * g4], v1 = acc[b][4 * g4 + 1], v2 = acc[b][4 * g4 + 2], v3 = acc[b][4 * g4 + 3];
                if constexpr (AFFINE) {
                    float v[4] = {v0, v1, v2, v3};
                    act_affine_quad(v, asc[g4], asf[g4], p.act);
                    v0 = v[0]; v1 = v[1]; v2 = v[2]; v3 = v[3];
                }
                if (!live[b]) v0 = v1 = v2 = v3 = 0.f;
                const int row = b * 32 + l31;
                ws3_wr64(sbase + (unsigned)(row * 64 + (((g4 ^ ((row >> 1) & 3)) << 4) | (h << 3))), pack_bf2(v0, v1), pack_bf2(v2, v3));
            }
        // ---- all but this tile's 12 DMA pieces have completed: the NEXT tile's patch (requested one tile ago) has landed; then everyone is done
        //      with the current buffer -----------------------------------------------------------------------------------------------------------
        ws3_wait_vm<WS3_NPW>();
        __builtin_amdgcn_s_barrier();
#ifdef WS3_TIMING
        t_tail += __builtin_readcyclecounter() - tc1;
#endif
        return far.pix0;                                          // first pixel of the tile two ahead (0 when there is none)
    };

    // ---- prologue: the first two patches ----------------------------------------------------------------------------------------------------
    {
        const TileAt t0 = tile_at(ntl > 0 ? tbeg : 0), t1 = tile_at(ntl > 1 ? tbeg + nloc : 0);
#pragma unroll
        for (int u = 0; u < WS3_NPW; u++) issue_piece(u, ntl > 0, t0, 0u);
#pragma unroll
        for (int u = 0; u < WS3_NPW; u++) issue_piece(u, ntl > 1, t1, PB);
    }
    ws3_wait_vm<0>();
    __builtin_amdgcn_s_barrier();

#ifdef WS3_TIMING
    const unsigned long long TT1 = __builtin_readcyclecounter();
#endif
    unsigned b0 = 0u, b1 = PB, b2 = 2u * PB;                   // buffers of tile k, k + 1, k + 2 (= the one tile k - 1 used)
    int prev_pix0 = 0, cur_pix0 = ntl > 0 ? tile_at(tbeg).pix0 : 0, next_pix0 = ntl > 1 ? tile_at(tbeg + nloc).pix0 : 0;
    for (int k = 0; k < ntl; k++) {
        const int far_tile = k + 2 < ntl ? tbeg + (k + 2) * nloc : -1;
        const int far_pix0 = k == 0 ? compute(std::true_type{}, b0, b2, far_tile, 0, cur_pix0)
                                    : compute(std::false_type{}, b0, b2, far_tile, prev_pix0, cur_pix0);
        prev_pix0 = cur_pix0;
        cur_pix0 = next_pix0;
        next_pix0 = far_pix0;
        const unsigned t = b0;
        b0 = b1; b1 = b2; b2 = t;
    }
    // ---- epilogue of the last tile (nothing left to hide it under) -----------------------------------------------------------------------------
    if (ntl > 0) {
        if (STATS) {
#pragma unroll
            for (int k4 = 0; k4 < 16; k4 += 4) {
                ws3_u2 w[4];
#pragma unroll
                for (int q = 0; q < 4; q++) w[q] = ws3_rd64u(stg_addr8(rg + 8 * (k4 + q)));
                ws3_wait_lds<0>(w[0], w[1], w[2], w[3]);
#pragma unroll
                for (int q = 0; q < 4; q++) stat_add(w[q]);
            }
        }
#pragma unroll
        for (int g0 = 0; g0 < 8; g0 += 4) {
            bf16_t* o[4];
            bool lv[4];
            ws3_u4 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                o[q] = out_ptr(prev_pix0, (g0 + q) * 16 + r0, lv[q]);
                v[q] = ws3_rd128u(stg_addr16((g0 + q) * 16 + r0));
            }
            ws3_wait_lds<0>(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int q = 0; q < 4; q++) store_seg(o[q], lv[q], v[q], oldv[g0 + q]);       // (old values: requested under the last tile's MFMAs)
        }
    }
#ifdef WS3_TIMING
    if (p.bias && tid == 0) {                                   // tools/bench_conv.py: [prologue, sum of MFMA loops, sum of tile tails, whole kernel], tiles
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias)) + (size_t)blockIdx.x * 4;
        dbg[0] = TT1 - TT0; dbg[1] = t_loop; dbg[2] = t_tail; dbg[3] = ((__builtin_readcyclecounter() - TT0) << 12) | (unsigned)ntl;
    }
#endif
    if (STATS) {
        // one partial row per workgroup: lanes park their 8 sums in LDS, one thread per channel folds 2 pixel halves x 8 row groups in a fixed order
        ws3_wait_vm<0>();                                        // (the dummy DMA pieces of the last tiles: nothing may still write LDS)
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
    if (p.pool_idx || p.s2d_cin) return false;
    if (p.oh_mul != 1 || p.ow_mul != 1 || p.OHf != p.OH || p.OWf != p.OW || tc.oh_add || tc.ow_add) return false;
    if (p.Cin != 64 || p.Nout > 64 || p.Nout < 8 || p.Nout % 8 || p.ldA % 8 || p.ldC % 8 || !p.zeros) return false;
    if (p.epi != EPI_RAW && p.epi != EPI_STATS && p.epi != EPI_ACCUM && p.epi != EPI_AFFINE_ACT) return false;
    if (p.epi == EPI_AFFINE_ACT && (!p.scale || !p.shift)) return false;
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
    g.lds_bytes = 3u * g.patch_bytes + (unsigned)WS3_WAVES * WS3_STG_BYTES + 1024u;
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
    if (p.epi == EPI_AFFINE_ACT) return ws3_launch_t<EPI_AFFINE_ACT>(p, g, stream);
    return ws3_launch_t<EPI_RAW>(p, g, stream);
}
