// Streaming 3x3 stride-2 convolution for 32 input channels (gfx950 / MI355X): forward of the SECOND layer of the yolov4 / yolov7 backbones
// (model/backbone.py: Conv(32, 64, 3, 2) right behind the stem; model/utils.py:13-23) — 800 x 800 x 32 -> 400 x 400 x 64 at the benchmark size,
// 2.6 GB in + 1.3 GB out per 64-image launch for 377 GFLOP: 96 FLOP per byte, BYTE-bound on a part whose ridge is ~310 FLOP / B (r05 floor table:
// 624 us of bytes, 151 us of matrix work).  On the generic implicit GEMM this launch ran its 256 x 64 tile at 354-369 TF/s = 1 023 us: N = 64
// columns give 32 MFMAs per 20 LDS-DMA pieces and every input row is requested once per tap (9 x), so the loop is bound by the ISSUE of its
// LDS-DMA requests, not by bytes.  ("put the 32 -> 64 @800^2 -> 400^2 layer ... on a streaming kernel like the stem's".)
//
// This kernel has NO LDS-DMA and no K loop over staged tiles:
//   * K = 9 taps x 32 channels = 288.  The whole weight matrix (64 x 288 bf16 = 36 KiB) sits in LDS for the life of the (persistent, one per CU)
//     workgroup in FRAGMENT-MAJOR order: fragment (tap, output-channel quarter) is 64 lanes x 16 bytes, read with one lane-linear ds_read_b128
//     (conflict-free by construction).
//   * A wave owns tiles of 32 output pixels (one output row segment) x all 64 output channels: 8 accumulator tiles of 16 x 16, 72 MFMAs
//     (v_mfma_f32_16x16x32_bf16, A = weights, B = pixels: a lane ends up with 4 consecutive channels of ONE pixel).  K = 32 per MFMA = ALL input
//     channels of a tap: a B fragment is 16 pixels x 64 contiguous bytes.  (The first cut used 32x32x16: 32 pixels x 32 bytes per load instruction —
//     1 006 us, and the counters named the bound: TCP_TOTAL_CACHE_ACCESSES 395 M per launch = 0.77 tag look-ups per cycle and CU, FETCH_SIZE 1.05 x the
//     tensor: the L1's tag rate, not HBM.  Half as many pixels per instruction = half the look-ups per byte.)
//   * The B fragment of (tap, pixel half) for lane (pixel p, channel octet) is 8 consecutive channels of input pixel (2 oh + kh - 1, 2 ow + kw - 1):
//     ONE 16-byte global load straight into the MFMA operand registers.  The 18 loads of tile t + 1 are issued before the MFMAs of tile t
//     (two register sets, 144 VGPRs): 18 KiB in flight per wave, 147 KiB per CU — what a byte-bound kernel needs to cover HBM latency.
//     The nine taps of a tile touch 3 input rows x 65 pixels; a wave walks DOWN the output rows of its 32-pixel column block, so two of the three
//     rows were fetched by its previous tile (L2 / Infinity Cache hits): every input byte leaves HBM once (+ 1 / 64 halo).
//   * Epilogue: the tile is staged in 4 KiB of wave-private LDS (packed 8-byte writes, swizzled) and leaves as whole 128-byte pixel rows
//     (16 bytes per lane); BatchNorm batch statistics are column sums of the STORED bf16 values read back from the staged tile, kept in 8
//     registers per lane over all tiles of the wave and folded per workgroup in a fixed order at the end: one [2][Nout] row per workgroup.
// Epilogues: raw, raw + statistics (training), folded BatchNorm + activation (inference plans).  Values: fp32 accumulation of the same bf16
// products in another order than the generic kernel (step order (tap, channel) is the same; no split) — parity tests hold both to the reference.
#include "conv_internal.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;

#define S2C_WAVES 8
#define S2C_STEPS 18                       // B fragments of a tile: 9 taps x 2 pixel halves of 16 (v_mfma_f32_16x16x32_bf16: one fragment = 16 pixels x 32 channels)
#define S2C_WFRAG 36                       // weight fragments: (tap, output-channel quarter), 1 KiB each
#define S2C_W_ELEMS (S2C_WFRAG * 64 * 8)   // 18 432 bf16 = 36 KiB
#define S2C_STAGE_ELEMS (32 * 64)          // per wave: [32 px][64 ch] bf16 = 4 KiB

template <int EPI>
__global__ __launch_bounds__(512, 1) void conv3x3s2_c32_kernel(const ConvGemmParams p, const int cblocks, const int64_t tiles)
{
    extern __shared__ __attribute__((aligned(16))) bf16_t s2c_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    bf16_t* const wl = s2c_lds;                                                // weights, fragment-major
    bf16_t* const stage = s2c_lds + S2C_W_ELEMS + wave * S2C_STAGE_ELEMS;      // this wave's staging tile
    float* const red = reinterpret_cast<float*>(s2c_lds + S2C_W_ELEMS + S2C_WAVES * S2C_STAGE_ELEMS);   // [8 waves][2][64] statistics fold / [2][64] coefficients

    // ---- weights -> LDS, fragment-major: fragment f = 4 * tap + quarter; lane l of it = row (16 * quarter + l % 16), channels [8 (l / 16), + 8) of the tap ----
    for (int idx = tid; idx < S2C_WFRAG * 64; idx += 512) {
        const int f = idx >> 6, l = idx & 63;
        const int row = (f & 3) * 16 + (l & 15);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < p.Nout) v = *reinterpret_cast<const uint4*>(p.W + (int64_t)row * (9 * 32) + (f >> 2) * 32 + (l >> 4) * 8);
        *reinterpret_cast<uint4*>(wl + (int64_t)idx * 8) = v;
    }
    if constexpr (EPI == EPI_AFFINE_ACT) {
        if (tid < 128) {
            const int c = tid & 63;
            red[tid] = c < p.Nout ? (tid < 64 ? p.scale[c] : p.shift[c]) : 0.f;
        }
    }
    __syncthreads();

    const int pn = lane & 15, kq = lane >> 4;                                  // pixel of a 16-pixel half, channel octet (B operand) / channel quad of a 16-channel quarter (accumulator)
    const int64_t gw = (int64_t)blockIdx.x * S2C_WAVES + wave, nw = (int64_t)gridDim.x * S2C_WAVES;
    // contiguous tile range per wave: consecutive tiles of a wave are consecutive OUTPUT ROWS of one (image, column block)
    const int64_t t0 = tiles * gw / nw, t1 = tiles * (gw + 1) / nw;
    const int64_t per_img = (int64_t)cblocks * p.OH;
    const bf16_t* const zp = p.zeros;
    // position of a tile, carried instead of divided out per tile (wave-uniform: scalar registers)
    struct Pos { int n, cb, oh; };
    auto advance = [&](Pos& q) {
        q.oh++;
        if (q.oh == p.OH) { q.oh = 0; q.cb++; if (q.cb == cblocks) { q.cb = 0; q.n++; } }
    };

    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};

    // the 18 B fragments of a tile: fragment (tap, half u), lane (pn, kq) reads channels [8 kq, + 8) of input pixel
    // (2 oh + kh - 1, 2 (32 cb + 16 u + pn) + kw - 1) — four lanes cover the pixel's 64 contiguous bytes, an instruction touches 16 pixels
    // (the 32x32x16 form of the first cut touched 32 pixels x 32 bytes per instruction: PMC showed the L1's tag rate, not HBM, as its bound)
    auto load_frag = [&](const Pos& q, int tap, int u) -> bf16x8 {               // (tap, u are compile-time at every call site)
        const int kh = tap / 3, kw = tap - 3 * kh;
        const int iy = 2 * q.oh + kh - 1;
        const int ow = q.cb * 32 + u * 16 + pn;
        const int ix = 2 * ow + kw - 1;
        const bool ok = (unsigned)iy < (unsigned)p.IH && ow < p.OW && (unsigned)ix < (unsigned)p.IW;
        const bf16_t* src = ok ? p.A + (((int64_t)q.n * p.IH + iy) * p.IW + ix) * p.ldA + kq * 8 : zp;
        return *reinterpret_cast<const bf16x8*>(src);
    };
    auto load_tile = [&](const Pos& q, bf16x8 (&b)[S2C_STEPS]) {
#pragma unroll
        for (int tap = 0; tap < 9; tap++)
#pragma unroll
            for (int u = 0; u < 2; u++) b[tap * 2 + u] = load_frag(q, tap, u);
    };

    // (Measured and not kept: refilling every fragment with the tile after next as soon as its MFMAs are issued — two tiles of loads in flight per
    // wave without a third register set.  Same box, three alternating runs: forward 1 058 vs 1 035 us, data gradient 1 151 vs 1 089 us — SLOWER: the
    // launch is not short of bytes in flight.  What the counters say about the 3.8-4.0 TB/s it reaches: DESIGN.md section 3.3.)
    auto compute_tile = [&](const Pos& q, const bf16x8 (&b)[S2C_STEPS]) {
        f32x4 acc[2][4];                                                       // [pixel half u][output-channel quarter]: rows 4 kq + i of the quarter, column pn
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int qq = 0; qq < 4; qq++) acc[u][qq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const bf16x8 w = *reinterpret_cast<const bf16x8*>(wl + ((tap * 4 + qq) * 64 + lane) * 8);
                acc[0][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b[tap * 2], acc[0][qq], 0, 0, 0);
                acc[1][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b[tap * 2 + 1], acc[1][qq], 0, 0, 0);
            }
        }
        // ---- epilogue: stage [32 px][64 ch]; 16-byte chunk ck of pixel row r sits at chunk ck ^ (r & 7) ----
        const int n = q.n, cb = q.cb, oh = q.oh;
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[u][qq][e];
                const int c0 = 16 * qq + 4 * kq;                                // the lane's 4 consecutive output channels of pixel row r
                if constexpr (EPI == EPI_AFFINE_ACT) {
                    const float4 sc = *reinterpret_cast<const float4*>(red + c0), sf = *reinterpret_cast<const float4*>(red + 64 + c0);
                    const float sc4[4] = {sc.x, sc.y, sc.z, sc.w}, sf4[4] = {sf.x, sf.y, sf.z, sf.w};
                    act_affine_quad(v, sc4, sf4, p.act);
                }
                const int r = 16 * u + pn, ck = c0 >> 3;
                *reinterpret_cast<uint2*>(stage + r * 64 + ((ck ^ (r & 7)) << 3) + (c0 & 4)) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        // (same-wave LDS hand-off: the wave's own ds_write -> ds_read ordering)
        const int c8 = lane & 7, r8 = lane >> 3;                                // store phase: chunk c8 of rows r8 + 8 it
        const int64_t orow = ((int64_t)n * p.OH + oh) * p.OW + cb * 32;
        const int live = p.OW - cb * 32;                                       // pixels of this block inside the image (>= 32: all)
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int r = it * 8 + r8;
            const uint4 v = *reinterpret_cast<const uint4*>(stage + r * 64 + ((c8 ^ (r & 7)) << 3));
            if (r < live && c8 * 8 < p.Nout) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (orow + r) * p.ldC + c8 * 8) = v;
        }
        if constexpr (EPI == EPI_STATS) {
            // column sums of the values actually stored (bf16-rounded); pixels outside the image are exact zeros (their taps read the zero page):
            // lane -> (4-channel quad cq, row group rg), rows rg + 4 k
            const int cq = lane & 15, rg = lane >> 4;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int r = rg + 4 * k;
                const uint2 w = *reinterpret_cast<const uint2*>(stage + r * 64 + (((cq >> 1) ^ (r & 7)) << 3) + (cq & 1) * 4);
                const float f0 = __uint_as_float(w.x << 16), f1 = __uint_as_float(w.x & 0xffff0000u);
                const float f2 = __uint_as_float(w.y << 16), f3 = __uint_as_float(w.y & 0xffff0000u);
                ssum[0] += f0; ssq[0] += f0 * f0;
                ssum[1] += f1; ssq[1] += f1 * f1;
                ssum[2] += f2; ssq[2] += f2 * f2;
                ssum[3] += f3; ssq[3] += f3 * f3;
            }
        }
        // the staged reads are done before the next tile's writes land in the same tile (the wave's own LDS operations retire in order;
        // the explicit wait keeps the compiler from hoisting the next tile's first ds_write above the last ds_read)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // Two register sets: the 18 loads of the NEXT tile are in flight under the MFMAs and the epilogue of the current one.  The steady state has no
    // load that depends on "is there a next tile" (a load under such a branch is a CFG join where hipcc drains vmcnt(0)): past the wave's last tile
    // the head cursor simply stays on it and the (unused) fragments are loaded again.
    bf16x8 ba[S2C_STEPS], bb[S2C_STEPS];
    if (t0 < t1) {
        Pos head;
        head.n = (int)(t0 / per_img);
        const int rem0 = (int)(t0 - (int64_t)head.n * per_img);
        head.cb = rem0 / p.OH;
        head.oh = rem0 - head.cb * p.OH;
        int64_t th = t0;                                                       // tile index of `head`: the newest tile whose loads were issued
        auto next = [&]() { if (th + 1 < t1) { advance(head); th++; } return head; };
        Pos pa = head;
        load_tile(pa, ba);
        for (int64_t t = t0; t < t1; t += 2) {
            const Pos pb = next();
            load_tile(pb, bb);                                                 // tile t + 1: in flight under the MFMAs and the epilogue of tile t
            compute_tile(pa, ba);
            if (t + 1 >= t1) break;
            pa = next();
            load_tile(pa, ba);                                                 // tile t + 2 under tile t + 1
            compute_tile(pb, bb);
        }
    }

    if constexpr (EPI == EPI_STATS) {
        // per-wave sums: lane (cq, rg) holds 4 channels x its 8 rows per tile; fold the 4 row groups, then the 8 waves, in a fixed order
        __syncthreads();                                                       // (every wave is done with its staging tile; `red` is free)
        float* mine = red + wave * 128;
        const int cq = lane & 15, rg = lane >> 4;
        // stage per-lane values in the wave's (dead) staging tile as floats: [rg][2][64]
        float* tmp = reinterpret_cast<float*>(stage);                          // 4 KiB = 1024 floats: [4 rg][2][64] = 512 floats
#pragma unroll
        for (int e = 0; e < 4; e++) {
            tmp[(rg * 2 + 0) * 64 + cq * 4 + e] = ssum[e];
            tmp[(rg * 2 + 1) * 64 + cq * 4 + e] = ssq[e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < 128; i += 64) mine[i] = ((tmp[i] + tmp[128 + i]) + tmp[256 + i]) + tmp[384 + i];
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < S2C_WAVES; w++) a += red[w * 128 + which * 64 + c];
            if (c < p.Nout) p.stats[((int64_t)blockIdx.x * 2 + which) * p.Nout + c] = a;
        }
    }
}

// =====================================================================================================================
// DATA GRADIENT of the same layer, in its space-to-depth form (ConvGemmParams.s2d_cin == 32; conv.hip pack_s2d_kernel): ONE stride-1 problem over
// the dY grid [NB, OH, OW, 64] — for dY pixel (a, b) the four input pixels (2a + ph, 2b + pw) x 32 channels from the taps (a + da, b + db),
// da <= ph, db <= pw: 9 live (parity, tap) blocks of the 16, 36 MFMAs per 32 dY pixels like the forward.  dY 1.3 GB in, dx 2.6 GB out at the
// benchmark size: byte-bound (624 us of bytes).  On the persistent pointwise kernel's S2D instantiation (gemm1x1.hip) this launch took
// 1 165-1 209 us: every dY row crossed L2 -> LDS once per tap (4 x) in 64-byte pieces.  Same scheme as the forward above: the 36 weight fragments
// (128 rows x 4 taps x 64 channels with the 7 dead blocks left out = 36 KiB) in LDS, the 16 B fragments of a tile (4 taps x 4 channel steps)
// loaded straight into the MFMA operand registers one tile ahead, the 128 x 32 result staged as TWO contiguous 4-KiB runs (input rows 2a and
// 2a + 1, 64 pixels x 64 bytes each) and stored 1 KiB per wave instruction.
#define S2D_BSTEPS 16                      // B fragments of a tile: 4 taps x 2 channel steps of 32 x 2 pixel halves of 16
#define S2D_STAGE_ELEMS (2 * 64 * 32)      // per wave: [2 rows][64 px][32 ch] bf16 = 8 KiB

__global__ __launch_bounds__(512, 1) void conv3x3s2_c32_dgrad_kernel(const ConvGemmParams p, const int cblocks, const int64_t tiles)
{
    extern __shared__ __attribute__((aligned(16))) bf16_t s2c_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    bf16_t* const wl = s2c_lds;
    bf16_t* const stage = s2c_lds + S2C_W_ELEMS + wave * S2D_STAGE_ELEMS;

    // ---- weights -> LDS: fragment f = ((pair * 2 + channel step s2) * 2 + row half mh); `pair` enumerates the live (plane, tap) pairs in MFMA issue order:
    //   tap (0,0): planes 0 1 2 3;  tap (0,1): planes 1 3;  tap (1,0): planes 2 3;  tap (1,1): plane 3      (plane = 2 ph + pw, tap = 2 da + db)
    // lane l of a fragment = weight-image row (plane * 32 + 16 mh + l % 16), tap, channels [32 s2 + 8 (l / 16), + 8) of pack_s2d's [128][4][64] image
    for (int idx = tid; idx < S2C_WFRAG * 64; idx += 512) {
        const int f = idx >> 6, l = idx & 63;
        const int pair = f >> 2, s2 = (f >> 1) & 1, mh = f & 1;
        const int tap = pair < 4 ? 0 : (pair < 6 ? 1 : (pair < 8 ? 2 : 3));
        const int plane = pair < 4 ? pair : (pair < 6 ? 1 + 2 * (pair - 4) : (pair < 8 ? 2 + (pair - 6) : 3));
        const uint4 v = *reinterpret_cast<const uint4*>(p.W + ((int64_t)(plane * 32 + 16 * mh + (l & 15)) * 4 + tap) * p.Cin + s2 * 32 + (l >> 4) * 8);
        *reinterpret_cast<uint4*>(wl + (int64_t)idx * 8) = v;
    }
    __syncthreads();

    const int pn = lane & 15, kq = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * S2C_WAVES + wave, nw = (int64_t)gridDim.x * S2C_WAVES;
    const int64_t t0 = tiles * gw / nw, t1 = tiles * (gw + 1) / nw;
    const int64_t per_img = (int64_t)cblocks * p.OH;
    const bf16_t* const zp = p.zeros;
    struct Pos { int n, cb, oh; };
    auto advance = [&](Pos& q) {
        q.oh++;
        if (q.oh == p.OH) { q.oh = 0; q.cb++; if (q.cb == cblocks) { q.cb = 0; q.n++; } }
    };

    // B fragments of a tile: fragment ((tap * 2 + s2) * 2 + u), lane (pn, kq) reads channels [32 s2 + 8 kq, + 8) of dY pixel (a + da, 32 cb + 16 u + pn + db):
    // four lanes cover 64 contiguous bytes of the pixel, an instruction touches 16 pixels
    auto load_frag = [&](const Pos& q, int tap, int s2, int u) -> bf16x8 {      // (tap, s2, u are compile-time at every call site)
        const int da = tap >> 1, db = tap & 1;
        const int iy = q.oh + da;
        const int bcol = q.cb * 32 + u * 16 + pn;
        const int ix = bcol + db;
        const bool ok = iy < p.OH && bcol < p.OW && ix < p.OW;
        const bf16_t* src = ok ? p.A + (((int64_t)q.n * p.OH + iy) * p.OW + ix) * p.ldA + kq * 8 + s2 * 32 : zp;
        return *reinterpret_cast<const bf16x8*>(src);
    };
    auto load_tile = [&](const Pos& q, bf16x8 (&b)[S2D_BSTEPS]) {
#pragma unroll
        for (int tap = 0; tap < 4; tap++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int u = 0; u < 2; u++) b[(tap * 2 + s2) * 2 + u] = load_frag(q, tap, s2, u);
    };

    auto compute_tile = [&](const Pos& q, const bf16x8 (&b)[S2D_BSTEPS]) {
        f32x4 acc[2][4][2];                                                    // [pixel half u][plane][row half mh]: input channels 16 mh + 4 kq + i, dY pixel pn
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int mh = 0; mh < 2; mh++) acc[u][j][mh] = f32x4{0.f, 0.f, 0.f, 0.f};
        // pairs (plane, tap) in weight-fragment order: tap 0 -> pairs 0..3 (planes 0 1 2 3), tap 1 -> 4, 5 (planes 1, 3), tap 2 -> 6, 7 (planes 2, 3), tap 3 -> 8 (plane 3)
        constexpr int TAP_FIRST[5] = {0, 4, 6, 8, 9};
        constexpr int PAIR_PLANE[9] = {0, 1, 2, 3, 1, 3, 2, 3, 3};
#pragma unroll
        for (int tap = 0; tap < 4; tap++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const int bt = (tap * 2 + s2) * 2;
#pragma unroll
                for (int pair = TAP_FIRST[tap]; pair < TAP_FIRST[tap + 1]; pair++)
#pragma unroll
                    for (int mh = 0; mh < 2; mh++) {
                        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wl + (((pair * 2 + s2) * 2 + mh) * 64 + lane) * 8);
                        const int pln = PAIR_PLANE[pair];
                        acc[0][pln][mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b[bt], acc[0][pln][mh], 0, 0, 0);
                        acc[1][pln][mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b[bt + 1], acc[1][pln][mh], 0, 0, 0);
                    }
            }
        // ---- epilogue: stage [2 input rows ph][64 input pixels 2 (16 u + pn) + pw][32 ch]; 16-byte chunk c of pixel x sits at chunk c ^ ((x >> 1) & 3) ----
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int ph = j >> 1, pw = j & 1;
                const int x = 2 * (16 * u + pn) + pw;
#pragma unroll
                for (int mh = 0; mh < 2; mh++) {
                    const f32x4 a4 = acc[u][j][mh];
                    const int c0 = 16 * mh + 4 * kq;
                    *reinterpret_cast<uint2*>(stage + (ph * 64 + x) * 32 + (((c0 >> 3) ^ (pn & 3)) << 3) + (c0 & 4)) =
                        make_uint2(pack_bf2(a4[0], a4[1]), pack_bf2(a4[2], a4[3]));
                }
            }
        // store: per input row one contiguous run of 64 pixels x 64 bytes (channel stride ldC: contiguous when the tensor is not a slice).
        // No branch around the stores of a row (a CFG join in front of in-flight loads makes hipcc drain vmcnt(0)): the row test is part of the
        // per-lane store predicate.
        const int c4 = lane & 3, x4 = lane >> 2;                                // chunk c4 of pixels x4 + 16 it
        const int xlive = p.OWf - q.cb * 64;                                     // input pixels of this block inside the image
#pragma unroll
        for (int ph = 0; ph < 2; ph++) {
            const int iy = 2 * q.oh + ph;
            const int lim = iy < p.OHf ? xlive : 0;
            bf16_t* const orow = reinterpret_cast<bf16_t*>(p.out) + (((int64_t)q.n * p.OHf + iy) * p.OWf + q.cb * 64) * p.ldC + c4 * 8;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int x = it * 16 + x4;
                const uint4 v = *reinterpret_cast<const uint4*>(stage + (ph * 64 + x) * 32 + ((c4 ^ ((x >> 1) & 3)) << 3));
                if (x < lim) *reinterpret_cast<uint4*>(orow + (int64_t)x * p.ldC) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    bf16x8 ba[S2D_BSTEPS], bb[S2D_BSTEPS];
    if (t0 < t1) {
        Pos head;
        head.n = (int)(t0 / per_img);
        const int rem0 = (int)(t0 - (int64_t)head.n * per_img);
        head.cb = rem0 / p.OH;
        head.oh = rem0 - head.cb * p.OH;
        int64_t th = t0;
        auto next = [&]() { if (th + 1 < t1) { advance(head); th++; } return head; };
        Pos pa = head;
        load_tile(pa, ba);
        for (int64_t t = t0; t < t1; t += 2) {
            const Pos pb = next();
            load_tile(pb, bb);                                                 // tile t + 1: in flight under the MFMAs and the epilogue of tile t
            compute_tile(pa, ba);
            if (t + 1 >= t1) break;
            pa = next();
            load_tile(pa, ba);                                                 // tile t + 2 under tile t + 1
            compute_tile(pb, bb);
        }
    }
}

static int s2c_mode()
{
    static const int v = [] { const char* e = getenv("RYOLO_S2C32"); return e ? atoi(e) : 1; }();       // A/B knob: 0 = the generic 256 x 64 tile
    return v;
}

bool s2c_geometry(const ConvGemmParams& p, S2cGeom& g)
{
    g = S2cGeom{};
    if (!s2c_mode() || (p.pipe & 0xff) != 1) return false;
    if (p.nclasses != 1 || p.Cin != 32 || p.wtaps != 9 || p.sh != 2 || p.sw != 2 || p.Nout > 64 || p.Nout % 8) return false;
    const TapClass& tc = p.cls[0];
    if (tc.ntaps != 9 || tc.oh_add || tc.ow_add) return false;
    for (int i = 0; i < 9; i++)
        if (tc.dh[i] != i / 3 - 1 || tc.dw[i] != i % 3 - 1 || tc.widx[i] != i) return false;
    if (p.oh_mul != 1 || p.ow_mul != 1 || p.OHf != p.OH || p.OWf != p.OW) return false;
    if (p.OH != (p.IH - 1) / 2 + 1 || p.OW != (p.IW - 1) / 2 + 1) return false;                        // 3x3, stride 2, padding 1
    if (p.pool_idx || p.s2d_cin || p.head_attrs || !p.zeros) return false;
    if (p.epi != EPI_RAW && p.epi != EPI_STATS && p.epi != EPI_AFFINE_ACT) return false;
    if (p.ldA % 8 || p.ldC % 8) return false;
    if (((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.out)) & 15)) return false;
    g.cblocks = (p.OW + 31) / 32;
    g.tiles = (int64_t)p.NB * g.cblocks * p.OH;
    if (g.tiles <= 0 || (int64_t)p.NB * p.IH * p.IW >= (1ll << 31)) return false;
    // a persistent workgroup pays for its 36-KiB weight tile and its two-tile prologue: below ~4 tiles per wave the generic kernel's small tiles win
    // (one 800 x 800 image, 5 200 tiles: 22.8 vs 21.1 us).  pipe bit 0x400 (the tests' "also on small grids" bit) overrides.
    if (g.tiles < 8192 && !(p.pipe & 0x400)) return false;
    // one persistent 8-wave workgroup per CU; small problems get as many workgroups as there are groups of 8 tiles
    const int64_t want = (g.tiles + S2C_WAVES - 1) / S2C_WAVES;
    g.nwg = (int)(want < 256 ? want : 256);
    g.lds_bytes = (unsigned)(S2C_W_ELEMS * 2 + S2C_WAVES * S2C_STAGE_ELEMS * 2 + S2C_WAVES * 128 * 4);
    g.ok = 1;
    return true;
}

template <int EPI> static int s2c_launch_t(const ConvGemmParams& p, const S2cGeom& g, hipStream_t stream)
{
    static RyLdsAttr attr;
    if (ry_max_dynamic_lds(attr, reinterpret_cast<const void*>(&conv3x3s2_c32_kernel<EPI>), 160 * 1024)) return RY_ERR_LAUNCH;
    hipLaunchKernelGGL((conv3x3s2_c32_kernel<EPI>), dim3((unsigned)g.nwg), dim3(512), g.lds_bytes, stream, p, g.cblocks, g.tiles);
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}

int s2c_launch(const ConvGemmParams& p, const S2cGeom& g, hipStream_t stream)
{
    switch (p.epi) {
        case EPI_RAW: return s2c_launch_t<EPI_RAW>(p, g, stream);
        case EPI_STATS: return s2c_launch_t<EPI_STATS>(p, g, stream);
        case EPI_AFFINE_ACT: return s2c_launch_t<EPI_AFFINE_ACT>(p, g, stream);
    }
    return RY_ERR_ARG;
}

// ---- the data gradient (space-to-depth form) ---------------------------------------------------------------------------------------
bool s2c_dgrad_geometry(const ConvGemmParams& p, S2cGeom& g)
{
    g = S2cGeom{};
    static const int on = [] { const char* e = getenv("RYOLO_S2C32_DGRAD"); return e ? atoi(e) : 1; }();   // A/B knob: 0 = the persistent pointwise kernel's S2D form
    if (!on || (p.pipe & 0xff) != 1 || !p.zeros) return false;
    if (p.s2d_cin != 32 || p.Nout != 128 || p.Cin != 64 || p.wtaps != 4 || p.nclasses != 1 || p.oh_mul != 2 || p.ow_mul != 2) return false;
    const TapClass& tc = p.cls[0];
    if (tc.ntaps != 4 || tc.oh_add || tc.ow_add) return false;
    for (int i = 0; i < 4; i++)
        if (tc.dh[i] != (i >> 1) || tc.dw[i] != (i & 1) || tc.widx[i] != i) return false;
    if (p.sh != 1 || p.sw != 1 || p.IH != p.OH || p.IW != p.OW) return false;
    if (p.OHf > 2 * p.OH || p.OHf < 2 * p.OH - 1 || p.OWf > 2 * p.OW || p.OWf < 2 * p.OW - 1) return false;
    // plain stores only: an accumulate epilogue would have to wait for its own read-back behind the next tile's prefetch (the gradient of this
    // layer's input has one writer in every plan of the three backbones; an accumulating caller stays on the persistent pointwise kernel's S2D form)
    if (p.pool_idx || p.head_attrs || p.epi != EPI_RAW || p.ldA % 8 || p.ldC % 8) return false;
    if (((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.out)) & 15)) return false;
    g.cblocks = (p.OW + 31) / 32;
    g.tiles = (int64_t)p.NB * g.cblocks * p.OH;
    if (g.tiles <= 0 || (int64_t)p.NB * p.OHf * p.OWf >= (1ll << 31)) return false;
    if (g.tiles < 8192 && !(p.pipe & 0x400)) return false;
    const int64_t want = (g.tiles + S2C_WAVES - 1) / S2C_WAVES;
    g.nwg = (int)(want < 256 ? want : 256);
    g.lds_bytes = (unsigned)(S2C_W_ELEMS * 2 + S2C_WAVES * S2D_STAGE_ELEMS * 2);
    g.ok = 1;
    return true;
}

int s2c_dgrad_launch(const ConvGemmParams& p, const S2cGeom& g, hipStream_t stream)
{
    static RyLdsAttr attr;
    if (ry_max_dynamic_lds(attr, reinterpret_cast<const void*>(&conv3x3s2_c32_dgrad_kernel), 160 * 1024)) return RY_ERR_LAUNCH;
    hipLaunchKernelGGL(conv3x3s2_c32_dgrad_kernel, dim3((unsigned)g.nwg), dim3(512), g.lds_bytes, stream, p, g.cblocks, g.tiles);
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}
