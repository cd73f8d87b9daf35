// First layer of the backbones: 3x3 stride-1 convolution of the fp32 NCHW image, Cin = 3 -> 32 channels
// (model/backbone.py:9,60 `Conv(3, 32, 3, 1)`), forward and weight gradient, DIRECT from the image.
//
// The generic route lowers this layer to an explicit im2col ([pixels][32] bf16, k = (r*3+s)*3+c padded 27 -> 32) followed by
// a single-tap GEMM: at 800^2 x batch 64 that is 41 M pixels, a 2.6 GB im2col write, a 2.6 GB read back by the GEMM and
// another 2.6 GB read by the weight-gradient GEMM — the layer is pure HBM traffic (K = 27), and the col tensor triples it.
// Here both directions gather their 27 taps straight from the image (0.5 GB, L1/L2 resident per tile):
//   forward   one wave = 32 consecutive pixels: the im2col row of a pixel is built in registers (16 scalar loads per lane, one
//             lane = one pixel x 8 k-values of each K16 half), 2 MFMA 32x32x16 against the weight fragments held in registers;
//             A = weights, B = pixels, so a lane ends with 4 consecutive channels of its pixel -> 8-byte stores; BatchNorm
//             statistics accumulate in registers over the wave's tiles and leave as one partial row per workgroup;
//   wgrad     K = pixels.  One wave = runs of 16 pixels of one image row: dY^T through a wave-private 1-KiB LDS-DMA piece +
//             transposed reads (ds_read_b64_tr_b16), col^T needs 8 CONSECUTIVE pixels of one (tap, channel) per lane = 8
//             consecutive floats of an image row: plain loads, no LDS.  Per-workgroup slab -> small deterministic fold.
// Both are memory streams (0.5 + 2.6 GB each); nothing here is worth an LDS tile.
#include "conv_internal.h"
#include <type_traits>
#include <stdlib.h>

#define STEM_BWD_SLAB 2112                 // floats per slab of the fused backward: G 1024 + XX 1024 + S0 32 + S1 32

// k -> (channel, row tap, column tap) of the im2col ordering k = (r*3 + s)*3 + c
__device__ __forceinline__ void stem_k(int k, int& c, int& r, int& s)
{
    const int tap = k / 3;
    c = k - tap * 3;
    r = tap / 3;
    s = tap - r * 3;
}

__global__ __launch_bounds__(256) void stem3x3_fwd_kernel(const StemParams p)
{
    __shared__ float red[4][2][32];
    __shared__ __attribute__((aligned(16))) bf16_t otile[4][32][40];   // per wave: 32 pixels x 32 channels (+8 pad), 80-byte rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px_l = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const int64_t M = (int64_t)p.NB * H * W;
    const int64_t ntiles = (M + 31) >> 5;
    // this lane's 16 k values: q*16 + h*8 + e  -> offsets relative to (n, c=0, oh, ow); validity is a 16-bit mask per tile built
    // from four per-lane constants: which of the 16 k use the top / bottom row tap and the left / right column tap
    int koff[16];
    unsigned kval = 0, m_top = 0, m_bot = 0, m_left = 0, m_right = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int k = (i >> 3) * 16 + h * 8 + (i & 7);
        int c, r, s;
        stem_k(k < 27 ? k : 0, c, r, s);
        koff[i] = (c * H + (r - 1)) * W + (s - 1);
        if (k < 27) {
            kval |= 1u << i;
            if (r == 0) m_top |= 1u << i;
            if (r == 2) m_bot |= 1u << i;
            if (s == 0) m_left |= 1u << i;
            if (s == 2) m_right |= 1u << i;
        }
    }
    // weight fragments (A operand: row = output channel, k = h*8 + e): loaded once
    bf16x8 wfrag[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        uint4 w = make_uint4(0, 0, 0, 0);
        if (px_l < p.Cout) w = *reinterpret_cast<const uint4*>(p.wf + px_l * 32 + q * 16 + h * 8);
        wfrag[q] = __builtin_bit_cast(bf16x8, w);
    }
    float ssum[16], ssq[16];
#pragma unroll
    for (int e = 0; e < 16; e++) { ssum[e] = 0.f; ssq[e] = 0.f; }

    const int HW = H * W;
    const bool aligned = (W & 31) == 0;                          // a 32-pixel tile never leaves its image row: (n, oh, ow0) are wave-uniform
    for (int64_t tt = (int64_t)blockIdx.x * 4 + wave; tt < ntiles; tt += (int64_t)gridDim.x * 4) {
        const int64_t pix = tt * 32 + px_l;
        const bool live = pix < M;
        int n, oh, ow;
        if (aligned) {
            const int t0 = __builtin_amdgcn_readfirstlane((int)(tt * 32 < M ? tt * 32 : 0));      // scalar divisions, once per tile
            n = t0 / HW;
            const int rem = t0 - n * HW;
            oh = rem / W;
            ow = rem - oh * W + px_l;
        } else {
            const int pp = (int)(live ? pix : 0);                 // M < 2^31 checked on the host
            n = pp / HW;
            const int rem = pp - n * HW;
            oh = rem / W;
            ow = rem - oh * W;
        }
        const int base = (n * 3 * H + oh) * W + ow;
        unsigned okm = live ? kval : 0u;
        if (oh < 1) okm &= ~m_top;
        if (oh + 1 >= H) okm &= ~m_bot;
        if (ow < 1) okm &= ~m_left;
        if (ow + 1 >= W) okm &= ~m_right;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int i = q * 8 + e;
                const bool ok = (okm >> i) & 1u;
                const float x = p.img[ok ? base + koff[i] : 0];   // unconditional load from a safe address, zeroed below
                v[e] = ok ? x : 0.f;
            }
            const uint4 b = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[q], __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        }
        // acc: column = this lane's pixel, rows = channels (e&3) + 8*(e>>2) + 4*h
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const int c0 = 8 * g4 + 4 * h;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = acc[4 * g4 + q];
            if (p.epi == EPI_AFFINE_ACT_R) {
                // training: the raw output as the two-pass path would have stored it (bf16), then BatchNorm + activation on THAT
                const uint2 r = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
                v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
            }
            if ((p.epi == EPI_AFFINE_ACT || p.epi == EPI_AFFINE_ACT_R) && c0 < p.Cout) {
                const float sc4[4] = {p.scale[c0], p.scale[c0 + 1], p.scale[c0 + 2], p.scale[c0 + 3]};
                const float sf4[4] = {p.shift[c0], p.shift[c0 + 1], p.shift[c0 + 2], p.shift[c0 + 3]};
                act_affine_quad(v, sc4, sf4, p.act);
            }
            const uint2 w = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            if (p.out) *reinterpret_cast<uint2*>(&otile[wave][px_l][c0]) = w;  // staged: the lane's 8-byte piece of its pixel row
            if (p.epi == EPI_STATS && live) {
                const float f0 = __uint_as_float(w.x << 16), f1 = __uint_as_float(w.x & 0xffff0000u);
                const float f2 = __uint_as_float(w.y << 16), f3 = __uint_as_float(w.y & 0xffff0000u);
                ssum[4 * g4 + 0] += f0; ssq[4 * g4 + 0] += f0 * f0;
                ssum[4 * g4 + 1] += f1; ssq[4 * g4 + 1] += f1 * f1;
                ssum[4 * g4 + 2] += f2; ssq[4 * g4 + 2] += f2 * f2;
                ssum[4 * g4 + 3] += f3; ssq[4 * g4 + 3] += f3 * f3;
            }
        }
        // whole pixel rows to HBM: lane -> (pixel = lane >> 2 (+16), 16-byte slot = lane & 3); one instruction = 16 rows x 64 B.
        // (direct 8-byte stores from the MFMA layout touched 32 rows per instruction, 16 B each: request-rate bound)
        if (p.out) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int pr = half * 16 + (lane >> 2), sl = lane & 3;
                const int64_t opix = tt * 32 + pr;
                const uint4 o = *reinterpret_cast<const uint4*>(&otile[wave][pr][sl * 8]);
                if (opix < M && sl * 8 < p.Cout) *reinterpret_cast<uint4*>(p.out + opix * p.ldC + sl * 8) = o;
            }
        }
    }
    if (p.epi == EPI_STATS) {
        // fold the 32 pixel lanes of each half-wave, then the 4 waves; one partial-statistics row per workgroup
#pragma unroll
        for (int e = 0; e < 16; e++) {
            float a = ssum[e], b = ssq[e];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                a += __shfl_xor(a, o, 64);
                b += __shfl_xor(b, o, 64);
            }
            if (px_l == 0) {
                const int c = (e & 3) + 8 * (e >> 2) + 4 * h;
                red[wave][0][c] = a;
                red[wave][1][c] = b;
            }
        }
        __syncthreads();
        if (tid < 2 * 32) {
            const int c = tid & 31, which = tid >> 5;
            if (c < p.Cout) p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- LDS patch scheme
// (W % 32 == 0: a 32-pixel tile never leaves its image row.)  The gather above keeps 16 loaded floats per lane alive per tile and
// the kernel runs at the latency of one tile per wave (128+ VGPRs -> 2-4 waves per SIMD, ~2.7 TB/s).  Here the 3 channels x 3 rows x
// 40 columns around a tile (90 16-byte pieces) arrive by LDS-DMA — two instructions per wave, no VGPRs, zero page for the image border —
// into a wave-private double buffer, one tile AHEAD of the one being computed (nothing register-resident crosses the loop back
// edge, so hipcc's waitcnt pass leaves the counted vmcnt alone).  MFMA operands are then built from LDS with compile-time offsets.
__device__ __attribute__((aligned(16))) float stem_zero_page[64];                               // DMA source of out-of-image elements

#define SP_ROW 40                                                  // floats per (channel, row): columns ow0 - 4 ... ow0 + 35
#define SP_PIECES 90                                               // 9 (channel, row) lines x ten 16-byte pieces
#define SP_BYTES 2048                                              // two 1-KiB DMA instructions (the second one 26 lanes wide)

// 16-byte pieces: ow0 % 32 == 0 and W % 32 == 0 make every piece of a line 16-byte aligned in the image, and a piece is either
// completely inside or completely outside the image (first / last piece of a line at the left / right border, whole lines at the
// top / bottom) -> outside pieces come from a zero page.  (The first version moved dwords: five instructions of 256 bytes per tile;
// an LDS-DMA instruction costs 100-185 issue cycles whatever its width.)
struct StemPatchLane { int off[2]; unsigned flg[2]; };             // per lane: piece j = 64 u + lane of the patch

__device__ __forceinline__ void stem_patch_lane(StemPatchLane& L, int lane, int H, int W)
{
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int j = 64 * u + lane;
        const int line = j / 10, pc = j - line * 10;              // line = c*3 + r
        const int c = line / 3, r = line - c * 3;
        L.off[u] = (c * H + (r - 1)) * W + 4 * pc - 4;
        L.flg[u] = (j < SP_PIECES ? 1u : 0u) | (r == 0 ? 2u : 0u) | (r == 2 ? 4u : 0u) | (pc == 0 ? 8u : 0u) | (pc == 9 ? 16u : 0u);
    }
}

// issue the patch of the tile at (n, oh, ow0): base = (n*3*H + oh)*W + ow0
__device__ __forceinline__ void stem_patch_issue(const StemPatchLane& L, const float* __restrict__ img, int base, int oh, int ow0, int H, int W,
                                                 unsigned char* lds)
{
    unsigned bad = 0;                                              // flag bits that make a piece fall outside the image
    if (oh < 1) bad |= 2u;
    if (oh + 1 >= H) bad |= 4u;
    if (ow0 < 1) bad |= 8u;
    if (ow0 + 32 >= W) bad |= 16u;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const bool ok = (L.flg[u] & 1u) && !(L.flg[u] & bad);
        const float* src = ok ? img + base + L.off[u] : stem_zero_page;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(lds + u * 1024), 16, 0, 0);
    }
}

// patch index of im2col position k (= (r*3 + s)*3 + c) for the pixel at column 0 of the tile: (c*3 + r)*40 + s + 3
__host__ __device__ constexpr int sp_kidx(int k) { return ((k % 3) * 3 + (k / 3) / 3) * SP_ROW + (k / 3) % 3 + 3; }

// K order of the 27-tap contraction.  The order of a GEMM's K dimension is free as long as both operands agree, so it is chosen
// for the LDS reads: lane (pixel l31, half h) owns 16 slots whose patch offsets are COMPILE-TIME constants relative to two
// per-lane bases (no address VALU per read — the first version spent ~60 VALU instructions per tile on selects and adds):
//   slots 0-8   channel h,  (r, s) = (j / 3, j % 3)            base A = l31 + 120 h  (one channel = 3 lines of 40 floats)
//   slots 9-14  channel 2,  r = (j - 9) / 3 + h, s = (j - 9) % 3   base B = l31 + 40 h   (h = 1 starts one line lower; its r = 1
//               slots duplicate h = 0's and carry ZERO weights)
//   slot 15     unused (zero weight)
__host__ __device__ constexpr int sp_slot_imm(int j) { return j < 9 ? (j / 3) * SP_ROW + j % 3 + 3 : (6 + (j - 9) / 3) * SP_ROW + (j - 9) % 3 + 3; }
// im2col index k of slot j for half h, -1: zero weight
__host__ __device__ constexpr int sp_slot_k(int h, int j)
{
    return j < 9 ? ((j / 3) * 3 + j % 3) * 3 + h
                 : (j < 15 ? ((h && (j - 9) / 3 == 0) ? -1 : ((((j - 9) / 3 + h) * 3 + (j - 9) % 3) * 3 + 2)) : -1);
}

// weight fragments in slot order for output channel `co` (row / column l31 of the operand), half h
__device__ __forceinline__ void stem_slot_weights(const bf16_t* __restrict__ wf, int co, int h, bool live, bf16x8& w0, bf16x8& w1)
{
    unsigned short v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int k0 = sp_slot_k(0, j), k1 = sp_slot_k(1, j);
        const int k = h ? k1 : k0;
        v[j] = (live && k >= 0) ? wf[co * 32 + (k >= 0 ? k : 0)] : (unsigned short)0;
    }
    w0 = __builtin_bit_cast(bf16x8, make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)));
    w1 = __builtin_bit_cast(bf16x8, make_uint4(v[8] | (v[9] << 16), v[10] | (v[11] << 16), v[12] | (v[13] << 16), v[14] | (v[15] << 16)));
}

// the 16 patch values of lane (pixel l31, half h) in slot order, packed as the two K16 operands; pa / pb = LDS byte address of the
// patch + base A / base B.  Inline-asm reads with their own lgkmcnt wait: next to the LDS staging stores of the forward kernel hipcc
// otherwise drains vmcnt(0) — the prefetch of the NEXT tile — in front of the first read of every tile (conv_internal.h).
template <int J> __device__ __forceinline__ float sp_read(unsigned pa, unsigned pb)
{
    float r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(J < 9 ? pa : pb), "n"(sp_slot_imm(J) * 4) : "memory");
    return r;
}
__device__ __forceinline__ void stem_patch_pixel_frags(unsigned pa, unsigned pb, bf16x8& f0, bf16x8& f1)
{
    float v0 = sp_read<0>(pa, pb), v1 = sp_read<1>(pa, pb), v2 = sp_read<2>(pa, pb), v3 = sp_read<3>(pa, pb), v4 = sp_read<4>(pa, pb);
    float v5 = sp_read<5>(pa, pb), v6 = sp_read<6>(pa, pb), v7 = sp_read<7>(pa, pb), v8 = sp_read<8>(pa, pb), v9 = sp_read<9>(pa, pb);
    float v10 = sp_read<10>(pa, pb), v11 = sp_read<11>(pa, pb), v12 = sp_read<12>(pa, pb), v13 = sp_read<13>(pa, pb), v14 = sp_read<14>(pa, pb);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9), "+v"(v10), "+v"(v11),
                   "+v"(v12), "+v"(v13), "+v"(v14));
    f0 = __builtin_bit_cast(bf16x8, make_uint4(pack_bf2(v0, v1), pack_bf2(v2, v3), pack_bf2(v4, v5), pack_bf2(v6, v7)));
    f1 = __builtin_bit_cast(bf16x8, make_uint4(pack_bf2(v8, v9), pack_bf2(v10, v11), pack_bf2(v12, v13), pack_bf2(v14, 0.f)));
}

// Tile order: every wave owns a CONTIGUOUS range of tiles in column-major order (down one 32-pixel column block of an image, then the
// next block), so a tile's three image rows are the previous tile's last two plus one new row: the patch is re-read from L1 / L2
// instead of HBM (a raster walk with a grid-sized stride fetched every image row three times: 2.2 GB for the 0.5 GB image, PMC).
struct StemWalk {
    int n, oh, ow;
    __device__ __forceinline__ void init(int64_t u, int H, int W)
    {
        const int64_t per = (int64_t)H * (W >> 5);
        n = (int)(u / per);
        const int r = (int)(u - (int64_t)n * per);
        const int cb = r / H;
        oh = r - cb * H;
        ow = cb << 5;
    }
    __device__ __forceinline__ void step(int H, int W)
    {
        if (++oh == H) { oh = 0; ow += 32; if (ow == W) { ow = 0; n++; } }
    }
};

template <int EPI>
__global__ __launch_bounds__(256) void stem3x3_fwd_lds_kernel(const StemParams p)
{
    // ring of NB tile buffers per wave, prefetch distance D.  Measured at 64 x 800^2: D = 1 -> 0.41 / 0.98 ms (statistics / fused pass),
    // D = 3 -> 0.45 / 1.19 ms: the loop is not waiting for its DMA, it runs at its instruction rate (~100 VALU + ~100 SALU per tile:
    // 16 waves per CU retire one tile per ~190 cycles), so one tile ahead is enough and the smaller footprint wins
    constexpr int NB = 2, D = NB - 1;
    __shared__ __attribute__((aligned(1024))) unsigned char patch[4][NB][SP_BYTES];
    __shared__ float red[4][2][32];
    __shared__ __attribute__((aligned(16))) bf16_t otile[4][32][40];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px_l = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W, HW = H * W;
    const int64_t M = (int64_t)p.NB * HW;
    const int64_t ntiles = M >> 5;
    StemPatchLane L;
    stem_patch_lane(L, lane, H, W);
    bf16x8 wfrag[2];
    stem_slot_weights(p.wf, px_l, h, px_l < p.Cout, wfrag[0], wfrag[1]);
    constexpr bool affine = EPI == EPI_AFFINE_ACT || EPI == EPI_AFFINE_ACT_R;
    float sc[16], sh[16];                                          // this lane's 16 channels: c = 8*g4 + 4*h + q
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int c = 8 * (e >> 2) + 4 * h + (e & 3);
        sc[e] = affine && c < p.Cout ? p.scale[c] : 1.f;
        sh[e] = affine && c < p.Cout ? p.shift[c] : 0.f;
    }
    float ssum[16], ssq[16];
#pragma unroll
    for (int e = 0; e < 16; e++) { ssum[e] = 0.f; ssq[e] = 0.f; }

    const int64_t nwaves = (int64_t)gridDim.x * 4, per_wave = (ntiles + nwaves - 1) / nwaves;
    const int64_t u0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave)) * per_wave;
    const int64_t u1 = u0 + per_wave < ntiles ? u0 + per_wave : ntiles;
    StemWalk nx, cur;                                              // nx: the tile whose patch is issued next; cur: the tile being computed
    nx.init(u0 < ntiles ? u0 : 0, H, W);
    cur = nx;
    auto issue = [&](int buf) { stem_patch_issue(L, p.img, (nx.n * 3 * H + nx.oh) * W + nx.ow, nx.oh, nx.ow, H, W, &patch[wave][buf][0]); };
    // prologue: tiles u0 .. u0 + D - 1 (always 2 DMA instructions each: past the end of the range they re-read its last tile, so the
    // counted wait below holds for every iteration)
#pragma unroll
    for (int d = 0; d < D; d++) {
        issue(d);
        if (u0 + d + 1 < u1) nx.step(H, W);
    }
    const int offA = (px_l + 120 * h) * 4, offB = (px_l + 40 * h) * 4;
    int buf = 0;
    for (int64_t tt = u0; tt < u1; tt++) {
        issue(buf == 0 ? D : buf - 1);                             // tile tt + D into the buffer tile tt - 1 just left
        if (tt + D + 1 < u1) nx.step(H, W);
        // everything but the D newest tiles' pieces has landed.  (The stores of earlier iterations sit between them in issue order;
        // counting only the 2 D newer loads is exact without stores and conservative with them, whatever order stores retire in.)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * D) : "memory");
        bf16x8 f0, f1;
        const unsigned pbase = lds_addr(&patch[wave][buf][0]);
        stem_patch_pixel_frags(pbase + (unsigned)offA, pbase + (unsigned)offB, f0, f1);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[0], f0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[1], f1, acc, 0, 0, 0);
        // r06: BatchNorm + activation of the tile's 16 values in ONE walk of the (wave-uniform) activation chain with branch-free arms
        // (act_affine_vec, conv_internal.h): per element the chain cost three scalar compare + branch pairs in an instruction-rate-bound kernel
        float v16[16];
#pragma unroll
        for (int e = 0; e < 16; e++) v16[e] = acc[e];
        if (EPI == EPI_AFFINE_ACT_R) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const unsigned r = pack_bf2(v16[e], v16[e + 1]);
                v16[e] = __uint_as_float(r << 16);
                v16[e + 1] = __uint_as_float(r & 0xffff0000u);
            }
        }
        if (affine) act_affine_vec<16>(v16, sc, sh, p.act);
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const int c0 = 8 * g4 + 4 * h;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = v16[4 * g4 + q];
            const uint2 w = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            if (p.out) *reinterpret_cast<uint2*>(&otile[wave][px_l][c0]) = w;
            if (EPI == EPI_STATS) {
                const float f0s = __uint_as_float(w.x << 16), f1s = __uint_as_float(w.x & 0xffff0000u);
                const float f2s = __uint_as_float(w.y << 16), f3s = __uint_as_float(w.y & 0xffff0000u);
                ssum[4 * g4 + 0] += f0s; ssq[4 * g4 + 0] += f0s * f0s;
                ssum[4 * g4 + 1] += f1s; ssq[4 * g4 + 1] += f1s * f1s;
                ssum[4 * g4 + 2] += f2s; ssq[4 * g4 + 2] += f2s * f2s;
                ssum[4 * g4 + 3] += f3s; ssq[4 * g4 + 3] += f3s * f3s;
            }
        }
        if (p.out) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int pr = half * 16 + (lane >> 2), sl = lane & 3;
                const int64_t opix = ((int64_t)cur.n * H + cur.oh) * W + cur.ow + pr;
                const uint4 o = *reinterpret_cast<const uint4*>(&otile[wave][pr][sl * 8]);
                if (sl * 8 < p.Cout) *reinterpret_cast<uint4*>(p.out + opix * p.ldC + sl * 8) = o;
            }
        }
        cur.step(H, W);
        buf = buf == D ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (EPI == EPI_STATS) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            float a = ssum[e], b = ssq[e];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                a += __shfl_xor(a, o, 64);
                b += __shfl_xor(b, o, 64);
            }
            if (px_l == 0) {
                const int c = (e & 3) + 8 * (e >> 2) + 4 * h;
                red[wave][0][c] = a;
                red[wave][1][c] = b;
            }
        }
        __syncthreads();
        if (tid < 2 * 32) {
            const int c = tid & 31, which = tid >> 5;
            if (c < p.Cout) p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- weight gradient
template <bool FUSE>
__global__ __launch_bounds__(256) void stem3x3_wgrad_kernel(const StemWgradParams p, float* __restrict__ slabs)
{
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    __shared__ __attribute__((aligned(1024))) unsigned char dyt[4][4][1024];     // wave-private: 4 pieces of [16 px][32 ch] bf16
    __shared__ __attribute__((aligned(1024))) unsigned char yt[FUSE ? 4 : 1][FUSE ? 4 : 1][FUSE ? 1024 : 16];   // FUSE: the raw conv output, same tiles
    __shared__ float redw[4][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W;
    const int HW = H * W;
    const int64_t M = (int64_t)p.NB * HW;
    const int64_t nsteps = M >> 4;                                 // W % 16 == 0: every 16-pixel run lies in one image row
    const int h = lane >> 5, kk = lane & 31;
    int c, r, s;
    stem_k(kk < 27 ? kk : 0, c, r, s);
    const bool kok = kk < 27;
    const int koff = (c * H + (r - 1)) * W + (s - 1) + h * 8;      // this lane's 8 consecutive pixels start at ow0 + h*8 (+ s - 1)
    // transposed-read addressing of the [16 px][32 ch] tile (see conv.hip / conv3x3.hip)
    const int s16 = lane & 15, grp = lane >> 4;
    const int fr_off = ((grp >> 1) * 8 + (s16 >> 2)) * 64 + (16 * (grp & 1) + 4 * (s16 & 3)) * 2;
    const int d_row = lane >> 2, d_slot = lane & 3;                // DMA lane -> (pixel row, 16-byte slot)
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    // FUSE: the A fragment of a lane is 8 pixels of ONE output channel (lane & 31): four per-channel constants, as in bn_act_bwd_apply
    float bn_sc = 0.f, bn_sh = 0.f, bn_A = 0.f, bn_B = 0.f;
    if constexpr (FUSE) {
        const int ch = lane & 31, C = p.Cout;
        const float mu = p.co[ch], is = p.co[C + ch], mg = p.bco[ch], mx = p.bco[C + ch];
        bn_sc = p.co[2 * C + ch];
        bn_sh = p.co[3 * C + ch];
        bn_A = -bn_sc * is * mx;
        bn_B = bn_sc * (is * mx * mu - mg);
    }

    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t ss = (int64_t)blockIdx.x * 4 + wave;
    // Four 16-pixel steps per iteration: their 4 dY pieces (LDS-DMA) and 4 x 8 image values (registers) are all issued up front and
    // waited for ONCE, so a memory round trip is paid per 4 MFMAs instead of per MFMA; occupancy (8 waves / SIMD) overlaps the rest.
    // (A one-step-ahead software pipeline was defeated by hipcc's waitcnt pass: across the loop back-edge it drains vmcnt to 0.)
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const int64_t img_elems = (int64_t)p.NB * 3 * HW;
    constexpr int UNR = 4;
    // (n, oh, ow0) of the wave's current step, advanced by `stride` steps with carries (no division in the loop)
    int cn, coh, cow;
    {
        const int64_t p0 = (ss < nsteps ? ss : 0) * 16;
        cn = (int)(p0 / HW);
        const int rem = (int)(p0 - (int64_t)cn * HW);
        coh = rem / W;
        cow = rem - coh * W;
    }
    const int64_t adv = stride * 16;                               // pixels per advance
    const int adv_n = (int)(adv / HW), adv_r = (int)(adv - (int64_t)adv_n * HW);
    const int adv_h = adv_r / W, adv_w = adv_r - adv_h * W;
    for (; ss < nsteps; ss += UNR * stride) {
        float v[UNR][8];
        unsigned okm[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int64_t st = ss + u * stride;
            const bool live = st < nsteps;
            const bf16_t* src = live ? p.dY + (st * 16 + d_row) * (int64_t)p.ldY + d_slot * 8 : p.dY;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(&dyt[wave][u][0]), 16, 0, 0);
            if constexpr (FUSE) {
                const bf16_t* ysrc = live ? p.y + (st * 16 + d_row) * (int64_t)p.ldy + d_slot * 8 : p.y;
                __builtin_amdgcn_global_load_lds((gbl_void_t*)ysrc, (lds_void_t*)(&yt[wave][u][0]), 16, 0, 0);
            }
            const int base = (cn * 3 * H + coh) * W + cow;
            const bool rowok = live && kok && (unsigned)(coh + r - 1) < (unsigned)H;
            const int iw0 = cow + h * 8 + s - 1;                   // column of this lane's first element
            // only the first / last element of a row run can fall outside: two compares instead of eight
            unsigned m = rowok ? 0xffu : 0u;
            if (iw0 < 0) m &= ~1u;
            if (iw0 + 7 >= W) m &= ~0x80u;
            // 8 consecutive floats of one image row: two unaligned 16-byte loads (one request per 16 B instead of per 4 B — the
            // scalar version was bound by the number of cache lines touched per instruction); the run may start one float before
            // the row (left tap) or end one after it (right tap): only at the very ends of the whole image buffer is that outside
            // the allocation, there the lane falls back to masked scalar loads
            const int64_t off = (int64_t)base + koff;
            if (m && off >= 0 && off + 8 <= img_elems) {
                const f4u lo4 = *reinterpret_cast<const f4u*>(p.img + off), hi4 = *reinterpret_cast<const f4u*>(p.img + off + 4);
                v[u][0] = lo4.x; v[u][1] = lo4.y; v[u][2] = lo4.z; v[u][3] = lo4.w;
                v[u][4] = hi4.x; v[u][5] = hi4.y; v[u][6] = hi4.z; v[u][7] = hi4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) v[u][e] = p.img[(m >> e) & 1u ? off + e : 0];
            }
            okm[u] = m;
            // advance to the wave's next step
            cow += adv_w;
            coh += adv_h;
            cn += adv_n;
            if (cow >= W) { cow -= W; coh++; }
            if (coh >= H) { coh -= H; cn++; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const unsigned char* a = &dyt[wave][u][0] + fr_off;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 256));
            bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            if constexpr (FUSE) {
                const unsigned char* ya = &yt[wave][u][0] + fr_off;
                const s16x4 ylo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ya);
                const s16x4 yhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ya + 256));
                const uint4 dq = __builtin_bit_cast(uint4, af);
                const uint4 yq = __builtin_bit_cast(uint4, __builtin_shufflevector(ylo, yhi, 0, 1, 2, 3, 4, 5, 6, 7));
                const unsigned dd[4] = {dq.x, dq.y, dq.z, dq.w}, yy[4] = {yq.x, yq.y, yq.z, yq.w};
                unsigned oo[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float r2[2];
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const float d = __uint_as_float(hh ? (dd[q] & 0xffff0000u) : (dd[q] << 16));
                        const float a = __uint_as_float(hh ? (yy[q] & 0xffff0000u) : (yy[q] << 16));
                        const float g = d * act_bwd(a * bn_sc + bn_sh, p.act);
                        r2[hh] = bn_sc * g + bn_A * a + bn_B;
                    }
                    oo[q] = pack_bf2(r2[0], r2[1]);
                }
                af = __builtin_bit_cast(bf16x8, make_uint4(oo[0], oo[1], oo[2], oo[3]));
            }
            float w[8];
#pragma unroll
            for (int e = 0; e < 8; e++) w[e] = (okm[u] >> e) & 1u ? v[u][e] : 0.f;   // dead steps: mask 0 -> contribute nothing
            const uint4 b = make_uint4(pack_bf2(w[0], w[1]), pack_bf2(w[2], w[3]), pack_bf2(w[4], w[5]), pack_bf2(w[6], w[7]));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        }
    }
    // acc[co][k]: column = lane & 31 = k, rows = co pattern.  Fold the 4 waves, one slab [32][32] per workgroup.
#pragma unroll
    for (int e = 0; e < 16; e++) redw[wave][(e & 3) + 8 * (e >> 2) + 4 * h][kk] = acc[e];
    __syncthreads();
    for (int i = tid; i < 32 * 32; i += 256) {
        const int co = i >> 5, k = i & 31;
        slabs[(int64_t)blockIdx.x * 1024 + i] = redw[0][co][k] + redw[1][co][k] + redw[2][co][k] + redw[3][co][k];
    }
}

// out[g][1024] = sum of slabs g, g + G, g + 2G, ... (fixed order: deterministic); G = gridDim.x
__global__ __launch_bounds__(1024) void stem_fold_kernel(const float* __restrict__ slabs, int nslab, float* __restrict__ out)
{
    const int i = threadIdx.x;                                     // (co, k) of the 32 x 32 slab
    float s = 0.f;
    for (int z = blockIdx.x; z < nslab; z += gridDim.x) s += slabs[(int64_t)z * 1024 + i];
    out[(int64_t)blockIdx.x * 1024 + i] = s;
}

// ---------------------------------------------------------------------------------------------------------- fused backward
// The layer's whole backward in ONE pass over dz (2.6 GB at 800^2 x 64) + the image (0.5 GB): before, the raw conv output y was
// stored by the forward (2.6 GB), re-read by the BatchNorm pass, by the BatchNorm-backward reduction (with dz) and by the weight
// gradient (with dz again): 8 tensor passes over the largest activation of the network.  Here y is RECOMPUTED (K = 27: two MFMAs
// per 32 pixels) and everything that needs the finished batch statistics is moved behind the pass by linearity:
//     dy = sc*g + A*y + B          (g = dz * act'(sc*y + sh);  A, B from S0 = sum g, S1 = sum g*y)
//     dW = sum dy (x) patch = sc * G + A * Yx + B * X1,   G = sum g (x) patch,  Yx = sum y (x) patch = W . XX,  XX = sum patch (x) patch
// One wave = 32 consecutive pixels of an image row per step.
//   recompute   D[px][c] = patch[px][k] . W[c][k]: the lane = (pixel, k half) gather of the forward kernel as the A operand, the weight
//               fragments as B -> a lane ends with ONE channel (lane & 31) and 16 pixels {4h + 8m + q}: per-channel constants are
//               four registers, the BatchNorm sums two scalar accumulators;
//   dz          LDS-DMA (2 KiB tile) + four transposed reads deliver exactly those (channel, 16 pixels) per lane;
//   G, XX       K = pixels.  The A operand (g) is packed straight from the registers above — the K order of a GEMM is free as long as
//               both operands agree, so the B operand (patch column k = lane & 31) loads the SAME pixel set: two runs of 4
//               consecutive floats per K16 step.  No LDS round trip, no second pass.  Column k = 27 carries ones: XX[27][k] = X1[k].
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void stem3x3_bwd_kernel(const StemBwdParams p, float* __restrict__ slabs)
{
    // wave-private double buffers: image patch (LDS-patch scheme above) + the dz tile [32 px][32 ch] bf16; reused for the final fold
    constexpr int NB = 2, D = NB - 1;                              // one tile ahead (D = 2 measured the same 1.19 ms: instruction-rate bound, see the forward kernel)
    constexpr int WAVE_BYTES = NB * (SP_BYTES + 2048);
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * WAVE_BYTES > 4 * 32 * 33 * 4 ? 4 * WAVE_BYTES : 4 * 32 * 33 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W, HW = H * W;
    const int64_t M = (int64_t)p.NB * HW;
    const int64_t ntiles = M >> 5;                                 // W % 32 == 0: every tile lies in one image row
    unsigned char* const wbase = lds + wave * WAVE_BYTES;
    auto patch_buf = [&](int b) { return wbase + b * (SP_BYTES + 2048); };
    auto dz_buf = [&](int b) { return wbase + b * (SP_BYTES + 2048) + SP_BYTES; };
    StemPatchLane L;
    stem_patch_lane(L, lane, H, W);
    bf16x8 wfrag[2];                                               // B operand of the recompute: column = output channel l31, slot order
    stem_slot_weights(p.wf, l31, h, true, wfrag[0], wfrag[1]);
    const float sc = p.co[2 * 32 + l31], sh = p.co[3 * 32 + l31];
    // patch column of the K = pixels GEMMs: lane = (k = l31, pixel group h) reads patch[krun + 8*rr + e]
    const bool kok = l31 < 27;
    const float kfill = l31 == 27 ? 1.f : 0.f;                     // column 27 carries ones: XX[27][k] = X1[k]
    const int krun = (kok ? sp_kidx(l31) : 0) + 4 * h;
    // transposed read of dz: 16-lane group -> (channel half, pixel group h); quad m adds 8*m pixel rows
    const int s16 = lane & 15, grp = lane >> 4;
    const int tr_off = ((grp >> 1) * 4 + (s16 >> 2)) * 64 + (16 * (grp & 1) + 4 * (s16 & 3)) * 2;
    const int d_row = lane >> 2, d_slot = lane & 3;

    f32x16 accG, accX;
#pragma unroll
    for (int e = 0; e < 16; e++) { accG[e] = 0.f; accX[e] = 0.f; }
    float s0 = 0.f, s1 = 0.f;

    const int64_t nwaves = (int64_t)gridDim.x * 4, per_wave = (ntiles + nwaves - 1) / nwaves;
    const int64_t u0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave)) * per_wave;
    const int64_t u1 = u0 + per_wave < ntiles ? u0 + per_wave : ntiles;
    StemWalk nx;                                                   // the tile whose operands are issued next
    nx.init(u0 < ntiles ? u0 : 0, H, W);
    const bf16_t* dz_lane = p.dz + (int64_t)d_row * p.lddz + d_slot * 8;
    auto issue = [&](int b) {                                       // 4 DMA instructions: the patch (2) + the 2-KiB dz tile (2)
        stem_patch_issue(L, p.img, (nx.n * 3 * H + nx.oh) * W + nx.ow, nx.oh, nx.ow, H, W, patch_buf(b));
        const int64_t t0 = ((int64_t)nx.n * H + nx.oh) * W + nx.ow;
#pragma unroll
        for (int u = 0; u < 2; u++)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(dz_lane + (t0 + 16 * u) * (int64_t)p.lddz), (lds_void_t*)(dz_buf(b) + u * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; d++) {                                   // prologue: tiles u0 .. u0 + D - 1 (past the end: re-reads of the last tile)
        issue(d);
        if (u0 + d + 1 < u1) nx.step(H, W);
    }
    const int offA = (l31 + 120 * h) * 4, offB = (l31 + 40 * h) * 4;
    int buf = 0;
    for (int64_t tt = u0; tt < u1; tt++, buf = buf == D ? 0 : buf + 1) {
        issue(buf == 0 ? D : buf - 1);                             // tile tt + D into the buffer tile tt - 1 just left
        if (tt + D + 1 < u1) nx.step(H, W);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * D) : "memory");   // this tile has landed (wave-private buffers: no barrier)
        const float* patch = reinterpret_cast<const float*>(patch_buf(buf));
        // dz for this lane's (channel, 16 pixels): four transposed reads, through inline asm (the builtin would make hipcc drain the
        // prefetch just issued, conv_internal.h); waited for below, after the recompute
        ry_s16x4 dq[4];
        const unsigned dz_a = lds_addr(dz_buf(buf)) + (unsigned)tr_off;
#pragma unroll
        for (int m = 0; m < 4; m++) dq[m] = lds_tr16(dz_a + (unsigned)(m * 8 * 64));
        // y[px][c] on this lane: channel l31, pixels q + 8m + 4h at acc[4m + q]
        bf16x8 f0, f1;
        const unsigned pbase = lds_addr(patch_buf(buf));
        stem_patch_pixel_frags(pbase + (unsigned)offA, pbase + (unsigned)offB, f0, f1);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, wfrag[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, wfrag[1], acc, 0, 0, 0);
        // patch column runs: pixels {8*rr + 4h + e}
        float xr[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float x = patch[krun + 8 * (i >> 2) + (i & 3)];
            xr[i] = kok ? x : kfill;
        }
        unsigned gp[8];                                              // g packed: pairs (e, e+1)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3]));
        // (r06: the activation derivative of the tile's 16 values in ONE walk of the activation chain, branch-free arms — act_bwd_vec; per
        // element the chain cost three scalar compare + branch pairs in a kernel that is instruction-rate bound.  Same values bit for bit.)
        float yv[16], uv[16], dv[16];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const unsigned yr = pack_bf2(acc[4 * m + 2 * hh], acc[4 * m + 2 * hh + 1]);        // raw output as stored by the two-pass path
                yv[4 * m + 2 * hh] = __uint_as_float(yr << 16);
                yv[4 * m + 2 * hh + 1] = __uint_as_float(yr & 0xffff0000u);
            }
#pragma unroll
        for (int e = 0; e < 16; e++) uv[e] = yv[e] * sc + sh;
        act_bwd_vec<16>(uv, dv, p.act);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const uint2 d2 = __builtin_bit_cast(uint2, dq[m]);
            const unsigned dd[2] = {d2.x, d2.y};
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const float y0 = yv[4 * m + 2 * hh], y1 = yv[4 * m + 2 * hh + 1];
                const float d0 = __uint_as_float(dd[hh] << 16), d1 = __uint_as_float(dd[hh] & 0xffff0000u);
                const float g0 = d0 * dv[4 * m + 2 * hh], g1 = d1 * dv[4 * m + 2 * hh + 1];
                s0 += g0 + g1;
                s1 += g0 * y0 + g1 * y1;
                gp[2 * m + hh] = pack_bf2(g0, g1);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, make_uint4(gp[4 * s2], gp[4 * s2 + 1], gp[4 * s2 + 2], gp[4 * s2 + 3]));
            const bf16x8 bf = __builtin_bit_cast(bf16x8, make_uint4(pack_bf2(xr[8 * s2 + 0], xr[8 * s2 + 1]), pack_bf2(xr[8 * s2 + 2], xr[8 * s2 + 3]),
                                                                   pack_bf2(xr[8 * s2 + 4], xr[8 * s2 + 5]), pack_bf2(xr[8 * s2 + 6], xr[8 * s2 + 7])));
            accG = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, accG, 0, 0, 0);
            accX = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, bf, accX, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- one slab per workgroup: G [32][32], XX [32][32], S0 [32], S1 [32]
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(lds);
    float* slab = slabs + (int64_t)blockIdx.x * STEM_BWD_SLAB;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; e++) red[wave][(e & 3) + 8 * (e >> 2) + 4 * h][l31] = pass ? accX[e] : accG[e];
        __syncthreads();
        for (int i = tid; i < 1024; i += 256) {
            const int r = i >> 5, c = i & 31;
            slab[pass * 1024 + i] = red[0][r][c] + red[1][r][c] + red[2][r][c] + red[3][r][c];
        }
    }
    __syncthreads();
    red[wave][h][l31] = s0;
    red[wave][2 + h][l31] = s1;
    __syncthreads();
    if (tid < 64) {
        const int c = tid & 31, which = tid >> 5;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; w++) t += red[w][2 * which][c] + red[w][2 * which + 1][c];
        slab[2048 + which * 32 + c] = t;
    }
}

// out[g][width] = sum of slabs g, g + G, g + 2G, ... (fixed order: deterministic)
__global__ __launch_bounds__(1024) void stem_fold_wide_kernel(const float* __restrict__ slabs, int nslab, int width, float* __restrict__ out)
{
    for (int i = threadIdx.x; i < width; i += 1024) {
        float s = 0.f;
        for (int z = blockIdx.x; z < nslab; z += gridDim.x) s += slabs[(int64_t)z * width + i];
        out[(int64_t)blockIdx.x * width + i] = s;
    }
}

// folded slab -> BatchNorm coefficients of the backward, dgamma / dbeta, and dW = sc*G + A*(W.XX) + B*X1 into the torch-layout .grad
__global__ __launch_bounds__(1024) void stem_bwd_finalize_kernel(const float* __restrict__ part /*[64][SLAB]*/, const StemBwdParams p, double count)
{
    __shared__ double G[32][33], XX[32][33], S[2][32];
    const int i = threadIdx.x, r = i >> 5, c = i & 31;
    double g = 0.0, x = 0.0;
    for (int z = 0; z < 64; z++) {
        g += (double)part[(int64_t)z * STEM_BWD_SLAB + i];
        x += (double)part[(int64_t)z * STEM_BWD_SLAB + 1024 + i];
    }
    G[r][c] = g;
    XX[r][c] = x;
    if (i < 64) {
        double t = 0.0;
        for (int z = 0; z < 64; z++) t += (double)part[(int64_t)z * STEM_BWD_SLAB + 2048 + i];
        S[i >> 5][i & 31] = t;
    }
    __syncthreads();
    // thread (r = output channel, c = k)
    const double mu = p.co[r], is = p.co[32 + r], scd = p.co[64 + r];
    const double S0 = S[0][r], S1 = S[1][r];
    const double gx = is * (S1 - mu * S0);                          // sum g * xhat
    const double mg = p.frozen ? 0.0 : S0 / count, mx = p.frozen ? 0.0 : gx / count;
    if (c == 0) {
        if (p.dgamma) p.dgamma[r] += (float)gx;
        if (p.dbeta) p.dbeta[r] += (float)S0;
    }
    if (c < 27) {
        double yx = 0.0;
        for (int k = 0; k < 27; k++) yx += (double)bf2f(p.wf[r * 32 + k]) * XX[k][c];
        const double A = -scd * is * mx, B = scd * (is * mx * mu - mg);
        const double dw = scd * G[r][c] + A * yx + B * XX[27][c];
        const int tap = c / 3, cin = c - tap * 3;
        p.dW[(r * 3 + cin) * 9 + tap] += (float)dw;
    }
}

// ---------------------------------------------------------------------------------------------------------- C ABI
static int stem_blocks(int64_t M) { const int64_t t = ry_cdiv(M, 32 * 4 * 8); return (int)(t > 2048 ? 2048 : (t < 1 ? 1 : t)); }
// LDS-patch kernels: the grid is what is resident at once (256 CUs x up to 4 workgroups), every wave walks its tiles with a prefetch
static int stem_lds_blocks(int64_t M, int per_cu = 4) { const int64_t t = ry_cdiv(M, 32 * 4 * 2), cap = 256 * per_cu; return (int)(t > cap ? cap : (t < 1 ? 1 : t)); }
static bool stem_no_lds() { static const bool v = getenv("RYOLO_STEM_NO_LDS") != nullptr; return v; }   // A/B: the register-gather forward

static bool stem_ok(int NB, int H, int W, int Cout)
{
    return NB > 0 && H > 0 && W > 0 && Cout > 0 && Cout <= 32 && Cout % 8 == 0 && (int64_t)NB * 3 * H * W < (1ll << 31);
}

extern "C" int ryolo_stem3x3_plan(int NB, int H, int W, int Cout, int* stats_rows, size_t* wgrad_workspace_bytes)
{
    if (!stem_ok(NB, H, W, Cout)) return RY_ERR_UNSUPPORTED;
    const int nb = stem_blocks((int64_t)NB * H * W);
    if (stats_rows) *stats_rows = (W % 32 == 0 && !stem_no_lds()) ? stem_lds_blocks((int64_t)NB * H * W) : nb;
    if (wgrad_workspace_bytes) *wgrad_workspace_bytes = (size_t)(nb + 64) * 1024 * sizeof(float);
    return RY_OK;
}

extern "C" int ryolo_stem3x3_fwd(const StemParams* pp, hipStream_t stream)
{
    if (!pp || !pp->img || !pp->wf) return RY_ERR_ARG;
    const StemParams& p = *pp;
    if (!p.out && p.epi != EPI_STATS) return RY_ERR_ARG;          // statistics-only pass: nothing stored
    if (!stem_ok(p.NB, p.H, p.W, p.Cout) || (p.out && (p.ldC % 8 || (reinterpret_cast<uintptr_t>(p.out) & 15)))) return RY_ERR_UNSUPPORTED;
    if (p.epi != EPI_RAW && p.epi != EPI_STATS && p.epi != EPI_AFFINE_ACT && p.epi != EPI_AFFINE_ACT_R) return RY_ERR_ARG;
    if ((p.epi == EPI_STATS && !p.stats) || ((p.epi == EPI_AFFINE_ACT || p.epi == EPI_AFFINE_ACT_R) && (!p.scale || !p.shift))) return RY_ERR_ARG;
    if (p.W % 32 == 0 && !stem_no_lds()) {
        const dim3 g((unsigned)stem_lds_blocks((int64_t)p.NB * p.H * p.W)), b(256);
        if (p.epi == EPI_STATS) hipLaunchKernelGGL(stem3x3_fwd_lds_kernel<EPI_STATS>, g, b, 0, stream, p);
        else if (p.epi == EPI_AFFINE_ACT) hipLaunchKernelGGL(stem3x3_fwd_lds_kernel<EPI_AFFINE_ACT>, g, b, 0, stream, p);
        else if (p.epi == EPI_AFFINE_ACT_R) hipLaunchKernelGGL(stem3x3_fwd_lds_kernel<EPI_AFFINE_ACT_R>, g, b, 0, stream, p);
        else hipLaunchKernelGGL(stem3x3_fwd_lds_kernel<EPI_RAW>, g, b, 0, stream, p);
    } else
        hipLaunchKernelGGL(stem3x3_fwd_kernel, dim3((unsigned)stem_blocks((int64_t)p.NB * p.H * p.W)), dim3(256), 0, stream, p);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_stem3x3_wgrad(const StemWgradParams* pp, hipStream_t stream)
{
    if (!pp || !pp->img || !pp->dY || !pp->scratch || !pp->workspace) return RY_ERR_ARG;
    const StemWgradParams& p = *pp;
    if (!stem_ok(p.NB, p.H, p.W, p.Cout) || p.Cout != 32 || p.W % 16 || p.ldY % 8) return RY_ERR_UNSUPPORTED;
    const int nb = stem_blocks((int64_t)p.NB * p.H * p.W);
    if (p.y) {
        if (!p.co || !p.bco || p.ldy % 8) return RY_ERR_ARG;
        hipLaunchKernelGGL(stem3x3_wgrad_kernel<true>, dim3((unsigned)nb), dim3(256), 0, stream, p, p.workspace);
    } else {
        hipLaunchKernelGGL(stem3x3_wgrad_kernel<false>, dim3((unsigned)nb), dim3(256), 0, stream, p, p.workspace);
    }
    float* part = p.workspace + (size_t)nb * 1024;                 // two-level fold: 64 groups, then one
    hipLaunchKernelGGL(stem_fold_kernel, dim3(64), dim3(1024), 0, stream, p.workspace, nb, part);
    hipLaunchKernelGGL(stem_fold_kernel, dim3(1), dim3(1024), 0, stream, part, 64, p.scratch);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_stem3x3_bwd_plan(int NB, int H, int W, int Cout, size_t* workspace_bytes)
{
    if (!stem_ok(NB, H, W, Cout) || Cout != 32 || W % 32) return RY_ERR_UNSUPPORTED;
    const int nb = stem_lds_blocks((int64_t)NB * H * W, 3);       // 158 VGPRs: three workgroups per CU
    if (workspace_bytes) *workspace_bytes = (size_t)(nb + 64) * STEM_BWD_SLAB * sizeof(float);
    return RY_OK;
}

extern "C" int ryolo_stem3x3_bwd(const StemBwdParams* pp, hipStream_t stream)
{
    if (!pp || !pp->img || !pp->dz || !pp->wf || !pp->co || !pp->workspace || !pp->dW) return RY_ERR_ARG;
    const StemBwdParams& p = *pp;
    if (!stem_ok(p.NB, p.H, p.W, 32) || p.W % 32 || p.lddz % 8 || (reinterpret_cast<uintptr_t>(p.dz) & 15)) return RY_ERR_UNSUPPORTED;
    const int nb = stem_lds_blocks((int64_t)p.NB * p.H * p.W, 3);
    hipLaunchKernelGGL(stem3x3_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, p, p.workspace);
    float* part = p.workspace + (size_t)nb * STEM_BWD_SLAB;
    hipLaunchKernelGGL(stem_fold_wide_kernel, dim3(64), dim3(1024), 0, stream, p.workspace, nb, STEM_BWD_SLAB, part);
    hipLaunchKernelGGL(stem_bwd_finalize_kernel, dim3(1), dim3(1024), 0, stream, part, p, (double)p.NB * p.H * p.W);
    RY_CHECK_LAUNCH();
    return RY_OK;
}
