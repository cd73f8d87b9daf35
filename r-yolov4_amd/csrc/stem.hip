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
            if (p.epi == EPI_AFFINE_ACT && c0 < p.Cout) {
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = act_fwd(v[q] * p.scale[c0 + q] + p.shift[c0 + q], p.act);
            }
            const uint2 w = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            *reinterpret_cast<uint2*>(&otile[wave][px_l][c0]) = w;             // staged: the lane's 8-byte piece of its pixel row
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
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int pr = half * 16 + (lane >> 2), sl = lane & 3;
            const int64_t opix = tt * 32 + pr;
            const uint4 o = *reinterpret_cast<const uint4*>(&otile[wave][pr][sl * 8]);
            if (opix < M && sl * 8 < p.Cout) *reinterpret_cast<uint4*>(p.out + opix * p.ldC + sl * 8) = o;
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

// ---------------------------------------------------------------------------------------------------------- C ABI
static int stem_blocks(int64_t M) { const int64_t t = ry_cdiv(M, 32 * 4 * 8); return (int)(t > 2048 ? 2048 : (t < 1 ? 1 : t)); }

static bool stem_ok(int NB, int H, int W, int Cout)
{
    return NB > 0 && H > 0 && W > 0 && Cout > 0 && Cout <= 32 && Cout % 8 == 0 && (int64_t)NB * 3 * H * W < (1ll << 31);
}

extern "C" int ryolo_stem3x3_plan(int NB, int H, int W, int Cout, int* stats_rows, size_t* wgrad_workspace_bytes)
{
    if (!stem_ok(NB, H, W, Cout)) return RY_ERR_UNSUPPORTED;
    const int nb = stem_blocks((int64_t)NB * H * W);
    if (stats_rows) *stats_rows = nb;
    if (wgrad_workspace_bytes) *wgrad_workspace_bytes = (size_t)(nb + 64) * 1024 * sizeof(float);
    return RY_OK;
}

extern "C" int ryolo_stem3x3_fwd(const StemParams* pp, hipStream_t stream)
{
    if (!pp || !pp->img || !pp->wf || !pp->out) return RY_ERR_ARG;
    const StemParams& p = *pp;
    if (!stem_ok(p.NB, p.H, p.W, p.Cout) || p.ldC % 8 || (reinterpret_cast<uintptr_t>(p.out) & 15)) return RY_ERR_UNSUPPORTED;
    if (p.epi != EPI_RAW && p.epi != EPI_STATS && p.epi != EPI_AFFINE_ACT) return RY_ERR_ARG;
    if ((p.epi == EPI_STATS && !p.stats) || (p.epi == EPI_AFFINE_ACT && (!p.scale || !p.shift))) return RY_ERR_ARG;
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
