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
#include "common.h"
#include "params.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define BK 32



enum { EPI_RAW = 0, EPI_STATS = 1, EPI_AFFINE_ACT = 2, EPI_F32_BIAS = 3, EPI_ACCUM = 4 };
enum { ACT_LINEAR = 0, ACT_MISH = 1, ACT_LEAKY = 2, ACT_SILU = 3 };

__device__ __forceinline__ float act_fwd(float u, int act)
{
    if (act == ACT_SILU) return u / (1.f + __expf(-u));
    if (act == ACT_LEAKY) return u > 0.f ? u : 0.1f * u;
    if (act == ACT_MISH) {
        const float sp = u > 20.f ? u : log1pf(__expf(u));
        return u * tanhf(sp);
    }
    return u;
}

// bijective XCD remap (cdna guide T1): workgroup b runs on XCD b%8; give each XCD a contiguous tile range
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmParams p)
{
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int PA = (BM * 4 + 255) / 256, PB = (BN * 4 + 255) / 256;       // 16-byte pieces per thread
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && BM % 64 == 0, "tile config");
    constexpr int WTM = BM / WM, WTN = BN / WN;               // rows x cols of one wave's output block
    constexpr int EP_LD = WTN + 8;                            // staging row stride (bf16), 16-byte aligned, breaks bank aliasing
    constexpr int MAINLOOP_ELEMS = 2 * (BM + BN) * BK, EPI_ELEMS = 4 * WTM * EP_LD;
    __shared__ __attribute__((aligned(16))) bf16_t smem[MAINLOOP_ELEMS > EPI_ELEMS ? MAINLOOP_ELEMS : EPI_ELEMS];
#define sA_(b) (smem + (b) * (BM + BN) * BK)
#define sB_(b) (smem + (b) * (BM + BN) * BK + BM * BK)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const TapClass& tc = p.cls[blockIdx.z];
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    const int gridN = (p.Nout + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int mb = tile / gridN, nb = tile - mb * gridN;
    const int64_t m0 = (int64_t)mb * BM;
    const int n0 = nb * BN;

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
    const int cchunks = p.Cin / BK;
    const int nk = tc.ntaps * cchunks;

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

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (k + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------------------
    // C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const bool identity = (p.oh_mul == 1 && p.ow_mul == 1 && p.OHf == p.OH && p.OWf == p.OW && tc.oh_add == 0 && tc.ow_add == 0);
    float csum[TN], csq[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) { csum[j] = 0.f; csq[j] = 0.f; }

    if (p.epi == EPI_F32_BIAS || p.epi == EPI_AFFINE_ACT) {
        // small / rare outputs (detection heads, fused eval epilogue): direct per-element stores
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int64_t m = m0 + row;
                if (m >= M) continue;
                int64_t pix = m;
                if (!identity) {
                    const int img = (int)(m / ((int64_t)p.OH * p.OW));
                    const int rem = (int)(m - (int64_t)img * p.OH * p.OW);
                    const int oh = rem / p.OW, ow = rem - oh * p.OW;
                    pix = ((int64_t)img * p.OHf + (oh * p.oh_mul + tc.oh_add)) * p.OWf + (ow * p.ow_mul + tc.ow_add);
                }
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const int n = n0 + wn * WTN + j * 32 + (lane & 31);
                    if (n >= p.Nout) continue;
                    float v = acc[i][j][e];
                    if (p.epi == EPI_F32_BIAS) {
                        if (p.bias) v += p.bias[n];
                        reinterpret_cast<float*>(p.out)[pix * p.ldC + n] = v;
                    } else {
                        v = act_fwd(v * p.scale[n] + p.shift[n], p.act);
                        reinterpret_cast<bf16_t*>(p.out)[pix * p.ldC + n] = f2bf(v);
                    }
                }
            }
        }
    } else {
        // bf16 outputs: stage the wave's WTM x WTN block in LDS, then write whole 16-byte row segments (8 channels per lane)
        __syncthreads();                                       // every wave is done reading the operand tiles
        bf16_t* stage = smem + wave * WTM * EP_LD;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int r = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    const bf16_t b = f2bf(acc[i][j][e]);
                    stage[r * EP_LD + j * 32 + (lane & 31)] = b;
                    if (p.epi == EPI_STATS) {
                        const bool live = (m0 + wm * WTM + r) < M;          // rows past M are zero anyway (zero-filled A rows)
                        const float rv = live ? bf2f(b) : 0.f;              // statistics of the values actually stored
                        csum[j] += rv;
                        csq[j] += rv * rv;
                    }
                }
        // (same-wave LDS hand-off: no workgroup barrier needed, only the wave's own ds_write -> ds_read ordering)
        constexpr int CH = WTN / 8;                            // 16-byte chunks per row
        constexpr int RPI = 64 / CH;                           // rows per iteration
        const int ch = lane % CH, r0 = lane / CH;
        const int n = n0 + wn * WTN + ch * 8;
#pragma unroll
        for (int it = 0; it < WTM / RPI; it++) {
            const int r = it * RPI + r0;
            const int64_t m = m0 + wm * WTM + r;
            if (m >= M || n >= p.Nout) continue;
            int64_t pix = m;
            if (!identity) {
                const int img = (int)(m / ((int64_t)p.OH * p.OW));
                const int rem = (int)(m - (int64_t)img * p.OH * p.OW);
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                pix = ((int64_t)img * p.OHf + (oh * p.oh_mul + tc.oh_add)) * p.OWf + (ow * p.ow_mul + tc.ow_add);
            }
            uint4 v = *reinterpret_cast<const uint4*>(stage + r * EP_LD + ch * 8);
            bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + pix * p.ldC + n;
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
    if (p.epi == EPI_STATS) {
        // column sums: combine the two half-waves, then the WM waves that share a column block (through LDS)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [WM][2][BN]
#pragma unroll
        for (int j = 0; j < TN; j++) {
            float s = csum[j] + __shfl_xor(csum[j], 32, 64);
            float q = csq[j] + __shfl_xor(csq[j], 32, 64);
            if (lane < 32) {
                const int col = wn * (BN / WN) + j * 32 + lane;
                red[(wm * 2 + 0) * BN + col] = s;
                red[(wm * 2 + 1) * BN + col] = q;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.Nout) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WM; w++) { s += red[(w * 2 + 0) * BN + tid]; q += red[(w * 2 + 1) * BN + tid]; }
            float* st = p.stats + (int64_t)mb * 2 * p.Nout;
            st[n0 + tid] = s;
            st[p.Nout + n0 + tid] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient

// BM = 128 output channels x BN = 128 (tap,cin) columns; K = pixels.  LDS tiles are pixel-major [BK][128+PAD].
// pixel-major LDS rows of 128 channels + 32 pad: row stride 320 B = 64 B (mod 256 B), so the 4 rows x 32 B a 16-lane group of
// ds_read_b64_tr_b16 touches (and the neighbouring group's +32 B) fall on 8 disjoint bank ranges -> conflict free
#define WG_LD 160
template <int BM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p)
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
    const int i0 = blockIdx.x * BM;                    // output-channel block
    const int q0 = blockIdx.y * 4;                     // first 32-wide column chunk (tap-major, then cin)
    const int cchunks = p.Cin / BK;
    const int nchunks = p.ntaps * cchunks;
    const int64_t kbeg = (int64_t)blockIdx.z * p.kchunk;
    const int64_t kend = min(M, kbeg + p.kchunk);
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg + BK - 1) / BK);

    // A pieces: BK pixels x BM channels -> BM/8 16B pieces per pixel
    constexpr int APP = BM / 8;                        // pieces per pixel row
    constexpr int PA = BK * APP / 256;                 // 2 (BM=128) or 1 (BM=64)
    // B pieces: 4 chunks x BK pixels x 4 slots = 512 -> 2 per thread
    uint4 ra[PA], rb[2];
    int b_tap[2], b_c0[2];
    bool b_chunk_ok[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int id = tid + 256 * u;
        const int q = q0 + (id >> 7);
        b_chunk_ok[u] = q < nchunks;
        const int qq = b_chunk_ok[u] ? q : 0;
        b_tap[u] = qq / cchunks;
        b_c0[u] = (qq - b_tap[u] * cchunks) * BK + (id & 3) * 8;
    }
    auto gload = [&](int step) {
        const int64_t mk = kbeg + (int64_t)step * BK;
#pragma unroll
        for (int u = 0; u < PA; u++) {
            const int id = tid + 256 * u;
            const int px = id / APP, pc = id % APP;
            const int64_t m = mk + px;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < kend && i0 + pc * 8 < p.CoutPad) v = *reinterpret_cast<const uint4*>(p.dY + m * p.ldY + i0 + pc * 8);
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int id = tid + 256 * u;
            const int px = (id & 127) >> 2;
            const int64_t m = mk + px;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < kend && b_chunk_ok[u]) {
                const int img = (int)(m / ((int64_t)p.OH * p.OW));
                const int rem = (int)(m - (int64_t)img * p.OH * p.OW);
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                const int ih = oh * p.sh + p.dh[b_tap[u]], iw = ow * p.sw + p.dw[b_tap[u]];
                if ((unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW)
                    v = *reinterpret_cast<const uint4*>(p.X + (((int64_t)img * p.IH + ih) * p.IW + iw) * p.ldX + b_c0[u]);
            }
            rb[u] = v;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PA; u++) {
            const int id = tid + 256 * u;
            const int px = id / APP, pc = id % APP;
            *reinterpret_cast<uint4*>(wA_(buf) + px * WG_LD + pc * 8) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int id = tid + 256 * u;
            const int ch = id >> 7, px = (id & 127) >> 2, sl = id & 3;
            *reinterpret_cast<uint4*>(wB_(buf) + px * WG_LD + ch * 32 + sl * 8) = rb[u];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    gload(0);
    sstore(0);
    __syncthreads();
    for (int k = 0; k < nk; k++) {
        const int buf = k & 1;
        if (k + 1 < nk) gload(k + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            // Transposed fragment reads (gfx950 ds_read_b64_tr_b16): the tiles are pixel-major [pixel][channel] as they come
            // from HBM, the MFMA wants 8 consecutive pixels (K) per lane for ONE channel.  Within each 16-lane group the
            // instruction transposes a 4(pixel) x 16(channel) block: lane s supplies the address of pixel (s>>2), channels
            // 4*(s&3)..+3 and receives channel s for the 4 pixels.  Two reads (pixels +0..3, +4..7) build one operand.
            const int s16 = lane & 15, grp = lane >> 4;
            const int prow = ks * 16 + (grp >> 1) * 8 + (s16 >> 2);
            const int pcol = 16 * (grp & 1) + 4 * (s16 & 3);
            bf16x8 af[TM], bfr[TN];
            unsigned long long lo[TM + TN], hi[TM + TN];
            unsigned addr[TM + TN];
#pragma unroll
            for (int i = 0; i < TM; i++)
                addr[i] = (unsigned)(size_t)(wA_(buf) + prow * WG_LD + wm * (BM / WM) + i * 32 + pcol);
#pragma unroll
            for (int j = 0; j < TN; j++)
                addr[TM + j] = (unsigned)(size_t)(wB_(buf) + prow * WG_LD + wn * (BN / WN) + j * 32 + pcol);
            // all reads and their wait in ONE asm statement (hipcc does not count asm LDS reads): early-clobber outputs
            if constexpr (TM + TN == 4) {
                asm volatile(
                    "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:%12\n\t"
                    "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:%12\n\t"
                    "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:%12\n\t"
                    "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:%12\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2]), "=&v"(lo[3]), "=&v"(hi[3])
                    : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "i"(4 * WG_LD * 2)
                    : "memory");
            } else {
                static_assert(TM + TN == 3, "fragment count");
                asm volatile(
                    "ds_read_b64_tr_b16 %0, %6\n\tds_read_b64_tr_b16 %1, %6 offset:%9\n\t"
                    "ds_read_b64_tr_b16 %2, %7\n\tds_read_b64_tr_b16 %3, %7 offset:%9\n\t"
                    "ds_read_b64_tr_b16 %4, %8\n\tds_read_b64_tr_b16 %5, %8 offset:%9\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2])
                    : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "i"(4 * WG_LD * 2)
                    : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;
#pragma unroll
            for (int i = 0; i < TM; i++) { u64x2 t = {lo[i], hi[i]}; af[i] = __builtin_bit_cast(bf16x8, t); }
#pragma unroll
            for (int j = 0; j < TN; j++) { u64x2 t = {lo[TM + j], hi[TM + j]}; bfr[j] = __builtin_bit_cast(bf16x8, t); }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (k + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    // split-K partial tile -> workspace [z][Cout][ntaps*Cin] (GEMM layout; 128-byte row segments per store instruction).
    // No float atomics: the reduction over z is a separate deterministic pass.
    const int NK = p.ntaps * p.Cin;
    float* part = p.partial + (int64_t)blockIdx.z * p.Cout * NK;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int col = wn * (BN / WN) + j * 32 + (lane & 31);
        const int q = q0 + (col >> 5);
        if (q >= nchunks) continue;
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

// dW[co][cin][tap] += sum_z partial[z][co][tap*Cin + cin]   (torch weight layout, one thread per element, fixed order)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int splitk, int Cout, int Cin, int ntaps,
                                                           float* __restrict__ dW)
{
    const int NK = ntaps * Cin;
    const int64_t total = (int64_t)Cout * NK;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < splitk; z++) s += partial[(int64_t)z * total + i];
        const int co = (int)(i / NK), kc = (int)(i - (int64_t)co * NK);
        const int tap = kc / Cin, cin = kc - tap * Cin;
        dW[((int64_t)co * Cin + cin) * ntaps + tap] += s;
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
template <int BM, int BN, int WM, int WN>
static int launch_gemm(const ConvGemmParams& p, hipStream_t stream)
{
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    const int64_t gm = ry_cdiv(M, BM), gn = ry_cdiv(p.Nout, BN);
    if (gm * gn > 0x7fffffff) return RY_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN>), dim3((unsigned)(gm * gn), 1, p.nclasses), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
}

extern "C" int ryolo_conv_gemm_stats_rows(int64_t M, int Nout, int* rows)
{
    // number of [2][Nout] partial-statistics rows the EPI_STATS epilogue writes (== gridM of the chosen tile)
    if (!rows) return RY_ERR_ARG;
    *rows = (int)ry_cdiv(M, Nout <= 32 ? 256 : 128);
    return RY_OK;
}

extern "C" int ryolo_conv_gemm(const ConvGemmParams* pp, hipStream_t stream)
{
    if (!pp) return RY_ERR_ARG;
    const ConvGemmParams& p = *pp;
    if (!p.A || !p.W || !p.out || p.Cin <= 0 || p.Cin % BK || p.ldA % 8 || p.Nout <= 0 || p.nclasses < 1 || p.nclasses > 4)
        return RY_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W)) & 15) return RY_ERR_ARG;
    for (int c = 0; c < p.nclasses; c++)
        if (p.cls[c].ntaps < 1 || p.cls[c].ntaps > RY_MAX_TAPS) return RY_ERR_ARG;
    if (p.epi == EPI_STATS && (!p.stats || p.nclasses != 1)) return RY_ERR_ARG;
    if ((int64_t)p.NB * p.OH * p.OW <= 0) return RY_OK;
    if (p.Nout <= 32) return launch_gemm<256, 32, 4, 1>(p, stream);
    if (p.Nout <= 64) return launch_gemm<128, 64, 2, 2>(p, stream);
    return launch_gemm<128, 128, 2, 2>(p, stream);
}

static int wgrad_geometry(WgradParams& p, int& bm, int& gx, int& gy)
{
    if (p.Cin <= 0 || p.Cin % BK || p.ldX % 8 || p.ldY % 8 || p.Cout <= 0 || p.CoutPad % 8 || p.CoutPad < p.Cout || p.CoutPad > p.ldY ||
        p.ntaps < 1 || p.ntaps > RY_MAX_TAPS)
        return RY_ERR_ARG;
    const int64_t M = (int64_t)p.NB * p.OH * p.OW;
    bm = p.Cout <= 64 ? 64 : 128;
    gx = (int)ry_cdiv(p.Cout, bm);
    gy = (int)ry_cdiv((int64_t)p.ntaps * (p.Cin / BK), 4);
    int64_t want = ry_cdiv(2048, (int64_t)gx * gy);                  // ~8 workgroups per CU
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
    *splitk = p.splitk;
    *workspace_bytes = (size_t)p.splitk * p.Cout * p.ntaps * p.Cin * sizeof(float);
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
    if (bm == 64)
        hipLaunchKernelGGL((conv_wgrad_kernel<64>), dim3(gx, gy, p.splitk), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<128>), dim3(gx, gy, p.splitk), dim3(256), 0, stream, p);
    const int64_t total = (int64_t)p.Cout * p.ntaps * p.Cin;
    int64_t g = ry_cdiv(total, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, p.partial, p.splitk, p.Cout, p.Cin, p.ntaps, p.dW);
    RY_CHECK_LAUNCH();
    return RY_OK;
}
