// Memory-bound layers of the conv stack for gfx950: training/eval BatchNorm + activation
// (+ RepConv's second branch, + Bottleneck's residual add), their backward, MaxPool (k2 s2; k5/9/13 s1), nearest 2x
// upsample, the small-Cin im2col front end, head finish (ImplicitM + [B,na,gs,gs,attrs] layout), ImplicitA, the bf16
// weight repack and the fused Nesterov SGD step.
// Reference: model/utils.py:6-32 (Conv = conv -> BatchNorm2d(eps 1e-5, momentum .1) -> Mish | LeakyReLU(.1) | SiLU),
// :35-46 (Bottleneck residual), :146-160 (MaxConv), :163-186 (ImplicitA/M), :189-215 (RepConv), :218-282 (SPP*),
// model/neck.py (nn.Upsample(scale_factor=2)), train.py:153-158 (SGD momentum .937 nesterov).
//
// All activations are NHWC bf16 with a channel stride (`ld`) so outputs land directly in their torch.cat slice.
// Every kernel moves 16 bytes (8 bf16) per lane per access; per-channel reductions are two-level and deterministic
// (per-workgroup partial rows, then a finalize kernel accumulating in double) — no float atomics.
#include "common.h"
#include "params.h"

enum { ACT_LINEAR = 0, ACT_MISH = 1, ACT_LEAKY = 2, ACT_SILU = 3 };

// Activations and their derivatives with ONE v_exp_f32 and one/two v_rcp_f32 per element (no IEEE division, no libm):
// these kernels move 4-6 bytes per element, so a libm tanhf/log1pf or a correctly-rounded divide per element makes them
// VALU-bound instead of HBM-bound.  Mish uses tanh(softplus(u)) = (n^2 + 2n) / (n^2 + 2n + 2) with n = e^u.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float act_f(float u, int act)
{
    if (act == ACT_SILU) return u * fast_rcp(1.f + __expf(-u));
    if (act == ACT_LEAKY) return u > 0.f ? u : 0.1f * u;
    if (act == ACT_MISH) {
        if (u > 20.f) return u;
        const float n = __expf(u), w = n * (n + 2.f);
        return u * w * fast_rcp(w + 2.f);
    }
    return u;
}
__device__ __forceinline__ float act_d(float u, int act)
{
    if (act == ACT_SILU) { const float s = fast_rcp(1.f + __expf(-u)); return s * (1.f + u * (1.f - s)); }
    if (act == ACT_LEAKY) return u > 0.f ? 1.f : 0.1f;
    if (act == ACT_MISH) {
        if (u > 20.f) return 1.f;
        const float n = __expf(u), w = n * (n + 2.f);
        const float t = w * fast_rcp(w + 2.f);                 // tanh(softplus(u))
        const float sg = n * fast_rcp(1.f + n);                // sigmoid(u)
        return t + u * (1.f - t * t) * sg;
    }
    return 1.f;
}

struct V8 { float v[8]; };
// Streaming access policy (same-box A/B on the whole step, yolov7 800^2 batch 64): these kernels touch every activation once or twice
// and the tensors are far larger than L2 + Infinity Cache, so loads and stores are NONTEMPORAL (-1.45 ms/step: they stop evicting
// what the neighbouring GEMMs re-read) — except the loads of the BN-backward REDUCE pass, whose two operands are read again by the
// apply pass right after it (plain loads there: another -0.4 ms).
typedef unsigned ew_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ V8 unpack8(const uint4 r)
{
    V8 o;
    o.v[0] = __uint_as_float(r.x << 16); o.v[1] = __uint_as_float(r.x & 0xffff0000u);
    o.v[2] = __uint_as_float(r.y << 16); o.v[3] = __uint_as_float(r.y & 0xffff0000u);
    o.v[4] = __uint_as_float(r.z << 16); o.v[5] = __uint_as_float(r.z & 0xffff0000u);
    o.v[6] = __uint_as_float(r.w << 16); o.v[7] = __uint_as_float(r.w & 0xffff0000u);
    return o;
}
__device__ __forceinline__ V8 ld8(const bf16_t* p)
{
    const ew_u32x4 rv = __builtin_nontemporal_load(reinterpret_cast<const ew_u32x4*>(p));
    return unpack8(make_uint4(rv.x, rv.y, rv.z, rv.w));
}
// asm loads with hand-counted waits (bn_act_bwd_reduce_kernel's read-ahead): the outputs are "defined" for the compiler at the request and
// must pass through ew_vmwait before their first use
__device__ __forceinline__ void ew_gload(ew_u32x4& r, const bf16_t* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory"); }
__device__ __forceinline__ void ew_gload_nt(ew_u32x4& r, const bf16_t* p) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r) : "v"(p) : "memory"); }
template <int N> __device__ __forceinline__ void ew_vmwait(ew_u32x4& a, ew_u32x4& b, ew_u32x4& c)
{
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
__device__ __forceinline__ uint4 ld8_raw(const bf16_t* p)
{
    const ew_u32x4 rv = __builtin_nontemporal_load(reinterpret_cast<const ew_u32x4*>(p));
    return make_uint4(rv.x, rv.y, rv.z, rv.w);
}
__device__ __forceinline__ V8 ld8_keep(const bf16_t* p) { return unpack8(*reinterpret_cast<const uint4*>(p)); }    // will be read again soon
__device__ __forceinline__ void st8(bf16_t* p, const V8& a)
{
    const ew_u32x4 rv = {pack_bf2(a.v[0], a.v[1]), pack_bf2(a.v[2], a.v[3]), pack_bf2(a.v[4], a.v[5]), pack_bf2(a.v[6], a.v[7])};
    __builtin_nontemporal_store(rv, reinterpret_cast<ew_u32x4*>(p));
}

// ------------------------------------------------------------------------------------------------ partial-row folding
// in [rows][K*C] float -> out [S][K*C] float: slice s sums rows s, s+S, s+2S, ... in double.  Keeps the finalize kernels
// (one workgroup per 32 channels) short when a big layer produced tens of thousands of per-tile partial rows.
__global__ __launch_bounds__(1024) void fold_rows_kernel(const float* __restrict__ in, int rows, int KC, int S, float* __restrict__ out)
{
    const int cl = threadIdx.x & 63;
    const int col = blockIdx.x * 64 + cl;
    const int s = blockIdx.y, rl = threadIdx.x >> 6;             // 16 row lanes per slice (4 lanes walked the 160 k tile rows of the first layers in 135 us)
    __shared__ double red[16][64];
    double acc = 0.0;
    if (col < KC) {
        int r = s + rl * S;
        for (; r + 48 * S < rows; r += 64 * S) {                  // 4 independent loads in flight
            const float v0 = in[(int64_t)r * KC + col], v1 = in[(int64_t)(r + 16 * S) * KC + col];
            const float v2 = in[(int64_t)(r + 32 * S) * KC + col], v3 = in[(int64_t)(r + 48 * S) * KC + col];
            acc += (double)v0; acc += (double)v1; acc += (double)v2; acc += (double)v3;
        }
        for (; r < rows; r += 16 * S) acc += (double)in[(int64_t)r * KC + col];
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && col < KC) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++) t += red[k][cl];
        out[(int64_t)s * KC + col] = (float)t;
    }
}
#define FOLD_S 64
#define FOLD_DIRECT 4096
// RYOLO_BN_FOLD_DIRECT (default FOLD_DIRECT = 4096; 256 = the round-3 behaviour): up to how many partial rows the finalize kernels sum without a
// fold_rows pass in front (A/B knob, read once)
static int bn_fold_direct()
{
    static const int v = [] { const char* e = getenv("RYOLO_BN_FOLD_DIRECT"); int x = e ? atoi(e) : FOLD_DIRECT; return x < 4 * FOLD_S ? 4 * FOLD_S : x; }();
    return v;
}
static inline const float* fold_rows(const float* partial, int& rows, int KC, float* scratch, hipStream_t stream)
{
    if (rows <= 4 * FOLD_S || !scratch) return partial;
    hipLaunchKernelGGL(fold_rows_kernel, dim3((unsigned)ry_cdiv(KC, 64), FOLD_S), dim3(1024), 0, stream, partial, rows, KC, FOLD_S, scratch);
    rows = FOLD_S;
    return scratch;
}

// ------------------------------------------------------------------------------------------------ BN finalize
// partial [rows][2][C] (sum, sumsq of the stored bf16 conv output) -> mean/invstd/scale/shift; running stats update
// (momentum 0.1, unbiased variance — nn.BatchNorm2d defaults used by model/utils.py:17)
// ld / c0: the partial rows are [2][ld] wide and this BatchNorm owns channels [c0, c0 + C) of them (sibling convolutions that share
// one GEMM launch share one statistics buffer; ld == C, c0 == 0 for a stand-alone layer)
// CH = channels per workgroup: 32 (32 row lanes: short row lists) or 8 (128 row lanes: up to FOLD_DIRECT partial rows are summed HERE, without the
// fold_rows launch in front — at 8 images per GPU nearly every BatchNorm had 300 ... 2500 partial rows and paid a 5 us launch + a dependent
// boundary for that pass, forward and backward: ~140 launches per step).  Row lanes sum in double with 4 loads in flight; the lanes of a channel
// are combined in two fixed-order LDS stages (deterministic).
template <int CH>
__device__ __forceinline__ void fold_lanes(double (*red)[1024], double* s, int nq, int cl, int rl)
{
    constexpr int NRL = 1024 / CH, G = NRL >= 8 ? 8 : NRL;       // stage 1: G lanes per channel each add NRL / G partials, stage 2: lane 0 adds G
    for (int q = 0; q < nq; q++) red[q][threadIdx.x] = s[q];
    __syncthreads();
    if (rl < G) {
        for (int q = 0; q < nq; q++) {
            double t = 0.0;
#pragma unroll 4
            for (int k = rl; k < NRL; k += G) t += red[q][k * CH + cl];
            s[q] = t;
        }
    }
    __syncthreads();
    if (rl < G)
        for (int q = 0; q < nq; q++) red[q][rl * CH + cl] = s[q];
    __syncthreads();
    if (rl == 0)
        for (int q = 0; q < nq; q++) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < G; k++) t += red[q][k * CH + cl];
            s[q] = t;
        }
}

template <int CH>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int rows, int ld, int c0, int C, double count, float eps,
                                                           float momentum, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ out /*[4][C]*/)
{
    __shared__ double red[2][1024];
    constexpr int NRL = 1024 / CH;
    const int cl = threadIdx.x % CH, rl = threadIdx.x / CH;
    const int c = blockIdx.x * CH + cl;
    double sq[2] = {0.0, 0.0};
    if (c < C) {
        const float* base = partial + c0 + c;
        int r = rl;
        for (; r + 3 * NRL < rows; r += 4 * NRL) {                 // 8 independent loads in flight
            float a[4], b[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { a[k] = base[((int64_t)(r + k * NRL) * 2 + 0) * ld]; b[k] = base[((int64_t)(r + k * NRL) * 2 + 1) * ld]; }
#pragma unroll
            for (int k = 0; k < 4; k++) { sq[0] += (double)a[k]; sq[1] += (double)b[k]; }
        }
        for (; r < rows; r += NRL) {
            sq[0] += (double)base[((int64_t)r * 2 + 0) * ld];
            sq[1] += (double)base[((int64_t)r * 2 + 1) * ld];
        }
    }
    fold_lanes<CH>(red, sq, 2, cl, rl);
    if (rl == 0 && c < C) {
        const double s = sq[0], q = sq[1];
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        out[0 * C + c] = (float)mean;
        out[1 * C + c] = invstd;
        out[2 * C + c] = sc;
        out[3 * C + c] = beta[c] - (float)mean * sc;
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

// eval mode: scale/shift from running statistics
__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                                      float* out /*[4][ld], channels [c0, c0 + C)*/, int ld, int c0)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * invstd;
    float* o = out + c0 + c;
    o[0] = rm[c]; o[ld] = invstd; o[2 * ld] = sc; o[3 * ld] = beta[c] - rm[c] * sc;
}

// ------------------------------------------------------------------------------------------------ BN + act forward

// Thread mapping shared by the three BN+act kernels: a thread owns ONE group of 8 channels and walks rows; 256 threads =
// cols column-groups x (256/cols) rows per iteration.  The kernels are templates over (activation, second branch, residual)
// so the per-element code is branch free and small enough for 5-8 waves per SIMD (the first, runtime-switched version
// needed ~200 VGPRs -> 2 waves per SIMD and ran at 2.4 TB/s).  Algebra is arranged for few per-channel registers:
//   forward          u = sc*y + sh                                   (2 coefficients / channel)
//   backward reduce  S0 = sum g, S1 = sum g*y  (g = dz*act'(u));  sum g*xhat = is*(S1 - mu*S0) is formed at finalize
//   backward apply   dy = sc*(g - mg - xhat*mx) = sc*g + A*y + Bc,  A = -sc*is*mx,  Bc = sc*(is*mx*mu - mg)
template <int ACT, bool Y2, bool RES>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const BnActParams p)
{
    const int c8 = p.C >> 3;
    const int cols = c8 < 256 ? c8 : 256;
    const int rpi = 256 / cols;
    const int rl = threadIdx.x / cols, cl = threadIdx.x - rl * cols;
    if (rl >= rpi) return;
    for (int cb = 0; cb < c8; cb += cols) {
        const int cc = cb + cl;
        if (cc >= c8) continue;
        const int c = cc << 3;
        float sc1[8], sh1[8], sc2[8], sh2[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            sc1[k] = p.co1[2 * p.C + c + k]; sh1[k] = p.co1[3 * p.C + c + k];
            if (Y2) { sc2[k] = p.co2[2 * p.C + c + k]; sh2[k] = p.co2[3 * p.C + c + k]; }
        }
#pragma unroll 2
        for (int64_t m = (int64_t)blockIdx.x * rpi + rl; m < p.M; m += (int64_t)gridDim.x * rpi) {
            const V8 a = ld8(p.y1 + m * p.ld1 + c);
            V8 b, r, o;
            if (Y2) b = ld8(p.y2 + m * p.ld2 + c);
            if (RES) r = ld8(p.res + m * p.ldr + c);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float u = a.v[k] * sc1[k] + sh1[k];
                if (Y2) u += b.v[k] * sc2[k] + sh2[k];
                float zv = act_f(u, ACT);
                if (RES) zv += r.v[k];
                o.v[k] = zv;
            }
            st8(p.z + m * p.ldz + c, o);
        }
    }
}

// backward pass 1: per-channel S0 = sum g, S1 = sum g*y1 (, S2 = sum g*y2); partial [nblk][K][C]
template <int ACT, bool Y2>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const BnActParams p)
{
    constexpr int K = Y2 ? 3 : 2;
    __shared__ float red[K][256][8 + 1];                          // 18 KiB without the second branch: 8 workgroups per CU
    const int c8 = p.C >> 3;
    const int cols = c8 < 256 ? c8 : 256;
    const int rl = threadIdx.x / cols, nrl = 256 / cols;
    const int cl = threadIdx.x - rl * cols;
    const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
    const int64_t r1 = min(p.M, r0 + p.rows_per_block);
    for (int cb = 0; cb < c8; cb += cols) {
        const int cc = cb + cl;
        float s0[8], s1[8], s2[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { s0[k] = 0.f; s1[k] = 0.f; s2[k] = 0.f; }
        if (rl < nrl && cc < c8) {
            const int c = cc << 3;
            float sc1[8], sh1[8], sc2[8], sh2[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                sc1[k] = p.co1[2 * p.C + c + k]; sh1[k] = p.co1[3 * p.C + c + k];
                if (Y2) { sc2[k] = p.co2[2 * p.C + c + k]; sh2[k] = p.co2[3 * p.C + c + k]; }
            }
            // One row ahead in flight: the raw 16-byte loads of row i + 1 are issued before the arithmetic of row i (the last round re-reads
            // its own row: branch-free).  With the fold below kept out of the register budget this is 5 waves per SIMD x 2 rows in flight;
            // the first version (no read-ahead, 168 VGPRs = 3 waves) ran at 4.5 TB/s beside the 5.9 of its siblings — it waited for
            // latency, ~500 VALU cycles per row and wave against a ~3000-cycle round trip.  (Compiler unrolling by 2 / 4 was measured
            // the same / 1.4x slower in r02: it doubled the registers instead of overlapping the loads.)
            // back-to-front sweep: the tail of dz (just written front-to-back by the dgrad kernels) is still in the 256 MiB
            // Infinity Cache, and the forward-sweeping apply pass then starts on what this pass read last (measured: -0.3 ms/step;
            // reversing the apply or the forward pass instead: -0.15 / -0.05 ms)
            const int64_t mf = r0 + rl;
            if (mf < r1) {
                const int n = (int)((r1 - 1 - mf) / nrl) + 1;          // rows of this lane
                const bf16_t* pd = p.dz + (p.M - 1 - mf) * p.lddz + c;
                const bf16_t* pa = p.y1 + (p.M - 1 - mf) * p.ld1 + c;
                const bf16_t* pb = Y2 ? p.y2 + (p.M - 1 - mf) * p.ld2 + c : nullptr;
                const int64_t sd = -(int64_t)nrl * p.lddz, sa = -(int64_t)nrl * p.ld1, sb = Y2 ? -(int64_t)nrl * p.ld2 : 0;
                auto row = [&](const ew_u32x4 dr, const ew_u32x4 ar, const ew_u32x4 br) {
                    const V8 d = unpack8(make_uint4(dr.x, dr.y, dr.z, dr.w)), a = unpack8(make_uint4(ar.x, ar.y, ar.z, ar.w));
                    V8 b;
                    if (Y2) b = unpack8(make_uint4(br.x, br.y, br.z, br.w));
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        float u = a.v[k] * sc1[k] + sh1[k];
                        if (Y2) u += b.v[k] * sc2[k] + sh2[k];
                        const float g = d.v[k] * act_d(u, ACT);
                        s0[k] += g;
                        s1[k] += g * a.v[k];
                        if (Y2) s2[k] += g * b.v[k];
                    }
                };
                // Two register sets, rows alternate between them (no copies); the loads and their waits are asm: hipcc's own vmcnt bookkeeping
                // put a vmcnt(0) at the loop header (the set requested in the previous half-round had to land before the next request went
                // out), which is the serialization this loop exists to remove.  NL loads per set, in-order return: vmcnt(NL) = "the older set
                // is here".  No other vector-memory instruction may sit inside the loop.
                constexpr int NL = Y2 ? 3 : 2;
                ew_u32x4 d0, a0, b0 = {0, 0, 0, 0}, d1, a1, b1 = {0, 0, 0, 0};
                __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0) hipcc can see: the coefficient loads end HERE, not at their first use inside the loop
                ew_gload(d0, pd); ew_gload(a0, pa);
                if (Y2) ew_gload_nt(b0, pb);
#pragma unroll 1
                for (int i = 0;; i += 2) {
                    const bool m1 = i + 1 < n;
                    if (m1) { pd += sd; pa += sa; if (Y2) pb += sb; }
                    ew_gload(d1, pd); ew_gload(a1, pa);
                    if (Y2) ew_gload_nt(b1, pb);
                    ew_vmwait<NL>(d0, a0, b0);
                    row(d0, a0, b0);
                    if (!m1) break;
                    const bool m2 = i + 2 < n;
                    if (m2) { pd += sd; pa += sa; if (Y2) pb += sb; }
                    ew_gload(d0, pd); ew_gload(a0, pa);
                    if (Y2) ew_gload_nt(b0, pb);
                    ew_vmwait<NL>(d1, a1, b1);
                    row(d1, a1, b1);
                    if (!m2) break;
                }
                ew_vmwait<0>(d0, a0, b0);                             // the read-ahead of the last round (a row already summed) is still in flight
                ew_vmwait<0>(d1, a1, b1);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) { red[0][threadIdx.x][k] = s0[k]; red[1][threadIdx.x][k] = s1[k]; if (Y2) red[2][threadIdx.x][k] = s2[k]; }
        __syncthreads();
        // fold of the row lanes: one (statistic, channel) per thread and round, row lanes added in ascending order (the order of the first
        // version, whose fully unrolled form held the kernel at 168 VGPRs = 3 waves per SIMD: the streaming loop above wants the occupancy)
        const int ncol = min(cols, c8 - cb) * 8;                   // channels of this column block
#pragma unroll 1
        for (int idx = threadIdx.x; idx < K * ncol; idx += 256) {
            const int q = idx / ncol, ch = idx - q * ncol;
            const float* r = &red[q][ch >> 3][ch & 7];
            float s = 0.f;
#pragma unroll 1
            for (int j = 0; j < nrl; j++) s += r[j * cols * 9];
            p.partial[((int64_t)blockIdx.x * K + q) * p.C + (cb << 3) + ch] = s;
        }
    }
}

// backward finalize: sums over blocks (double) -> coefficients + dgamma/dbeta accumulation.
// sum g*xhat = invstd * (S1 - mean*S0) per branch (formed in double).
template <int K, int CH>
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count, int frozen,
                                                               const float* __restrict__ co1, const float* __restrict__ co2,
                                                               float* __restrict__ bco /*[K][C]*/, float* __restrict__ dgamma1,
                                                               float* __restrict__ dbeta1, float* __restrict__ dgamma2,
                                                               float* __restrict__ dbeta2)
{
    // K is a template parameter so that both loops unroll: with a runtime K the 32-step LDS fold ran as 96 dependent
    // ds_read + wait iterations (18 us per launch, 89 launches on the critical path of a step; 4.8 us for the forward twin).
    // CH: see bn_finalize_kernel.
    __shared__ double red[K][1024];
    constexpr int NRL = 1024 / CH;
    const int cl = threadIdx.x % CH, rl = threadIdx.x / CH;
    const int c = blockIdx.x * CH + cl;
    double s[K];
#pragma unroll
    for (int q = 0; q < K; q++) s[q] = 0.0;
    if (c < C) {
        int r = rl;
        for (; r + 3 * NRL < nblk; r += 4 * NRL) {
            float v[4][K];
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int q = 0; q < K; q++) v[k][q] = partial[((int64_t)(r + k * NRL) * K + q) * C + c];
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int q = 0; q < K; q++) s[q] += (double)v[k][q];
        }
        for (; r < nblk; r += NRL)
#pragma unroll
            for (int q = 0; q < K; q++) s[q] += (double)partial[((int64_t)r * K + q) * C + c];
    }
    fold_lanes<CH>(red, s, K, cl, rl);
    if (rl == 0 && c < C) {
        const double gx1 = (double)co1[C + c] * (s[1] - (double)co1[c] * s[0]);
        double gx2 = 0.0;
        if constexpr (K == 3) gx2 = (double)co2[C + c] * (s[2] - (double)co2[c] * s[0]);
        // frozen statistics (eval-mode BatchNorm used as a fixed affine map): no coupling through the batch mean/variance
        const double rc = 1.0 / count;
        bco[0 * C + c] = frozen ? 0.f : (float)(s[0] * rc);
        bco[1 * C + c] = frozen ? 0.f : (float)(gx1 * rc);
        if constexpr (K == 3) bco[2 * C + c] = frozen ? 0.f : (float)(gx2 * rc);
        if (dgamma1) { dgamma1[c] += (float)gx1; dbeta1[c] += (float)s[0]; }
        if constexpr (K == 3)
            if (dgamma2) { dgamma2[c] += (float)gx2; dbeta2[c] += (float)s[0]; }
    }
}

// backward pass 2: dy = sc*g + A*y + Bc (see header); residual gradient = dz
template <int ACT, bool Y2, bool DRES>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const BnActParams p)
{
    const int c8 = p.C >> 3;
    const int cols = c8 < 256 ? c8 : 256;
    const int rpi = 256 / cols;
    const int rl = threadIdx.x / cols, cl = threadIdx.x - rl * cols;
    if (rl >= rpi) return;
    for (int cb = 0; cb < c8; cb += cols) {
        const int cc = cb + cl;
        if (cc >= c8) continue;
        const int c = cc << 3;
        float sc1[8], sh1[8], A1[8], B1[8], sc2[8], sh2[8], A2[8], B2[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float mg = p.bco[c + k];
            {
                const float mu = p.co1[c + k], is = p.co1[p.C + c + k], sc = p.co1[2 * p.C + c + k], mx = p.bco[p.C + c + k];
                sc1[k] = sc; sh1[k] = p.co1[3 * p.C + c + k]; A1[k] = -sc * is * mx; B1[k] = sc * (is * mx * mu - mg);
            }
            if (Y2) {
                const float mu = p.co2[c + k], is = p.co2[p.C + c + k], sc = p.co2[2 * p.C + c + k], mx = p.bco[2 * p.C + c + k];
                sc2[k] = sc; sh2[k] = p.co2[3 * p.C + c + k]; A2[k] = -sc * is * mx; B2[k] = sc * (is * mx * mu - mg);
            }
        }
        // one row = 16 bytes of dz and of y per lane; TWO rows are requested before the first is worked on (5 waves per SIMD x one row in
        // flight left the pass latency-bound at 5.2-5.8 TB/s)
        auto row = [&](const int64_t m, const uint4 dr, const uint4 ar, const uint4 br) {
            const V8 d = unpack8(dr), a = unpack8(ar);
            V8 b, o1, o2;
            if (Y2) b = unpack8(br);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float u = a.v[k] * sc1[k] + sh1[k];
                if (Y2) u += b.v[k] * sc2[k] + sh2[k];
                const float g = d.v[k] * act_d(u, ACT);
                o1.v[k] = sc1[k] * g + A1[k] * a.v[k] + B1[k];
                if (Y2) o2.v[k] = sc2[k] * g + A2[k] * b.v[k] + B2[k];
            }
            st8(p.dy1 + m * p.lddy1 + c, o1);
            if (Y2) st8(p.dy2 + m * p.lddy2 + c, o2);
            if (DRES) {
                V8 r = d;
                if (p.dres_accum) {
                    const V8 e = ld8(p.dres + m * p.lddres + c);
#pragma unroll
                    for (int k = 0; k < 8; k++) r.v[k] += e.v[k];
                }
                st8(p.dres + m * p.lddres + c, r);
            }
        };
        const int64_t mstep = (int64_t)gridDim.x * rpi;
        int64_t m = (int64_t)blockIdx.x * rpi + rl;
        const uint4 z4 = make_uint4(0, 0, 0, 0);
#pragma unroll 1
        for (; m + mstep < p.M; m += 2 * mstep) {
            const int64_t m1 = m + mstep;
            const uint4 d0 = ld8_raw(p.dz + m * p.lddz + c), a0 = ld8_raw(p.y1 + m * p.ld1 + c), b0 = Y2 ? ld8_raw(p.y2 + m * p.ld2 + c) : z4;
            const uint4 d1 = ld8_raw(p.dz + m1 * p.lddz + c), a1 = ld8_raw(p.y1 + m1 * p.ld1 + c), b1 = Y2 ? ld8_raw(p.y2 + m1 * p.ld2 + c) : z4;
            row(m, d0, a0, b0);
            row(m1, d1, a1, b1);
        }
        if (m < p.M) row(m, ld8_raw(p.dz + m * p.lddz + c), ld8_raw(p.y1 + m * p.ld1 + c), Y2 ? ld8_raw(p.y2 + m * p.ld2 + c) : z4);
    }
}

// eval-mode / frozen-statistics backward is not needed: the reference only back-propagates in train() mode.

// ------------------------------------------------------------------------------------------------ pooling / upsample

// Index plan shared by the pooling kernels: a thread keeps ONE 8-channel chunk (cc) and walks pixels; the pixel index is split into
// (n, h, w) with exact float-reciprocal divisions (pixels < 2^24; larger tensors take the integer path).  The first version did
// three 64-bit divisions per 16 bytes moved and ran at ~1.7 TB/s (VALU bound).
struct PixIter {
    int c8w, ppb, pl, cc;
    bool active;
    __device__ PixIter(int c8)
    {
        c8w = c8 < 256 ? c8 : 256;
        ppb = 256 / c8w;
        pl = (int)threadIdx.x / c8w;
        cc = (int)threadIdx.x - pl * c8w;
        active = pl < ppb;
    }
};
__device__ __forceinline__ void split_pixel(int64_t pix, int H, int W, float rHW, float rW, int& n, int& h, int& w)
{
    if (pix < (1 << 24)) {
        const int q = (int)pix, HW = H * W;
        int nn = (int)((float)q * rHW);
        if (nn * HW > q) nn--;
        if ((nn + 1) * HW <= q) nn++;
        const int rem = q - nn * HW;
        int hh = (int)((float)rem * rW);
        if (hh * W > rem) hh--;
        if ((hh + 1) * W <= rem) hh++;
        n = nn; h = hh; w = rem - hh * W;
    } else {
        n = (int)(pix / ((int64_t)H * W));
        const int rem = (int)(pix - (int64_t)n * H * W);
        h = rem / W;
        w = rem - h * W;
    }
}
__device__ __forceinline__ int div_stride(int x, int stride) { return stride == 1 ? x : (stride == 2 && x >= 0 ? x >> 1 : x / stride); }

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const PoolParams p)
{
    const int c8 = p.C >> 3;
    const PixIter it(c8);
    if (!it.active) return;
    const int64_t npix = (int64_t)p.NB * p.OH * p.OW;
    const float rHW = 1.0f / (float)(p.OH * p.OW), rW = 1.0f / (float)p.OW;
    for (int64_t pix = (int64_t)blockIdx.x * it.ppb + it.pl; pix < npix; pix += (int64_t)gridDim.x * it.ppb) {
        int n, oh, ow;
        split_pixel(pix, p.OH, p.OW, rHW, rW, n, oh, ow);
        for (int cb = it.cc; cb < c8; cb += it.c8w) {
            const int c = cb << 3;
            float best[8];
            int bi[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { best[k] = -INFINITY; bi[k] = 0; }
            for (int dy = 0; dy < p.k; dy++) {
                const int ih = oh * p.stride - p.pad + dy;
                if ((unsigned)ih >= (unsigned)p.H) continue;
                for (int dx = 0; dx < p.k; dx++) {
                    const int iw = ow * p.stride - p.pad + dx;
                    if ((unsigned)iw >= (unsigned)p.W) continue;
                    const V8 v = ld8(p.x + (((int64_t)n * p.H + ih) * p.W + iw) * p.ldx + c);
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        if (v.v[k] > best[k]) { best[k] = v.v[k]; bi[k] = dy * p.k + dx; }      // strict >: first max wins
                }
            }
            V8 o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = best[k];
            st8(p.z + pix * p.ldz + c, o);
            if (p.idx) {
                unsigned long long packed = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) packed |= (unsigned long long)(bi[k] & 0xff) << (8 * k);
                *reinterpret_cast<unsigned long long*>(p.idx + pix * p.C + c) = packed;
            }
        }
    }
}

// gather-form backward (deterministic, no atomics): every input pixel scans the windows that contain it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const PoolParams p)
{
    const int c8 = p.C >> 3;
    const PixIter it(c8);
    if (!it.active) return;
    const int64_t npix = (int64_t)p.NB * p.H * p.W;
    const float rHW = 1.0f / (float)(p.H * p.W), rW = 1.0f / (float)p.W;
    for (int64_t pix = (int64_t)blockIdx.x * it.ppb + it.pl; pix < npix; pix += (int64_t)gridDim.x * it.ppb) {
        int n, h, w;
        split_pixel(pix, p.H, p.W, rHW, rW, n, h, w);
        // windows (oh, ow) with oh*stride - pad <= h < oh*stride - pad + k
        const int oh_lo = max(0, div_stride(h + p.pad - p.k + p.stride, p.stride)), oh_hi = min(p.OH - 1, div_stride(h + p.pad, p.stride));
        const int ow_lo = max(0, div_stride(w + p.pad - p.k + p.stride, p.stride)), ow_hi = min(p.OW - 1, div_stride(w + p.pad, p.stride));
        for (int cb = it.cc; cb < c8; cb += it.c8w) {
            const int c = cb << 3;
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] = 0.f;
            for (int oh = oh_lo; oh <= oh_hi; oh++)
                for (int ow = ow_lo; ow <= ow_hi; ow++) {
                    const int want = (h - (oh * p.stride - p.pad)) * p.k + (w - (ow * p.stride - p.pad));
                    const int64_t op = ((int64_t)n * p.OH + oh) * p.OW + ow;
                    const unsigned long long packed = *reinterpret_cast<const unsigned long long*>(p.idx + op * p.C + c);
                    const V8 g = ld8(p.dz + op * p.lddz + c);
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        if ((int)((packed >> (8 * k)) & 0xff) == want) acc[k] += g.v[k];
                }
            V8 o;
            if (p.accum) {
                const V8 e = ld8(p.dx + pix * p.lddx + c);
#pragma unroll
                for (int k = 0; k < 8; k++) o.v[k] = e.v[k] + acc[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) o.v[k] = acc[k];
            }
            st8(p.dx + pix * p.lddx + c, o);
        }
    }
}


__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const UpParams p)      // z[2H,2W] <- x[H,W]
{
    const int c8 = p.C >> 3;
    const int64_t total = (int64_t)p.NB * (2 * p.H) * (2 * p.W) * c8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i / c8;
        const int c = (int)(i - pix * c8) << 3;
        const int ow = (int)(pix % (2 * p.W));
        const int oh = (int)((pix / (2 * p.W)) % (2 * p.H));
        const int n = (int)(pix / ((int64_t)4 * p.W * p.H));
        const uint4 v = *reinterpret_cast<const uint4*>(p.x + (((int64_t)n * p.H + (oh >> 1)) * p.W + (ow >> 1)) * p.ldx + c);
        *reinterpret_cast<uint4*>(p.z + pix * p.ldz + c) = v;
    }
}

__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const UpParams p)      // x-grad[H,W] (=|+=) sum of 4 z-grads; here x=dz(2H,2W) z=dx(H,W)
{
    const int c8 = p.C >> 3;
    const int64_t total = (int64_t)p.NB * p.H * p.W * c8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i / c8;
        const int c = (int)(i - pix * c8) << 3;
        const int w = (int)(pix % p.W);
        const int h = (int)((pix / p.W) % p.H);
        const int n = (int)(pix / ((int64_t)p.W * p.H));
        V8 o;
#pragma unroll
        for (int k = 0; k < 8; k++) o.v[k] = 0.f;
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                const V8 g = ld8(p.x + (((int64_t)n * 2 * p.H + 2 * h + a) * (2 * p.W) + 2 * w + b) * p.ldx + c);
#pragma unroll
                for (int k = 0; k < 8; k++) o.v[k] += g.v[k];
            }
        if (p.accum) {
            const V8 e = ld8(p.z + pix * p.ldz + c);
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] += e.v[k];
        }
        st8(p.z + pix * p.ldz + c, o);
    }
}

// ---- stride-1 pools with large windows (SPP k = 5 / 9 / 13): row pass + column pass -------------------------------------------------
// The direct form reads k*k inputs per output (169 at k = 13) and its gather-form backward scans k*k windows per input.  A window
// maximum is separable; so is its FIRST-maximum argmax in (dy, dx) scanning order: the winner is the smallest dy whose row maximum
// equals the window maximum, and within that row the smallest dx — exactly (column pass first-max over dy) of (row pass first-max over
// dx).  2k reads per output in both directions, same indices and gradients as the direct kernels.
__global__ __launch_bounds__(256) void pool_rows_fwd_kernel(const PoolParams p)
{
    const int c8 = p.C >> 3;
    const PixIter it(c8);
    if (!it.active) return;
    const int64_t npix = (int64_t)p.NB * p.H * p.W;
    const float rHW = 1.0f / (float)(p.H * p.W), rW = 1.0f / (float)p.W;
    for (int64_t pix = (int64_t)blockIdx.x * it.ppb + it.pl; pix < npix; pix += (int64_t)gridDim.x * it.ppb) {
        int n, h, ow;
        split_pixel(pix, p.H, p.W, rHW, rW, n, h, ow);
        for (int cb = it.cc; cb < c8; cb += it.c8w) {
            const int c = cb << 3;
            float best[8];
            int bi[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { best[k] = -INFINITY; bi[k] = 0; }
            for (int dx = 0; dx < p.k; dx++) {
                const int iw = ow - p.pad + dx;
                if ((unsigned)iw >= (unsigned)p.W) continue;
                const V8 v = ld8(p.x + (pix - ow + iw) * p.ldx + c);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (v.v[k] > best[k]) { best[k] = v.v[k]; bi[k] = dx; }
            }
            V8 o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = best[k];
            st8(p.rowmax + pix * p.C + c, o);
            if (p.rowidx) {
                unsigned long long packed = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) packed |= (unsigned long long)(bi[k] & 0xff) << (8 * k);
                *reinterpret_cast<unsigned long long*>(p.rowidx + pix * p.C + c) = packed;
            }
        }
    }
}

__global__ __launch_bounds__(256) void pool_cols_fwd_kernel(const PoolParams p)
{
    const int c8 = p.C >> 3;
    const PixIter it(c8);
    if (!it.active) return;
    const int64_t npix = (int64_t)p.NB * p.H * p.W;                // stride 1, pad k/2: the output grid equals the input grid
    const float rHW = 1.0f / (float)(p.H * p.W), rW = 1.0f / (float)p.W;
    for (int64_t pix = (int64_t)blockIdx.x * it.ppb + it.pl; pix < npix; pix += (int64_t)gridDim.x * it.ppb) {
        int n, oh, ow;
        split_pixel(pix, p.H, p.W, rHW, rW, n, oh, ow);
        for (int cb = it.cc; cb < c8; cb += it.c8w) {
            const int c = cb << 3;
            float best[8];
            int bi[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { best[k] = -INFINITY; bi[k] = 0; }
            for (int dy = 0; dy < p.k; dy++) {
                const int ih = oh - p.pad + dy;
                if ((unsigned)ih >= (unsigned)p.H) continue;
                const int64_t rp = pix + (int64_t)(ih - oh) * p.W;
                const V8 v = ld8(p.rowmax + rp * p.C + c);
                unsigned long long rix = 0;
                if (p.idx) rix = *reinterpret_cast<const unsigned long long*>(p.rowidx + rp * p.C + c);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (v.v[k] > best[k]) { best[k] = v.v[k]; bi[k] = dy * p.k + (int)((rix >> (8 * k)) & 0xff); }
            }
            V8 o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = best[k];
            st8(p.z + pix * p.ldz + c, o);
            if (p.idx) {
                unsigned long long packed = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) packed |= (unsigned long long)(bi[k] & 0xff) << (8 * k);
                *reinterpret_cast<unsigned long long*>(p.idx + pix * p.C + c) = packed;
            }
        }
    }
}

// backward, column pass: g_row[n, ih, ow] = sum over the outputs (oh, ow) whose winning row is ih of dz[n, oh, ow]
__global__ __launch_bounds__(256) void pool_cols_bwd_kernel(const PoolParams p)
{
    const int c8 = p.C >> 3;
    const PixIter it(c8);
    if (!it.active) return;
    const int64_t npix = (int64_t)p.NB * p.H * p.W;
    const float rHW = 1.0f / (float)(p.H * p.W), rW = 1.0f / (float)p.W, rk = 1.0f / (float)p.k;
    for (int64_t pix = (int64_t)blockIdx.x * it.ppb + it.pl; pix < npix; pix += (int64_t)gridDim.x * it.ppb) {
        int n, ih, ow;
        split_pixel(pix, p.H, p.W, rHW, rW, n, ih, ow);
        for (int cb = it.cc; cb < c8; cb += it.c8w) {
            const int c = cb << 3;
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] = 0.f;
            for (int dy = 0; dy < p.k; dy++) {
                const int oh = ih + p.pad - dy;                    // output row whose window row dy is ih
                if ((unsigned)oh >= (unsigned)p.OH) continue;
                const int64_t op = pix + (int64_t)(oh - ih) * p.W;
                const unsigned long long packed = *reinterpret_cast<const unsigned long long*>(p.idx + op * p.C + c);
                const V8 g = ld8(p.dz + op * p.lddz + c);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int id = (int)((packed >> (8 * k)) & 0xff);
                    const int wdy = (int)(((float)id + 0.5f) * rk);                      // id / k, exact for id < 256
                    if (wdy == dy) acc[k] += g.v[k];
                }
            }
            float4* o = reinterpret_cast<float4*>(p.growws + pix * p.C + c);
            o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
}

// backward, row pass: dx[n, ih, iw] (+)= sum over the row-window positions (ih, ow) whose winning column is iw of g_row[n, ih, ow]
__global__ __launch_bounds__(256) void pool_rows_bwd_kernel(const PoolParams p)
{
    const int c8 = p.C >> 3;
    const PixIter it(c8);
    if (!it.active) return;
    const int64_t npix = (int64_t)p.NB * p.H * p.W;
    const float rHW = 1.0f / (float)(p.H * p.W), rW = 1.0f / (float)p.W;
    for (int64_t pix = (int64_t)blockIdx.x * it.ppb + it.pl; pix < npix; pix += (int64_t)gridDim.x * it.ppb) {
        int n, ih, iw;
        split_pixel(pix, p.H, p.W, rHW, rW, n, ih, iw);
        for (int cb = it.cc; cb < c8; cb += it.c8w) {
            const int c = cb << 3;
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] = 0.f;
            for (int dx = 0; dx < p.k; dx++) {
                const int ow = iw + p.pad - dx;                    // row-window position whose column dx is iw
                if ((unsigned)ow >= (unsigned)p.W) continue;
                const int64_t rp = pix - iw + ow;
                const unsigned long long packed = *reinterpret_cast<const unsigned long long*>(p.rowidx + rp * p.C + c);
                const float4* g = reinterpret_cast<const float4*>(p.growws + rp * p.C + c);
                const float4 g0 = g[0], g1 = g[1];
                const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if ((int)((packed >> (8 * k)) & 0xff) == dx) acc[k] += gv[k];
            }
            V8 o;
            if (p.accum) {
                const V8 e = ld8(p.dx + pix * p.lddx + c);
#pragma unroll
                for (int k = 0; k < 8; k++) o.v[k] = e.v[k] + acc[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) o.v[k] = acc[k];
            }
            st8(p.dx + pix * p.lddx + c, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------ small-Cin im2col
// img fp32 NCHW [NB,3,H,W] (what train.py:186 hands over) -> col [NB*OH*OW][Kpad] bf16, k = (r*kw + s)*Cin + c
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, int NB, int Cin, int H, int W, int kh, int kw,
                                                     int stride, int pad, int OH, int OW, int Kpad, bf16_t* __restrict__ col)
{
    const int64_t total = (int64_t)NB * OH * OW * (Kpad >> 3);
    const int k8 = Kpad >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i / k8;
        const int kk = (int)(i - pix * k8) << 3;
        const int ow = (int)(pix % OW);
        const int oh = (int)((pix / OW) % OH);
        const int n = (int)(pix / ((int64_t)OW * OH));
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int k = kk + e;
            float v = 0.f;
            if (k < kh * kw * Cin) {
                const int tap = k / Cin, c = k - tap * Cin;
                const int r = tap / kw, s = tap - r * kw;
                const int ih = oh * stride - pad + r, iw = ow * stride - pad + s;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = img[(((int64_t)n * Cin + c) * H + ih) * W + iw];
            }
            o.v[e] = v;
        }
        st8(col + pix * Kpad + kk, o);
    }
}

// ------------------------------------------------------------------------------------------------ head finish / implicit
// pre [M][ldp] fp32 (head conv + bias) -> out [B, na, gs, gs, attrs] fp32, times ImplicitM (mul may be null)
// One workgroup = one image x 128 cells x all anchors, a thread owns channels ch = tid, tid + 256, ...: pre rows are read whole and
// contiguous, the outputs of one anchor are contiguous runs of cells x attrs floats; no index division per element.
// Both head kernels move fp32 tensors between two layouts — GEMM rows [cell][na*attrs] and the reference's [na][cell][attrs] — whose
// contiguous pieces are 88 bytes (attrs = 22) on one side: the direct version did 4-byte accesses on that side and ran at
// 2.6-3.0 TB/s.  Here a tile of TC cells x all channels is staged in LDS, so BOTH sides use 16-byte accesses: whole rows of the
// GEMM layout, and the (TC x attrs)-float run of each anchor, which is contiguous in the reference layout.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte access at 4-byte alignment (runs start at odd offsets)
static int head_tile_cells(int C) { int tc = 32; while (tc > 1 && ((size_t)tc * (C + 1) + C) * 4 > 60u * 1024u) tc >>= 1; return tc; }

template <bool OBJ>
__global__ __launch_bounds__(256) void head_finish_fwd_kernel(const float* __restrict__ pre, int ldp, const float* __restrict__ mul, int B, int gs,
                                                              int na, int attrs, int TC, float* __restrict__ out, float* __restrict__ preobj, int och,
                                                              float* __restrict__ xobj)
{
    extern __shared__ float hl[];                              // t[TC][C + 1], mulv[C]
    const int cells = gs * gs, C = na * attrs, LD = C + 1;
    const int ncb = (cells + TC - 1) / TC;
    const int b = blockIdx.x / ncb, cb = blockIdx.x - b * ncb;
    const int c0 = cb * TC;
    const int ncell = min(TC, cells - c0);
    const float* src = pre + ((int64_t)b * cells + c0) * ldp;
    // OBJ (r05, ryolo_head_finish_fwd_obj): the tile is staged UNSCALED and ImplicitM is applied on the way out, so that the objectness column of
    // every (anchor, cell) BEFORE ImplicitM can be copied out of the tile as a compact [B, na, cells] array — what the sparse head backward
    // (head_bwd_sparse_kernel) multiplies the objectness gradients with for the ImplicitM gradient, instead of re-reading all of `pre`.
    float* const mulv = hl + TC * LD;
    const bool late = OBJ && mul;
    if (late) for (int ch = threadIdx.x; ch < C; ch += 256) mulv[ch] = mul[ch];
    if (((C | ldp) & 3) == 0 && (reinterpret_cast<uintptr_t>(pre) & 15) == 0) {
        const int c4n = C >> 2;
        for (int i = threadIdx.x; i < ncell * c4n; i += 256) {
            const int cell = i / c4n, c4 = i - cell * c4n;
            float4 v = *reinterpret_cast<const float4*>(src + (int64_t)cell * ldp + c4 * 4);
            if (mul && !late) { const float4 m = *reinterpret_cast<const float4*>(mul + c4 * 4); v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w; }
            float* t = hl + cell * LD + c4 * 4;
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        }
    } else {
        for (int i = threadIdx.x; i < ncell * C; i += 256) {
            const int cell = i / C, ch = i - cell * C;
            hl[cell * LD + ch] = src[(int64_t)cell * ldp + ch] * ((mul && !late) ? mul[ch] : 1.f);
        }
    }
    __syncthreads();
    if (OBJ)
        for (int i = threadIdx.x; i < na * ncell; i += 256) {
            const int a = i / ncell, cell = i - a * ncell;
            const float v = hl[cell * LD + a * attrs + och];
            preobj[((int64_t)b * na + a) * cells + c0 + cell] = v;
            if (xobj) xobj[((int64_t)b * na + a) * cells + c0 + cell] = late ? v * mulv[a * attrs + och] : v;     // = out's element, same product
        }
    const int run = ncell * attrs, nq = (run + 3) >> 2;
    const float rattrs = 1.0f / (float)attrs;
    for (int i = threadIdx.x; i < na * nq; i += 256) {
        const int a = i / nq, q = i - a * nq;
        const int e0 = q * 4;
        float* dst = out + (((int64_t)b * na + a) * cells + c0) * attrs + e0;
        float v[4];
        int cell = (int)((float)e0 * rattrs);                     // e0 / attrs (exact after the two corrections; e0 < 32 * attrs)
        if (cell * attrs > e0) cell--;
        if ((cell + 1) * attrs <= e0) cell++;
        int at = e0 - cell * attrs;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = (e0 + k < run) ? hl[cell * LD + a * attrs + at] : 0.f;
            if (late) v[k] *= mulv[a * attrs + at];
            if (++at == attrs) { at = 0; cell++; }
        }
        if (e0 + 3 < run) { const f4u w = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f4u*>(dst) = w; }
        else for (int k = 0; e0 + k < run; k++) dst[k] = v[k];
    }
}

// backward: dout [B,na,gs,gs,attrs] fp32 -> dpre [M][ldd] bf16 (GEMM operand; columns >= na*attrs inside the last 8-channel chunk are
// written as zeros, the rest stay zero from allocation), per-workgroup partial column sums of dpre (= bias gradient) and, with
// ImplicitM, of dout*pre.  One workgroup = one image x HEAD_CPB cells, walked as sub-tiles of TC cells staged in LDS; the column sums
// live in registers across the sub-tiles (4 channels x every other cell per thread), one partial row per workgroup as before.
#define HEAD_CPB 128
template <int MAXQ>                                           // channel quads per thread of the column sums: 1 (C <= 512: every head of the reference) or 4 (C <= 2048)
__global__ __launch_bounds__(256) void head_finish_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ pre, int ldp,
                                                              const float* __restrict__ mul, int B, int gs, int na, int attrs, int TC,
                                                              bf16_t* __restrict__ dpre, int ldd, float* __restrict__ partial /*[nblk][2][C]*/,
                                                              const float* __restrict__ objgrad, const int* __restrict__ owner, int och)
{
    extern __shared__ float hl[];                              // t[TC][C + 1], then mulv[C]
    const int cells = gs * gs, C = na * attrs, LD = C + 1;
    float* const mulv = hl + TC * LD;
    const int ncb = (cells + HEAD_CPB - 1) / HEAD_CPB;
    const int b = blockIdx.x / ncb, cb = blockIdx.x - b * ncb;
    const int cbase = cb * HEAD_CPB;
    const int nblock = min(HEAD_CPB, cells - cbase);
    for (int ch = threadIdx.x; ch < C; ch += 256) mulv[ch] = mul ? mul[ch] : 1.f;
    // column sums: thread (cl, c4) owns channels 4*c4 .. 4*c4+3 of the cells with (cell & 1) == cl; loop when C > 512
    const int c4n = (C + 3) >> 2;
    const int cl = threadIdx.x & 1, c4l = threadIdx.x >> 1;       // 128 channel-quad lanes x 2 cell lanes
    float sb[MAXQ][4], sm[MAXQ][4];
#pragma unroll
    for (int j = 0; j < MAXQ; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) { sb[j][k] = 0.f; sm[j][k] = 0.f; }
    const float rattrs = 1.0f / (float)attrs;
    const bool pre4 = mul && ((ldp & 3) == 0) && (reinterpret_cast<uintptr_t>(pre) & 15) == 0;
    for (int s0 = 0; s0 < nblock; s0 += TC) {
        const int c0 = cbase + s0, ncell = min(TC, nblock - s0);
        const int64_t m0 = (int64_t)b * cells + c0;
        __syncthreads();                                          // previous sub-tile fully consumed (and mulv visible)
        const int run = ncell * attrs, nq = (run + 3) >> 2;
        if (objgrad) {
            // r05, SPARSE description of dout (ryolo_head_finish_bwd_sparse): a cell no target was matched to is zero except its objectness
            // element, which the loss also leaves in the compact [B, na, cells] array `objgrad`; matched cells (owner >= 0: a few thousand of
            // 15 M) keep their dense 88-byte rows.  8 bytes read per (anchor, cell) instead of 4 * attrs.
            for (int i = threadIdx.x; i < ncell * LD; i += 256) hl[i] = 0.f;
            __syncthreads();
            for (int i = threadIdx.x; i < na * ncell; i += 256) {
                const int a = i / ncell, cell = i - a * ncell;
                const int64_t idx = ((int64_t)b * na + a) * cells + c0 + cell;
                float* row = hl + cell * LD + a * attrs;
                row[och] = objgrad[idx];
                if (owner[idx] >= 0) {
                    const float* src = dout + idx * attrs;
                    for (int e = 0; e < attrs; e++)
                        if (e != och) row[e] = src[e];
                }
            }
        } else
        // four 16-byte loads of a thread are requested before the first one is scattered: with one load per loop trip every trip paid its
        // own memory round trip, and the few workgroups of the small scales (40 at 8 images x 25^2) made that the launch time
        for (int i0 = threadIdx.x; i0 < na * nq; i0 += 256 * 4) {
            float v[4][4];
            int aa[4], ee[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * 256;
                aa[u] = -1;
#pragma unroll
                for (int k = 0; k < 4; k++) v[u][k] = 0.f;
                if (i >= na * nq) continue;
                const int a = i / nq, q = i - a * nq;
                const int e0 = q * 4;
                aa[u] = a;
                ee[u] = e0;
                const float* src = dout + (((int64_t)b * na + a) * cells + c0) * attrs + e0;
                if (e0 + 3 < run) { const f4u w = *reinterpret_cast<const f4u*>(src); v[u][0] = w.x; v[u][1] = w.y; v[u][2] = w.z; v[u][3] = w.w; }
                else for (int k = 0; e0 + k < run; k++) v[u][k] = src[k];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (aa[u] < 0) continue;
                const int e0 = ee[u];
                int cell = (int)((float)e0 * rattrs);
                if (cell * attrs > e0) cell--;
                if ((cell + 1) * attrs <= e0) cell++;
                int at = e0 - cell * attrs;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (e0 + k < run) hl[cell * LD + aa[u] * attrs + at] = v[u][k];
                    if (++at == attrs) { at = 0; cell++; }
                }
            }
        }
        __syncthreads();
        // dpre rows: 8 channels = one 16-byte store
        const int c8n = (C + 7) >> 3;
        for (int i = threadIdx.x; i < ncell * c8n; i += 256) {
            const int cell = i / c8n, c8 = i - cell * c8n;
            const float* t = hl + cell * LD + c8 * 8;
            float g[8];
#pragma unroll
            for (int k = 0; k < 8; k++) g[k] = (c8 * 8 + k < C) ? t[k] * mulv[c8 * 8 + k] : 0.f;
            const uint4 w = make_uint4(pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]), pack_bf2(g[4], g[5]), pack_bf2(g[6], g[7]));
            bf16_t* o = dpre + (m0 + cell) * ldd + c8 * 8;
            if (c8 * 8 + 8 <= ldd && (reinterpret_cast<uintptr_t>(o) & 15) == 0) *reinterpret_cast<uint4*>(o) = w;
            else for (int k = 0; k < 8 && c8 * 8 + k < C; k++) o[k] = f2bf(g[k]);
        }
        // column sums of this sub-tile
#pragma unroll
        for (int j = 0; j < MAXQ; j++) {
            const int c4 = c4l + j * 128;
            if (c4 >= c4n) break;
            // cells in groups of 8 per thread: the `pre` rows of a group are requested together (same reason as above), the sums stay in
            // ascending cell order
            for (int cg = cl; cg < ncell; cg += 16) {
                float pv[8][4];
#pragma unroll
                for (int u = 0; u < 8; u++) {
#pragma unroll
                    for (int k = 0; k < 4; k++) pv[u][k] = 0.f;
                    const int cell = cg + 2 * u;
                    if (mul && cell < ncell) {
                        const float* pp = pre + (m0 + cell) * ldp + c4 * 4;
                        if (pre4 && c4 * 4 + 3 < C) { const float4 w = *reinterpret_cast<const float4*>(pp); pv[u][0] = w.x; pv[u][1] = w.y; pv[u][2] = w.z; pv[u][3] = w.w; }
                        else for (int k = 0; k < 4 && c4 * 4 + k < C; k++) pv[u][k] = pp[k];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int cell = cg + 2 * u;
                    if (cell >= ncell) continue;
                    const float* t = hl + cell * LD + c4 * 4;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (c4 * 4 + k < C) {
                            const float d = t[k];
                            sb[j][k] += bf2f(f2bf(d * mulv[c4 * 4 + k]));      // bias gradient = column sum of the bf16 operand the wgrad GEMM reads
                            sm[j][k] = fmaf(d, pv[u][k], sm[j][k]);               // (explicit: the sparse kernel must round the same way)
                        }
                    }
                }
            }
        }
    }
    // fold the two cell lanes (adjacent lanes of one wave) and write the partial row
#pragma unroll
    for (int j = 0; j < MAXQ; j++) {
        const int c4 = c4l + j * 128;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float a0 = sb[j][k] + __shfl_xor(sb[j][k], 1, 64);
            const float a1 = sm[j][k] + __shfl_xor(sm[j][k], 1, 64);
            if (cl == 0 && c4 < c4n && c4 * 4 + k < C) {
                partial[((int64_t)blockIdx.x * 2 + 0) * C + c4 * 4 + k] = a0;
                partial[((int64_t)blockIdx.x * 2 + 1) * C + c4 * 4 + k] = a1;
            }
        }
    }
}

// r05: the head backward over the SPARSE description of dout the fused loss leaves behind (ryolo_head_finish_bwd_sparse) without staging dense
// tiles: an unmatched cell's row of dout is zero except its objectness element, so its dpre row is a constant pattern with one value per anchor,
// the bias / ImplicitM column sums of the objectness columns run over the compact arrays (objgrad, preobj: [B, na, cells]) and every other column
// only sees the matched cells (owner >= 0; bit masks per anchor).  HBM per row of dpre: its 16-byte stores + 12 bytes per anchor, instead of the
// 4 * C bytes of dout and 4 * ldp bytes of pre.  Same workgroup -> cells mapping (HEAD_CPB), same partial rows, same summation order per column
// (cells of one parity ascending, the two parities added last) as head_finish_bwd_kernel: results are bit-identical to the dense pass.
// Needs C <= 512 (one 8-channel chunk per lane) and na <= HS_MAXNA; anything else takes the dense kernel's sparse fill.
#define HS_MAXNA 32
#define HS_LDA (HEAD_CPB + 1)
__global__ __launch_bounds__(256) void head_bwd_sparse_kernel(const float* __restrict__ dout, const float* __restrict__ objgrad,
                                                              const float* __restrict__ preobj, const int* __restrict__ owner, int och,
                                                              const float* __restrict__ pre, int ldp, const float* __restrict__ mul, int B, int gs, int na,
                                                              int attrs, bf16_t* __restrict__ dpre, int ldd, float* __restrict__ partial /*[nblk][2][C]*/)
{
    extern __shared__ float hl[];                              // og[na][HS_LDA], po[na][HS_LDA], mulv[C], rw[4][C], mask[na][2] (64-bit)
    const int cells = gs * gs, C = na * attrs;
    float* const og = hl;
    float* const po = og + na * HS_LDA;
    float* const mulv = po + na * HS_LDA;
    float* const rw = mulv + C;
    unsigned long long* const mask = reinterpret_cast<unsigned long long*>(rw + 4 * C + ((na * 2 * HS_LDA + 5 * C) & 1));   // (8-byte aligned)
    const int ncb = (cells + HEAD_CPB - 1) / HEAD_CPB;
    const int b = blockIdx.x / ncb, cb = blockIdx.x - b * ncb;
    const int cbase = cb * HEAD_CPB;
    const int nblock = min(HEAD_CPB, cells - cbase);
    const int64_t m0 = (int64_t)b * cells + cbase;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int ch = threadIdx.x; ch < C; ch += 256) mulv[ch] = mul ? mul[ch] : 1.f;
    for (int it = wave; it < na * 2; it += 4) {
        const int a = it >> 1, cell = (it & 1) * 64 + lane;
        const bool valid = cell < nblock;
        const int64_t idx = ((int64_t)b * na + a) * cells + cbase + cell;
        og[a * HS_LDA + cell] = valid ? objgrad[idx] : 0.f;
        po[a * HS_LDA + cell] = (valid && mul) ? preobj[idx] : 0.f;
        const unsigned long long m = __ballot(valid && owner[idx] >= 0);
        if (lane == 0) mask[it] = m;
    }
    __syncthreads();
    unsigned long long cm0 = 0, cm1 = 0;                        // cells of this block with a matched anchor
    for (int a = 0; a < na; a++) { cm0 |= mask[2 * a]; cm1 |= mask[2 * a + 1]; }

    // ---- dpre rows: lane = 8-channel chunk, a wave walks every 4th cell
    const int c8n = (C + 7) >> 3;
    {
        const int c8 = lane;
        float zf[8], m1 = 0.f, m2 = 0.f;                          // this chunk of an unmatched row with the objectness gradients at zero; the (<= 2: attrs >= 7)
        int k1 = -1, k2 = -1, a1 = 0, a2 = 0;                     // objectness columns inside the chunk
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int col = c8 * 8 + k;
            zf[k] = 0.f;
            if (c8 < c8n && col < C) {
                zf[k] = 0.f * mulv[col];                          // (the dense pass multiplies its zeros too: -0 where ImplicitM is negative)
                const int a = col / attrs;
                if (col - a * attrs == och) {
                    if (k1 < 0) { k1 = k; a1 = a; m1 = mulv[col]; } else { k2 = k; a2 = a; m2 = mulv[col]; }
                }
            }
        }
        float* const row = rw + wave * C;
        for (int cell = wave; cell < nblock; cell += 4) {
            const bool matched = ((cell < 64 ? cm0 >> cell : cm1 >> (cell - 64)) & 1ull) != 0;      // (wave-uniform)
            float g[8];
            if (!matched) {
                const float v1 = k1 >= 0 ? og[a1 * HS_LDA + cell] * m1 : 0.f;
                const float v2 = k2 >= 0 ? og[a2 * HS_LDA + cell] * m2 : 0.f;
#pragma unroll
                for (int k = 0; k < 8; k++) g[k] = k == k1 ? v1 : (k == k2 ? v2 : zf[k]);
            } else {
                // a matched anchor's row comes from dout, the other anchors of the cell from the compact array
                for (int col = lane; col < C; col += 64) {
                    const int a = col / attrs, e = col - a * attrs;
                    const bool dense = ((mask[2 * a + (cell >> 6)] >> (cell & 63)) & 1ull) != 0;
                    float v = e == och ? og[a * HS_LDA + cell] : 0.f;
                    if (dense && e != och) v = dout[(((int64_t)b * na + a) * cells + cbase + cell) * attrs + e];
                    row[col] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 8; k++) g[k] = (c8 < c8n && c8 * 8 + k < C) ? row[c8 * 8 + k] * mulv[c8 * 8 + k] : 0.f;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (c8 < c8n) {
                const uint4 w = make_uint4(pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]), pack_bf2(g[4], g[5]), pack_bf2(g[6], g[7]));
                bf16_t* o = dpre + (m0 + cell) * ldd + c8 * 8;
                if (c8 * 8 + 8 <= ldd && (reinterpret_cast<uintptr_t>(o) & 15) == 0) *reinterpret_cast<uint4*>(o) = w;
                else for (int k = 0; k < 8 && c8 * 8 + k < C; k++) o[k] = f2bf(g[k]);
            }
        }
    }

    // ---- column sums.  Objectness columns: thread (anchor, parity) over the compact arrays (matched cells carry the same objectness element)
    float* const p0 = partial + (int64_t)blockIdx.x * 2 * C;
    if ((int)threadIdx.x < 2 * na) {                              // (2 * na <= 64: adjacent lanes of one wave hold the two parities)
        const int a = threadIdx.x >> 1, par = threadIdx.x & 1;
        const int col = a * attrs + och;
        const float mv = mulv[col];
        float sb = 0.f, sm = 0.f;
        for (int cell = par; cell < nblock; cell += 2) {
            const float d = og[a * HS_LDA + cell];
            sb += bf2f(f2bf(d * mv));
            sm = fmaf(d, po[a * HS_LDA + cell], sm);
        }
        const float a0 = sb + __shfl_xor(sb, 1, 64);
        const float a1 = sm + __shfl_xor(sm, 1, 64);
        if (par == 0) { p0[col] = a0; p0[C + col] = a1; }
    }
    // every other column: the matched cells of its anchor, ascending per parity (none in most blocks: zeros)
    for (int col = threadIdx.x; col < C; col += 256) {
        const int a = col / attrs, e = col - a * attrs;
        if (e == och) continue;
        float sb[2] = {0.f, 0.f}, sm[2] = {0.f, 0.f};
        const float mv = mulv[col];
        for (int h = 0; h < 2; h++) {
            unsigned long long m = mask[2 * a + h];
            while (m) {
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int cell = h * 64 + bit;
                const float d = dout[(((int64_t)b * na + a) * cells + cbase + cell) * attrs + e];
                const float pv = mul ? pre[(m0 + cell) * ldp + col] : 0.f;
                sb[cell & 1] += bf2f(f2bf(d * mv));
                sm[cell & 1] = fmaf(d, pv, sm[cell & 1]);
            }
        }
        p0[col] = sb[0] + sb[1];
        p0[C + col] = sm[0] + sm[1];
    }
}

// out0[c] += sum_r partial[r][0][c];  out1[c] += sum_r partial[r][1][c]   (rows already folded to <= 256; double accumulation)
__global__ __launch_bounds__(256) void head_grad_rows_kernel(const float* __restrict__ partial, int rows, int C, float* __restrict__ out0,
                                                             float* __restrict__ out1)
{
    // 32 columns x 8 row groups per workgroup, groups folded in a fixed order (one thread per column walking all rows: 17-23 us per head for a few
    // hundred kilobytes — dependent loads)
    __shared__ double red[8][33];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float* const out = c < C ? out0 : out1;
    const bool live = c < 2 * C && out != nullptr;
    double s = 0.0;
    if (live)
        for (int r = g; r < rows; r += 8) s += (double)partial[(int64_t)r * 2 * C + c];
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && live) {
        double t = red[0][cl];
#pragma unroll
        for (int q = 1; q < 8; q++) t += red[q][cl];
        out[c < C ? c : c - C] += (float)t;
    }
}

__global__ void colsum_rows_kernel(const float* __restrict__ partial, int rows, int C, int Cvalid, float* __restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cvalid) return;
    double s = 0.0;
    for (int r = 0; r < rows; r++) s += (double)partial[(int64_t)r * C + c];
    out[c] += (float)s;
}

// ImplicitA: z = x + a[c]  (model/utils.py:172-173); backward of `a` = column sums of dz
__global__ __launch_bounds__(256) void chan_add_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ a, int64_t M, int C,
                                                       bf16_t* __restrict__ z, int ldz)
{
    const int c8 = C >> 3;
    const int64_t total = M * c8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / c8;
        const int c = (int)(i - m * c8) << 3;
        V8 v = ld8(x + m * ldx + c);
#pragma unroll
        for (int k = 0; k < 8; k++) v.v[k] += a[c + k];
        st8(z + m * ldz + c, v);
    }
}

__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, int ldx, int64_t M, int C, int rows_per_block,
                                                          float* __restrict__ partial)
{
    // C is a multiple of 8 here (padded head widths / ImplicitA channels): 8 channels per thread, 16-byte loads
    __shared__ float red[256][8 + 1];
    const int c8 = C >> 3;
    const int cols = c8 < 256 ? c8 : 256, nrl = 256 / cols;
    const int rl = threadIdx.x / cols, cl = threadIdx.x - rl * cols;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int cb = 0; cb < c8; cb += cols) {
        const int cc = cb + cl;
        float s[8];
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = 0.f;
        if (rl < nrl && cc < c8)
            for (int64_t m = r0 + rl; m < r1; m += (int64_t)nrl * 4) {          // four rows requested before the first is added (order of the sums unchanged)
                V8 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int64_t mm = m + (int64_t)u * nrl;
                    if (mm < r1) v[u] = ld8(x + mm * ldx + (cc << 3));
                    else
#pragma unroll
                        for (int k = 0; k < 8; k++) v[u].v[k] = 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int k = 0; k < 8; k++) s[k] += v[u].v[k];
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) red[threadIdx.x][k] = s[k];
        __syncthreads();
        if (rl == 0 && cc < c8)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float t = 0.f;
                for (int j = 0; j < nrl; j++) t += red[j * cols + cl][k];
                partial[(int64_t)blockIdx.x * C + (cc << 3) + k] = t;
            }
    }
}

// ------------------------------------------------------------------------------------------------ weights / optimizer
// fp32 master [Cout][Cin][taps] (torch layout) -> Wf bf16 [Cout][taps][CinP] and Wd bf16 [Cin][taps][Cout] (Wd may be null).
// CinP >= Cin: small-Cin first layers are packed as a single tap with k = tap*Cin + c zero-padded to CinP.

// one (32 output channels x 32 input channels) tile of pack_weights_kernel; TAPS = 0: tap count from the entry
template <int TAPS>
__device__ __forceinline__ void pack_tile(const PackEntry& e, float (*tile)[32 * RY_MAX_TAPS + 1], int64_t t)
{
    const int taps = TAPS ? TAPS : e.taps;
    const int nct = e.Cin / 32;
    const int co0 = (int)(t / nct) * 32, c0 = (int)(t % nct) * 32;
    const int run = 32 * taps;
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
        const int r = i / run, k = i - r * run;
        tile[r][k] = (co0 + r < e.Cout) ? e.src[((int64_t)(co0 + r) * e.Cin + c0) * taps + k] : 0.f;
    }
    __syncthreads();
    // both images are written as PAIRS (4-byte stores: 16 lanes cover a 64-byte segment; the 2-byte stores of r01-r04 moved 128 bytes per wave
    // instruction)
    const int run2 = 16 * taps;
    for (int i = threadIdx.x; i < 32 * run2; i += 256) {         // Wf[co][t][c]: c fastest
        const int c = (i & 15) * 2, tp = (i >> 4) % taps, r = i / run2;
        if (co0 + r < e.Cout)
            *reinterpret_cast<unsigned*>(e.wf + ((int64_t)(co0 + r) * taps + tp) * e.CinP + c0 + c) = pack_bf2(tile[r][c * taps + tp], tile[r][(c + 1) * taps + tp]);
    }
    if (e.wd) {
        const int ldw = e.ldWd ? e.ldWd : e.CoutP;
        const bool pairs = ((ldw | e.Cout) & 1) == 0 && (reinterpret_cast<uintptr_t>(e.wd) & 3) == 0;
        for (int i = threadIdx.x; i < 32 * run2; i += 256) {     // Wd[c][t][co]: co fastest
            const int r = (i & 15) * 2, tp = (i >> 4) % taps, c = i / run2;
            if (co0 + r >= e.Cout) continue;
            float v0 = tile[r][c * taps + tp], v1 = tile[r + 1][c * taps + tp];
            if (e.wd_scale) { v0 *= e.wd_scale[co0 + r]; if (co0 + r + 1 < e.Cout) v1 *= e.wd_scale[co0 + r + 1]; }
            bf16_t* o = e.wd + ((int64_t)(c0 + c) * taps + tp) * ldw + co0 + r;
            if (pairs) *reinterpret_cast<unsigned*>(o) = pack_bf2(v0, v1);
            else { o[0] = f2bf(v0); if (co0 + r + 1 < e.Cout) o[1] = f2bf(v1); }
        }
    }
}

// Tiled repack: one workgroup per (conv, 32 output channels, 32 input channels).  Reads are whole contiguous runs of
// 32*taps floats per output channel, the [co][c][t] block is transposed in LDS, and both bf16 images are written as
// 64-byte segments (Wf along c, Wd along co).  `start` of an entry = index of its first tile.
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackEntry* __restrict__ table, int n, int64_t total_tiles)
{
    __shared__ float tile[32][32 * RY_MAX_TAPS + 1];
    for (int64_t b = blockIdx.x; b < total_tiles; b += gridDim.x) {
        int lo = 0, hi = n - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (table[mid].start <= b) lo = mid; else hi = mid - 1; }
        const PackEntry e = table[lo];
        const int64_t t = b - e.start;
        if (e.CinP != e.Cin) {
            // small-Cin stem: single-tap image [Cout][CinP], k = tap*Cin + c (zero padded); one tile = 256 elements
            const int64_t j = t * 256 + threadIdx.x;
            if (j < (int64_t)e.Cout * e.CinP) {
                const int k = (int)(j % e.CinP), co = (int)(j / e.CinP);
                float v = 0.f;
                if (k < e.taps * e.Cin) { const int tp = k / e.Cin, c = k - tp * e.Cin; v = e.src[((int64_t)co * e.Cin + c) * e.taps + tp]; }
                e.wf[j] = f2bf(v);
            }
            continue;
        }
        if (e.taps == 9) pack_tile<9>(e, tile, t);              // (uniform per tile; with the tap count a runtime value every element paid three
        else if (e.taps == 1) pack_tile<1>(e, tile, t);         //  integer divisions: the kernel was ALU-bound at 1.5 TB/s)
        else pack_tile<0>(e, tile, t);
    }
}

// Inference re-parameterisation of RepConv (model/utils.py:189-215; the reference never fuses): with eval-mode
// BatchNorm both branches are affine, so  act(bn3(conv3x3(x)) + bn1(conv1x1(x))) = act(conv3x3'(x) + shift)  with
// W'[co][t][c] = s3[co] W3[co][c][t] + [t == centre] s1[co] W1[co][c]  and  shift = shift3 + shift1.  One 3x3 GEMM with the
// folded-BN epilogue replaces two GEMMs and a two-branch BN + activation pass.  coa / cob: [4][Cout] from bn_eval_coeffs.
__global__ void repconv_fold_kernel(const float* __restrict__ w3, const float* __restrict__ w1, const float* __restrict__ coa,
                                    const float* __restrict__ cob, int Cout, int Cin, bf16_t* __restrict__ wf, float* __restrict__ co_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)Cout * 9 * Cin;
    if (i < Cout) {
        co_out[0 * Cout + i] = 0.f;
        co_out[1 * Cout + i] = 1.f;
        co_out[2 * Cout + i] = 1.f;
        co_out[3 * Cout + i] = coa[3 * Cout + i] + cob[3 * Cout + i];
    }
    if (i >= total) return;
    const int c = (int)(i % Cin), t = (int)((i / Cin) % 9), co = (int)(i / ((int64_t)9 * Cin));
    float v = coa[2 * Cout + co] * w3[((int64_t)co * Cin + c) * 9 + t];
    if (t == 4) v += cob[2 * Cout + co] * w1[(int64_t)co * Cin + c];
    wf[i] = f2bf(v);
}

// dW scratch [Cout][CinP] (single-tap layout of a small-Cin layer) -> += into torch layout [Cout][Cin][taps]
__global__ void unpack_wgrad_kernel(const float* __restrict__ scratch, int Cout, int Cin, int taps, int CinP, float* __restrict__ grad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * Cin * taps) return;
    const int t = i % taps;
    const int c = (i / taps) % Cin;
    const int co = i / (taps * Cin);
    grad[i] += scratch[(int64_t)co * CinP + t * Cin + c];
}

// SGD with Nesterov momentum, dampening 0, no weight decay (train.py:156): buf = mu*buf + g; p -= lr*(g + mu*buf)
__global__ __launch_bounds__(256) void sgd_nesterov_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                                                           int64_t n4, float lr, float mu, float gscale, int zero_grad)
{
    // 16-byte lanes; the flat buffers are 256-byte aligned and padded to a multiple of 64 floats
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* b4 = reinterpret_cast<float4*>(buf);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 gv = g4[i], bv = b4[i], pv = p4[i];
        gv.x *= gscale; gv.y *= gscale; gv.z *= gscale; gv.w *= gscale;
        bv.x = mu * bv.x + gv.x; bv.y = mu * bv.y + gv.y; bv.z = mu * bv.z + gv.z; bv.w = mu * bv.w + gv.w;
        pv.x -= lr * (gv.x + mu * bv.x); pv.y -= lr * (gv.y + mu * bv.y); pv.z -= lr * (gv.z + mu * bv.z); pv.w -= lr * (gv.w + mu * bv.w);
        b4[i] = bv;
        p4[i] = pv;
        if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);       // optimizer.zero_grad() fused (train.py:202)
    }
}

// torch.optim.Adam defaults of train.py:154 (betas given by the caller, eps, no weight decay, no amsgrad), single pass over the flat buffers:
//   m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// bias corrections arrive as the two host-computed scalars torch forms in double (step_size, sqrt(bias_correction2)).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n4, float step_size, float bc2s, float w1, float b2, float w2, float eps, float gscale,
                                                   int zero_grad)
{
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 gv = g4[i];
        float4 mv = m4[i], vv = v4[i], pv = p4[i];
        const float ge[4] = {gv.x * gscale, gv.y * gscale, gv.z * gscale, gv.w * gscale};
        float* me = &mv.x; float* ve = &vv.x; float* pe = &pv.x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            me[k] = me[k] + w1 * (ge[k] - me[k]);                       // exp_avg.lerp_(grad, 1 - beta1)
            ve[k] = ve[k] * b2 + w2 * (ge[k] * ge[k]);                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
            pe[k] -= step_size * (me[k] / (sqrtf(ve[k]) / bc2s + eps));   // denom = (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
        }
        m4[i] = mv;
        v4[i] = vv;
        p4[i] = pv;
        if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// fp32 NCHW image -> is consumed directly by im2col_kernel; nothing else needed for the input side.

// ------------------------------------------------------------------------------------------------ C ABI
static inline unsigned grid_rows(int64_t M, int C)
{
    const int c8 = C >> 3, cols = c8 < 256 ? c8 : 256, rpi = 256 / cols;
    // 4 rows per thread and NO small cap on the grid: with the 8192-workgroup cap of the first three rounds a thread of a 400^2 layer walked ~40
    // rows that lie gridDim * rpi rows apart, the resident workgroups drifted apart and the requests in flight at any moment were scattered
    // over the whole tensor; workgroups that each finish after 4 rows keep them inside a window that moves through memory in dispatch order:
    // forward 5.0-5.5 -> 6.2 TB/s, backward apply 5.1-5.7 -> 6.3-7.0 TB/s isolated (tools/bench_bnact.py; 2 rows: same forward, apply -6 %;
    // 1 row: apply 4.5 TB/s; 8 rows: -3 %).  RYOLO_EW_GRID / RYOLO_EW_ROWS: A/B knobs.
    static const int cap = getenv("RYOLO_EW_GRID") ? atoi(getenv("RYOLO_EW_GRID")) : (1 << 20);
    static const int rpt = getenv("RYOLO_EW_ROWS") ? atoi(getenv("RYOLO_EW_ROWS")) : 4;
    int64_t g = ry_cdiv(M, (int64_t)rpi * rpt);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}
static inline unsigned grid_for(int64_t work_items)
{
    static const int cap = getenv("RYOLO_EW_GRID2") ? atoi(getenv("RYOLO_EW_GRID2")) : 8192;     // A/B knob
    int64_t g = ry_cdiv(work_items, 256);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// Statistics of channels [c0, c0 + C) of partial rows that are [2][ld] wide (several BatchNorms behind ONE GEMM launch: sibling
// convolutions of a block that read the same input are emitted as one GEMM with concatenated output channels).  coeffs: this
// BatchNorm's own [4][C].  Every slice call folds the full-width rows again when there are many (a few microseconds).
extern "C" int ryolo_bn_finalize_slice(const float* partial, int rows, int ld, int c0, int C, double count, float eps, float momentum,
                                       const float* gamma, const float* beta, float* running_mean, float* running_var, float* coeffs,
                                       hipStream_t stream)
{
    if (!partial || !gamma || !beta || !coeffs || C <= 0 || rows <= 0 || c0 < 0 || c0 + C > ld) return RY_ERR_ARG;
    // very many rows: folded into a scratch area the caller appends to the partial buffer (rows + FOLD_S rows allocated); up to FOLD_DIRECT rows
    // the finalize kernel's 128 row lanes sum them directly (one launch instead of two)
    if (rows > bn_fold_direct()) {
        float* scratch = const_cast<float*>(partial) + (int64_t)rows * 2 * ld;
        partial = fold_rows(partial, rows, 2 * ld, scratch, stream);
    }
    if (rows > 4 * FOLD_S)
        hipLaunchKernelGGL(bn_finalize_kernel<8>, dim3((unsigned)ry_cdiv(C, 8)), dim3(1024), 0, stream, partial, rows, ld, c0, C, count, eps,
                           momentum, gamma, beta, running_mean, running_var, coeffs);
    else
        hipLaunchKernelGGL(bn_finalize_kernel<32>, dim3((unsigned)ry_cdiv(C, 32)), dim3(1024), 0, stream, partial, rows, ld, c0, C, count, eps,
                           momentum, gamma, beta, running_mean, running_var, coeffs);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_bn_finalize(const float* partial, int rows, int C, double count, float eps, float momentum, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, float* coeffs, hipStream_t stream)
{
    return ryolo_bn_finalize_slice(partial, rows, C, 0, C, count, eps, momentum, gamma, beta, running_mean, running_var, coeffs, stream);
}

// coeffs [4][ld]: this BatchNorm fills channels [c0, c0 + C) of every row
extern "C" int ryolo_bn_eval_coeffs_slice(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                                          float* coeffs, int ld, int c0, hipStream_t stream)
{
    if (!gamma || !beta || !rm || !rv || !coeffs || C <= 0 || c0 < 0 || c0 + C > ld) return RY_ERR_ARG;
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((unsigned)ry_cdiv(C, 256)), dim3(256), 0, stream, gamma, beta, rm, rv, eps, C, coeffs,
                       ld, c0);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                                    float* coeffs, hipStream_t stream)
{
    return ryolo_bn_eval_coeffs_slice(gamma, beta, rm, rv, eps, C, coeffs, C, 0, stream);
}

static int check_bnact(const BnActParams& p) { return (!p.y1 || !p.co1 || p.C <= 0 || (p.C & 7) || (p.ld1 & 7) || p.M < 0) ? RY_ERR_ARG : RY_OK; }

extern "C" int ryolo_bn_act_fwd(const BnActParams* pp, hipStream_t stream)
{
    if (!pp || check_bnact(*pp) || !pp->z) return RY_ERR_ARG;
    if (pp->M == 0) return RY_OK;
    {
        const BnActParams& p = *pp;
        const dim3 g(grid_rows(p.M, p.C)), b(256);
#define RY_FWD(ACT)                                                                                                   \
    if (p.y2 && p.res) hipLaunchKernelGGL((bn_act_fwd_kernel<ACT, true, true>), g, b, 0, stream, p);                  \
    else if (p.y2) hipLaunchKernelGGL((bn_act_fwd_kernel<ACT, true, false>), g, b, 0, stream, p);                     \
    else if (p.res) hipLaunchKernelGGL((bn_act_fwd_kernel<ACT, false, true>), g, b, 0, stream, p);                    \
    else hipLaunchKernelGGL((bn_act_fwd_kernel<ACT, false, false>), g, b, 0, stream, p);
        switch (p.act) {
            case ACT_MISH: RY_FWD(ACT_MISH) break;
            case ACT_LEAKY: RY_FWD(ACT_LEAKY) break;
            case ACT_SILU: RY_FWD(ACT_SILU) break;
            default: RY_FWD(ACT_LINEAR) break;
        }
#undef RY_FWD
    }
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_bn_act_bwd_blocks(int64_t M, int C, int* nblk, int* rows_per_block)
{
    if (!nblk || !rows_per_block || M <= 0 || C <= 0) return RY_ERR_ARG;
    const int c8 = C >> 3;
    const int cols = c8 < 256 ? c8 : 256;
    const int nrl = 256 / cols;
    int64_t blocks = ry_cdiv(M, (int64_t)nrl * 8);                  // >= 8 rows per row lane
    // 1280 = 5 resident workgroups x 256 CUs: one round of workgroups, each walking one contiguous run of rows, and a third of the partial rows
    // for the finalize kernel (r04 kernel, same box, alternating runs: 856 img/s at 1280 vs 853 at 4096 / 2560 / 2048, 850 at 640; with the r02
    // kernel — 3 waves per SIMD, no read-ahead — 2048 blocks had been 3 ms/step slower than 4096).  RYOLO_BN_RED_BLOCKS: A/B knob.
    static const int cap = getenv("RYOLO_BN_RED_BLOCKS") ? atoi(getenv("RYOLO_BN_RED_BLOCKS")) : 1280;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    *rows_per_block = (int)ry_cdiv(M, blocks);
    *nblk = (int)ry_cdiv(M, *rows_per_block);
    return RY_OK;
}

// reduce -> finalize -> apply; `partial` needs nblk*K*C floats, bco needs 3*C floats
extern "C" int ryolo_bn_act_bwd(const BnActParams* pp, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, float* bco,
                                int frozen, hipStream_t stream)
{
    if (!pp || check_bnact(*pp) || !pp->dz || !pp->partial || !bco) return RY_ERR_ARG;
    if (!pp->dy1 && (pp->y2 || pp->dres)) return RY_ERR_ARG;     // dy1 == null: statistics only (the consumer applies them itself: stem wgrad)
    BnActParams p = *pp;
    if (p.M == 0) return RY_OK;
    int nblk, rpb;
    ryolo_bn_act_bwd_blocks(p.M, p.C, &nblk, &rpb);
    p.rows_per_block = rpb;
    p.bco = bco;
    const int K = p.y2 ? 3 : 2;
#define RY_RED(ACT)                                                                                              \
    if (p.y2) hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<ACT, true>), dim3(nblk), dim3(256), 0, stream, p);        \
    else hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<ACT, false>), dim3(nblk), dim3(256), 0, stream, p);
    switch (p.act) {
        case ACT_MISH: RY_RED(ACT_MISH) break;
        case ACT_LEAKY: RY_RED(ACT_LEAKY) break;
        case ACT_SILU: RY_RED(ACT_SILU) break;
        default: RY_RED(ACT_LINEAR) break;
    }
#undef RY_RED
    int frows = nblk;
    const float* fpart = p.partial;
    if (nblk > bn_fold_direct()) fpart = fold_rows(p.partial, frows, K * p.C, p.partial + (int64_t)nblk * K * p.C, stream);   // caller allocates nblk + 64 rows
#define RY_FIN(KK, CH)                                                                                                                           \
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<KK, CH>), dim3((unsigned)ry_cdiv(p.C, CH)), dim3(1024), 0, stream, fpart, frows, p.C, (double)p.M, \
                       frozen, p.co1, p.co2, bco, dgamma1, dbeta1, dgamma2, dbeta2)
    if (frows > 4 * FOLD_S) { if (K == 3) RY_FIN(3, 8); else RY_FIN(2, 8); }
    else { if (K == 3) RY_FIN(3, 32); else RY_FIN(2, 32); }
#undef RY_FIN
    if (p.dy1) {
        const dim3 g(grid_rows(p.M, p.C)), b(256);
#define RY_APP(ACT)                                                                                                   \
    if (p.y2 && p.dres) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<ACT, true, true>), g, b, 0, stream, p);           \
    else if (p.y2) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<ACT, true, false>), g, b, 0, stream, p);               \
    else if (p.dres) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<ACT, false, true>), g, b, 0, stream, p);             \
    else hipLaunchKernelGGL((bn_act_bwd_apply_kernel<ACT, false, false>), g, b, 0, stream, p);
        switch (p.act) {
            case ACT_MISH: RY_APP(ACT_MISH) break;
            case ACT_LEAKY: RY_APP(ACT_LEAKY) break;
            case ACT_SILU: RY_APP(ACT_SILU) break;
            default: RY_APP(ACT_LINEAR) break;
        }
#undef RY_APP
    }
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_maxpool_fwd(const PoolParams* pp, hipStream_t stream)
{
    if (!pp || !pp->x || !pp->z || (pp->C & 7) || pp->k < 1 || pp->k > 15) return RY_ERR_ARG;
    if (pp->rowmax && pp->stride == 1 && pp->OH == pp->H && pp->OW == pp->W && (!pp->idx || pp->rowidx)) {
        const dim3 g(grid_for((int64_t)pp->NB * pp->H * pp->W * ((pp->C >> 3) < 256 ? (pp->C >> 3) : 256)));
        hipLaunchKernelGGL(pool_rows_fwd_kernel, g, dim3(256), 0, stream, *pp);
        hipLaunchKernelGGL(pool_cols_fwd_kernel, g, dim3(256), 0, stream, *pp);
        RY_CHECK_LAUNCH();
        return RY_OK;
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((int64_t)pp->NB * pp->OH * pp->OW * ((pp->C >> 3) < 256 ? (pp->C >> 3) : 256))), dim3(256), 0, stream, *pp);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_maxpool_bwd(const PoolParams* pp, hipStream_t stream)
{
    if (!pp || !pp->dz || !pp->dx || !pp->idx || (pp->C & 7)) return RY_ERR_ARG;
    if (pp->rowmax && pp->rowidx && pp->growws && pp->stride == 1 && pp->OH == pp->H && pp->OW == pp->W) {
        const dim3 g(grid_for((int64_t)pp->NB * pp->H * pp->W * ((pp->C >> 3) < 256 ? (pp->C >> 3) : 256)));
        hipLaunchKernelGGL(pool_cols_bwd_kernel, g, dim3(256), 0, stream, *pp);
        hipLaunchKernelGGL(pool_rows_bwd_kernel, g, dim3(256), 0, stream, *pp);
        RY_CHECK_LAUNCH();
        return RY_OK;
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((int64_t)pp->NB * pp->H * pp->W * ((pp->C >> 3) < 256 ? (pp->C >> 3) : 256))), dim3(256), 0, stream, *pp);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_upsample2x_fwd(const UpParams* pp, hipStream_t stream)
{
    if (!pp || !pp->x || !pp->z || (pp->C & 7)) return RY_ERR_ARG;
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(grid_for((int64_t)pp->NB * pp->H * pp->W * 4 * (pp->C >> 3))), dim3(256), 0, stream, *pp);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_upsample2x_bwd(const UpParams* pp, hipStream_t stream)
{
    if (!pp || !pp->x || !pp->z || (pp->C & 7)) return RY_ERR_ARG;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((int64_t)pp->NB * pp->H * pp->W * (pp->C >> 3))), dim3(256), 0, stream, *pp);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_im2col(const float* img, int NB, int Cin, int H, int W, int kh, int kw, int stride, int pad, int OH, int OW, int Kpad,
                            bf16_t* col, hipStream_t stream)
{
    if (!img || !col || (Kpad & 31) || Kpad < kh * kw * Cin) return RY_ERR_ARG;
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((int64_t)NB * OH * OW * (Kpad >> 3))), dim3(256), 0, stream, img, NB, Cin, H, W, kh, kw,
                       stride, pad, OH, OW, Kpad, col);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

static int head_finish_fwd_impl(const float* pre, int ldp, const float* mul, int B, int gs, int na, int attrs, float* out, float* preobj, int och,
                                float* xobj, hipStream_t stream)
{
    if (!pre || !out) return RY_ERR_ARG;
    const int64_t total = (int64_t)B * na * gs * gs * attrs;
    if (total == 0) return RY_OK;
    const int TC = head_tile_cells(na * attrs);
    const size_t lds = ((size_t)TC * (na * attrs + 1) + na * attrs) * sizeof(float);
    const dim3 grid((unsigned)(B * ry_cdiv((int64_t)gs * gs, TC)));
    if (preobj) hipLaunchKernelGGL(head_finish_fwd_kernel<true>, grid, dim3(256), lds, stream, pre, ldp, mul, B, gs, na, attrs, TC, out, preobj, och, xobj);
    else hipLaunchKernelGGL(head_finish_fwd_kernel<false>, grid, dim3(256), lds, stream, pre, ldp, mul, B, gs, na, attrs, TC, out, preobj, och, xobj);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_head_finish_fwd(const float* pre, int ldp, const float* mul, int B, int gs, int na, int attrs, float* out,
                                     hipStream_t stream)
{
    return head_finish_fwd_impl(pre, ldp, mul, B, gs, na, attrs, out, nullptr, 0, nullptr, stream);
}

// the same pass + preobj [B, na, gs, gs] fp32 = pre[.., a * attrs + och] (the objectness column before ImplicitM) for ryolo_head_finish_bwd_sparse
extern "C" int ryolo_head_finish_fwd_obj(const float* pre, int ldp, const float* mul, int B, int gs, int na, int attrs, float* out, int och,
                                         float* preobj, float* xobj, hipStream_t stream)
{
    if (!preobj || och < 0 || och >= attrs) return RY_ERR_ARG;
    return head_finish_fwd_impl(pre, ldp, mul, B, gs, na, attrs, out, preobj, och, xobj, stream);
}

// dbias (conv bias gradient) and dmul (ImplicitM gradient, with mul) are ACCUMULATED; scratch needs
// (B*ceil(gs*gs/128) + 64) * 2 * na*attrs floats; dpre columns >= na*attrs must have been zeroed once by the caller
static int head_finish_bwd_impl(const float* dout, const float* pre, int ldp, const float* mul, int B, int gs, int na, int attrs,
                                bf16_t* dpre, int ldd, float* dbias, float* dmul, float* scratch, const float* objgrad, const int* owner, int och,
                                const float* preobj, hipStream_t stream)
{
    if (!dout || !dpre || !scratch || (mul && (!dmul || !pre))) return RY_ERR_ARG;       // (pre is read for the ImplicitM gradient only)
    if (objgrad && (!owner || och < 0 || och >= attrs)) return RY_ERR_ARG;
    if ((int64_t)B * gs * gs == 0) return RY_OK;
    const int C = na * attrs;
    const int ncb = (int)ry_cdiv((int64_t)gs * gs, HEAD_CPB);
    int rows = B * ncb;
    if (C > 2048) return RY_ERR_UNSUPPORTED;
    const int TC = head_tile_cells(C);
    const size_t lds = ((size_t)TC * (C + 1) + C) * sizeof(float);
    // (with four quads per thread compiled in for every C the kernel held 179 VGPRs = 2 waves per SIMD)
    static const int direct = getenv("RYOLO_HEAD_SPARSE_DIRECT") ? atoi(getenv("RYOLO_HEAD_SPARSE_DIRECT")) : 1;      // 0: the dense kernel's sparse tile fill (what larger heads take)
    if (objgrad && direct && C <= 512 && na <= HS_MAXNA && (preobj || !mul)) {
        const size_t l2 = ((size_t)na * 2 * HS_LDA + 5 * C + 1) * sizeof(float) + (size_t)na * 2 * sizeof(unsigned long long);
        hipLaunchKernelGGL(head_bwd_sparse_kernel, dim3(rows), dim3(256), l2, stream, dout, objgrad, preobj, owner, och, pre, ldp, mul, B, gs, na, attrs,
                           dpre, ldd, scratch);
    } else if (C <= 512) hipLaunchKernelGGL(head_finish_bwd_kernel<1>, dim3(rows), dim3(256), lds, stream, dout, pre, ldp, mul, B, gs, na, attrs, TC, dpre, ldd, scratch, objgrad, owner, och);
    else hipLaunchKernelGGL(head_finish_bwd_kernel<4>, dim3(rows), dim3(256), lds, stream, dout, pre, ldp, mul, B, gs, na, attrs, TC, dpre, ldd, scratch, objgrad, owner, och);
    const float* part = fold_rows(scratch, rows, 2 * C, scratch + (int64_t)rows * 2 * C, stream);
    hipLaunchKernelGGL(head_grad_rows_kernel, dim3((unsigned)ry_cdiv(2 * C, 32)), dim3(256), 0, stream, part, rows, C, dbias, mul ? dmul : nullptr);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_head_finish_bwd(const float* dout, const float* pre, int ldp, const float* mul, int B, int gs, int na, int attrs,
                                     bf16_t* dpre, int ldd, float* dbias, float* dmul, float* scratch, hipStream_t stream)
{
    return head_finish_bwd_impl(dout, pre, ldp, mul, B, gs, na, attrs, dpre, ldd, dbias, dmul, scratch, nullptr, nullptr, 0, nullptr, stream);
}

// the same pass over the SPARSE description of dout the fused loss leaves behind (LossParams.objgrad + ryolo_loss_owner_grids): objgrad
// [B, na, gs, gs] = the objectness element of every cell, owner[cell] >= 0 = the cell's dense row of dout is to be read (matched cells), every
// other element of dout is zero and is NOT read.  och = index of the objectness element inside a row (4 csl, 5 kfiou).  preobj (optional; from
// ryolo_head_finish_fwd_obj) = the objectness column of pre as a compact [B, na, gs, gs] array: with it (and na * attrs <= 512, na <= 32) `pre` is
// read at matched cells only.  Same results bit for bit.
extern "C" int ryolo_head_finish_bwd_sparse(const float* dout, const float* objgrad, const int* owner, int och, const float* preobj, const float* pre,
                                            int ldp, const float* mul, int B, int gs, int na, int attrs, bf16_t* dpre, int ldd, float* dbias,
                                            float* dmul, float* scratch, hipStream_t stream)
{
    if (!objgrad || !owner) return RY_ERR_ARG;
    return head_finish_bwd_impl(dout, pre, ldp, mul, B, gs, na, attrs, dpre, ldd, dbias, dmul, scratch, objgrad, owner, och, preobj, stream);
}

// Detection head with ImplicitM written by the GEMM epilogue (ConvGemmParams.head_attrs): out = (W (x + a) + b) * m  (a = ImplicitA or absent).
// Its backward runs the weight-gradient GEMM on the UNSCALED head gradient and the layer's input x (NOT x + a), G[c][k] = sum_rows dout[., c] x[., k]
// (s[c] = sum_rows dout[., c] for the bias), and this pass finishes the parameter gradients from G without the pre-ImplicitM activations and without
// a pass over the input gradient:
//   Ge = G + s (x) a        (the weight gradient against x + a: a is the same for every row)
//   dW[c][k] += m[c] Ge[c][k]     db[c] += m[c] s[c]     dm[c] += sum_k W[c][k] Ge[c][k] + b[c] s[c]   (= sum_rows dout (W (x + a) + b))
//   da[k]    += sum_c m[c] s[c] W[c][k]                  (= column sums of the input gradient: dx = (dout * m) W)
// and clears G and s for the next step.  One workgroup per output channel; da by a launch of its own in front (fixed summation order).
__global__ __launch_bounds__(256) void head_da_kernel(const float* __restrict__ s, const float* __restrict__ W, const float* __restrict__ m, int Cout, int K,
                                                      float* __restrict__ da)
{
    // 32 input channels x 8 groups of output channels per workgroup (the first cut — one thread per k walking all Cout rows, K / 256 workgroups —
    // took 152 us per head); groups folded in a fixed order
    __shared__ float red[8][33];
    const int kl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + kl;
    float acc = 0.f;
    if (k < K) {
#pragma unroll 8
        for (int c = g; c < Cout; c += 8) acc = fmaf(m[c] * s[c], W[(int64_t)c * K + k], acc);
    }
    red[g][kl] = acc;
    __syncthreads();
    if (g == 0 && k < K) {
        float t = red[0][kl];
#pragma unroll
        for (int q = 1; q < 8; q++) t += red[q][kl];
        da[k] += t;
    }
}

__global__ __launch_bounds__(256) void head_wgrad_finish_kernel(float* __restrict__ G, float* __restrict__ s, const float* __restrict__ W,
                                                                const float* __restrict__ b, const float* __restrict__ m, const float* __restrict__ a,
                                                                int K, float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dm)
{
    __shared__ float red[4];
    const int c = blockIdx.x;
    const float mc = m[c], sc = s[c];
    float dot = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        float g = G[(int64_t)c * K + k];
        if (a) g = fmaf(sc, a[k], g);
        dW[(int64_t)c * K + k] += mc * g;
        dot = fmaf(W[(int64_t)c * K + k], g, dot);
        G[(int64_t)c * K + k] = 0.f;
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (db) db[c] += mc * sc;
        dm[c] += (red[0] + red[1]) + (red[2] + red[3]) + (b ? b[c] * sc : 0.f);
        s[c] = 0.f;
    }
}

extern "C" int ryolo_head_wgrad_finish(float* G, float* s, const float* W, const float* b, const float* m, const float* a, int Cout, int K, float* dW,
                                       float* db, float* dm, float* da, hipStream_t stream)
{
    if (!G || !s || !W || !m || !dW || !dm || Cout <= 0 || K <= 0 || (a && !da)) return RY_ERR_ARG;
    if (a) hipLaunchKernelGGL(head_da_kernel, dim3((unsigned)ry_cdiv(K, 32)), dim3(256), 0, stream, s, W, m, Cout, K, da);
    hipLaunchKernelGGL(head_wgrad_finish_kernel, dim3(Cout), dim3(256), 0, stream, G, s, W, b, m, a, K, dW, db, dm);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

// bias of a detection head whose ImplicitA (model/neck.py:173-179: x + a in front of the 1x1 convolution) is folded into it: W (x + a) + b =
// W x + (b + W a) — out[c] = b[c] + sum_k W[c][k] a[k]; one workgroup per output channel.  Replaces the pass that wrote x + a.
__global__ __launch_bounds__(256) void head_bias_fold_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ a, int K,
                                                             float* __restrict__ out)
{
    __shared__ float red[4];
    const int c = blockIdx.x;
    float dot = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) dot = fmaf(W[(int64_t)c * K + k], a[k], dot);
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = (b ? b[c] : 0.f) + (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int ryolo_head_bias_fold(const float* W, const float* b, const float* a, int Cout, int K, float* out, hipStream_t stream)
{
    if (!W || !a || !out || Cout <= 0 || K <= 0) return RY_ERR_ARG;
    hipLaunchKernelGGL(head_bias_fold_kernel, dim3(Cout), dim3(256), 0, stream, W, b, a, K, out);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_chan_add(const bf16_t* x, int ldx, const float* a, int64_t M, int C, bf16_t* z, int ldz, hipStream_t stream)
{
    if (!x || !a || !z || (C & 7)) return RY_ERR_ARG;
    if (M == 0) return RY_OK;
    hipLaunchKernelGGL(chan_add_kernel, dim3(grid_for(M * (C >> 3))), dim3(256), 0, stream, x, ldx, a, M, C, z, ldz);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

// out[c] += sum_m x[m][c] for c < Cout (C = padded width, multiple of 8); scratch needs ceil(M/1024)*C floats
extern "C" int ryolo_colsum_bf16(const bf16_t* x, int ldx, int64_t M, int C, int Cvalid, float* out, float* scratch, hipStream_t stream)
{
    if (!x || !out || !scratch || (C & 7) || (ldx & 7) || Cvalid > C) return RY_ERR_ARG;
    if (M == 0) return RY_OK;
    const int rpb = 256;                                       // rows per workgroup (1024 left 78 workgroups for 8 images x 100^2: a latency-bound launch)
    const int nblk = (int)ry_cdiv(M, rpb);
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(nblk), dim3(256), 0, stream, x, ldx, M, C, rpb, scratch);
    int rows = nblk;                                           // one thread per column walked every partial row serially: fold to 64 rows first
    const float* part = fold_rows(scratch, rows, C, scratch + (int64_t)nblk * C, stream);
    hipLaunchKernelGGL(colsum_rows_kernel, dim3((unsigned)ry_cdiv(C, 256)), dim3(256), 0, stream, part, rows, C, Cvalid, out);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_pack_weights(const PackEntry* table_dev, int n, int64_t total, hipStream_t stream)
{
    if (!table_dev || n <= 0 || total <= 0) return RY_ERR_ARG;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)(total > 16384 ? 16384 : total)), dim3(256), 0, stream, table_dev, n, total);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_repconv_fold(const float* w3, const float* w1, const float* coa, const float* cob, int Cout, int Cin, bf16_t* wf,
                                  float* co_out, hipStream_t stream)
{
    if (!w3 || !w1 || !coa || !cob || !wf || !co_out || Cout <= 0 || Cin <= 0) return RY_ERR_ARG;
    const int64_t total = (int64_t)Cout * 9 * Cin;
    hipLaunchKernelGGL(repconv_fold_kernel, dim3((unsigned)ry_cdiv(total, 256)), dim3(256), 0, stream, w3, w1, coa, cob, Cout, Cin, wf, co_out);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_unpack_wgrad(const float* scratch, int Cout, int Cin, int taps, int CinP, float* grad, hipStream_t stream)
{
    if (!scratch || !grad) return RY_ERR_ARG;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((unsigned)ry_cdiv((int64_t)Cout * Cin * taps, 256)), dim3(256), 0, stream, scratch, Cout, Cin,
                       taps, CinP, grad);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_sgd_nesterov(float* p, float* g, float* buf, int64_t n, float lr, float mu, float gscale, int zero_grad,
                                  hipStream_t stream)
{
    if (!p || !g || !buf || n < 0 || (n & 3)) return RY_ERR_ARG;
    if (n == 0) return RY_OK;
    hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, p, g, buf, n / 4, lr, mu, gscale, zero_grad);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_adam(float* p, float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps, int64_t step,
                          float gscale, int zero_grad, hipStream_t stream)
{
    if (!p || !g || !m || !v || n < 0 || (n & 3) || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return RY_ERR_ARG;
    if (n == 0) return RY_OK;
    // every scalar is formed in double and rounded to fp32 once, as torch does with its Python floats (1 - 0.999 in fp32 is 4.7e-5 off 0.001)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, p, g, m, v, n / 4, (float)(lr / bc1), (float)sqrt(bc2),
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, gscale, zero_grad);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_struct_sizes(int* sizes /*[11]*/)
{
    if (!sizes) return RY_ERR_ARG;
    sizes[0] = (int)sizeof(BnActParams); sizes[1] = (int)sizeof(PoolParams); sizes[2] = (int)sizeof(UpParams); sizes[3] = (int)sizeof(PackEntry);
    sizes[4] = (int)sizeof(ConvGemmParams); sizes[5] = (int)sizeof(WgradParams); sizes[6] = (int)sizeof(LossParams); sizes[7] = (int)sizeof(TapClass);
    sizes[8] = (int)sizeof(StemParams); sizes[9] = (int)sizeof(StemWgradParams); sizes[10] = (int)sizeof(StemBwdParams);
    return RY_OK;
}
