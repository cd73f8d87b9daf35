// Score ordering on the device for gfx950: the `argsort(descending=True)` + `[:max_nms]` of
// lib/general.py:166-168 and the score sort inside detectron2's nms_rotated (call site lib/general.py:177), with the tie rule
// SURVEY §7 fixes (equal scores -> ascending candidate index).  Replaces torch.sort (rocPRIM) on the product path: everything
// stays on the stream, counts stay on the device, nothing is read back -> capturable in a hipGraph with worst-case buffers.
//
//   ryolo_topk_desc     per image (one 1024-thread workgroup each): 4-pass 8-bit RADIX SELECT of the K-th largest key over up to
//                       ~400 k candidates (387 072 at 1024^2), histogram in LDS with wave-aggregated atomics (detection scores share
//                       their exponent byte: a plain LDS atomic would serialise 64 lanes on one bin), then an index-ordered
//                       compaction (ballot / popcount: ties at the threshold are taken in ascending index order) into a
//                       power-of-two buffer, an LDS bitonic sort of 64-bit (key, ~index) composites, and the emit pass.
//                       Non-candidates (key = -inf, lib/general.py:161's `conf > conf_thres` filter) are never selected.
//   ryolo_argsort_desc  full stable descending argsort of N scores (nms_rotated's own sort; N = 10 000 for BASELINE's metric, 50 000
//                       for config C5): chunks of 16 384 composites sorted in LDS (128 KiB of the CU's 160), larger N by global
//                       bitonic merge steps between the LDS stages.
// Composite = (order-preserving u32 image of the fp32 key) << 32 | ~index: one 64-bit descending sort gives (key desc, index asc).
// HBM-bound byte work: 4 B * M per radix pass (5 passes over the keys), 8 B * P for the sort.
#include "common.h"

typedef unsigned long long u64;
#define TK_THREADS 1024
#define TK_CHUNK 16384                      // composites per LDS sort (128 KiB)

__device__ __forceinline__ unsigned key_image(float f)
{
    f += 0.0f;                                                     // -0.0 -> +0.0 (they compare equal: the index decides)
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);             // larger float <-> larger unsigned
}
__device__ __forceinline__ float key_unimage(unsigned u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
#define TK_NEGINF_IMAGE 0x007fffffu                                // key_image(-inf)

// histogram increment with the lanes of a wave that hit the same bin combined into one LDS atomic
__device__ __forceinline__ void hist_add(unsigned* hist, unsigned bin, bool active)
{
    unsigned long long todo = __ballot(active);
    int guard = 0;
    while (todo) {
        if (++guard > 6) {                                         // many distinct bins in this wave: contention is low anyway
            if (active) atomicAdd(&hist[bin], 1u);
            return;
        }
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned lb = (unsigned)__shfl((int)bin, leader, 64);
        const unsigned long long same = __ballot(active && bin == lb);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lb], (unsigned)__popcll(same));
        if (active && bin == lb) active = false;
        todo &= ~same;
    }
}

// ---- select: composites of the K largest keys of row b -> comp[b][0 .. nsel) (unordered among > T, index order at == T), zero padded to P
// Each of the 16 waves walks ONE contiguous segment of the row (64 consecutive keys per load, 4 loads in flight), so the index-ordered
// compaction needs no workgroup barrier inside the walk: a counting sweep, one 16-entry scan, a writing sweep.
#define TK_WAVES (TK_THREADS / 64)
#define TK_UNROLL 4
__global__ __launch_bounds__(TK_THREADS) void topk_select_kernel(const float* __restrict__ key, int64_t M, int K, int P, u64* __restrict__ comp,
                                                                 int32_t* __restrict__ nsel)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_remaining, s_valid;
    __shared__ unsigned wave_gt[TK_WAVES], wave_eq[TK_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float* krow = key + (int64_t)b * M;
    u64* crow = comp + (int64_t)b * P;
    const int64_t per = (M + TK_WAVES - 1) / TK_WAVES, seg = (per + 64 * TK_UNROLL - 1) / (64 * TK_UNROLL) * (64 * TK_UNROLL);
    const int64_t s0 = (int64_t)wave * seg, s1 = s0 + seg < M ? s0 + seg : M;
    if (tid == 0) { s_prefix = 0; s_valid = 0; }
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        unsigned valid = 0;
        for (int64_t i0 = s0; i0 < s1; i0 += 64 * TK_UNROLL) {
            unsigned u[TK_UNROLL];
#pragma unroll
            for (int j = 0; j < TK_UNROLL; j++) {
                const int64_t i = i0 + j * 64 + lane;
                u[j] = i < s1 ? key_image(krow[i]) : 0u;           // 0 < TK_NEGINF_IMAGE: never a candidate
            }
#pragma unroll
            for (int j = 0; j < TK_UNROLL; j++) {
                const bool cand = u[j] > TK_NEGINF_IMAGE;          // -inf (and anything below) is never a candidate
                const bool act = cand && (pass == 0 || (u[j] >> (shift + 8)) == prefix);
                hist_add(hist, (u[j] >> shift) & 255u, act);
                if (pass == 0) valid += __popcll(__ballot(cand));  // identical in every lane of the wave
            }
        }
        if (pass == 0 && lane == 0) atomicAdd(&s_valid, valid);
        __syncthreads();
        if (tid == 0) {
            if (pass == 0) s_remaining = min((unsigned)K, s_valid);
            unsigned rem = s_remaining, c = 0;
            if (rem) {
                for (int d = 255; d >= 0; d--) {
                    if (c + hist[d] >= rem) { s_prefix = (prefix << 8) | (unsigned)d; s_remaining = rem - c; break; }
                    c += hist[d];
                }
            }
        }
        __syncthreads();
    }
    const unsigned T = s_prefix, take_eq = s_remaining;             // T: image of the K-th largest key; take_eq of the keys == T (lowest indices)
    const unsigned total = min((unsigned)K, s_valid);
    if (total) {
        unsigned ngt = 0, neq = 0;                                  // counting sweep (wave-uniform)
        for (int64_t i0 = s0; i0 < s1; i0 += 64 * TK_UNROLL) {
            unsigned u[TK_UNROLL];
#pragma unroll
            for (int j = 0; j < TK_UNROLL; j++) {
                const int64_t i = i0 + j * 64 + lane;
                u[j] = i < s1 ? key_image(krow[i]) : 0u;
            }
#pragma unroll
            for (int j = 0; j < TK_UNROLL; j++) {
                const bool cand = u[j] > TK_NEGINF_IMAGE;
                ngt += (unsigned)__popcll(__ballot(cand && u[j] > T));
                neq += (unsigned)__popcll(__ballot(cand && u[j] == T));
            }
        }
        if (lane == 0) { wave_gt[wave] = ngt; wave_eq[wave] = neq; }
        __syncthreads();
        unsigned eq_seen = 0, pos = 0;                              // keys == T before this wave's segment / output slots before it
        for (int w = 0; w < wave; w++) {
            const unsigned e = wave_eq[w];
            const unsigned te = eq_seen < take_eq ? min(e, take_eq - eq_seen) : 0u;
            pos += wave_gt[w] + te;
            eq_seen += e;
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int64_t i0 = s0; i0 < s1; i0 += 64 * TK_UNROLL) {      // writing sweep
            unsigned u[TK_UNROLL];
#pragma unroll
            for (int j = 0; j < TK_UNROLL; j++) {
                const int64_t i = i0 + j * 64 + lane;
                u[j] = i < s1 ? key_image(krow[i]) : 0u;
            }
#pragma unroll
            for (int j = 0; j < TK_UNROLL; j++) {
                const int64_t i = i0 + j * 64 + lane;
                const bool cand = u[j] > TK_NEGINF_IMAGE;
                const bool gt = cand && u[j] > T, eq = cand && u[j] == T;
                const unsigned long long beq = __ballot(eq);
                const bool take = gt || (eq && eq_seen + (unsigned)__popcll(beq & lt) < take_eq);
                const unsigned long long bt = __ballot(take);
                if (take) crow[pos + (unsigned)__popcll(bt & lt)] = ((u64)u[j] << 32) | (u64)(~(unsigned)i);
                pos += (unsigned)__popcll(bt);
                eq_seen += (unsigned)__popcll(beq);
            }
        }
    }
    for (int i = (int)total + tid; i < P; i += TK_THREADS) crow[i] = 0ull;
    if (nsel && tid == 0) nsel[b] = (int32_t)total;
}

// ---- composites of a full score vector (argsort), zero padded to P
__global__ void compose_kernel(const float* __restrict__ scores, int64_t N, int64_t P, u64* __restrict__ comp)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    comp[i] = i < N ? (((u64)key_image(scores[i]) << 32) | (u64)(~(unsigned)i)) : 0ull;
}

// ---- bitonic network, descending.  Local kernel: a chunk lives in LDS and runs every (k, j) stage with j < chunk for k in
// [k_first, k_last]; global kernel: one (k, j) stage with j >= chunk.  Row stride P (blockIdx.y = row).
extern __shared__ __attribute__((aligned(16))) unsigned char tk_lds[];

__global__ __launch_bounds__(TK_THREADS) void bitonic_local_kernel(u64* __restrict__ data, int64_t P, int chunk, int64_t k_first, int64_t k_last)
{
    u64* s = reinterpret_cast<u64*>(tk_lds);
    const int64_t off = (int64_t)blockIdx.x * chunk;
    u64* row = data + (int64_t)blockIdx.y * P + off;
    for (int i = threadIdx.x; i < chunk; i += TK_THREADS) s[i] = row[i];
    __syncthreads();
    for (int64_t k = k_first; k <= k_last; k <<= 1) {
        int j0 = (int)((k >> 1) < chunk ? (k >> 1) : (chunk >> 1));
        for (int j = j0; j >= 1; j >>= 1) {
            for (int t = threadIdx.x; t < (chunk >> 1); t += TK_THREADS) {
                const int i = ((t / j) * 2 * j) + (t % j);
                const u64 a = s[i], c = s[i + j];
                const bool desc = ((off + i) & k) == 0;            // this run of the network is descending
                if (desc ? (a < c) : (a > c)) { s[i] = c; s[i + j] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < chunk; i += TK_THREADS) row[i] = s[i];
}

__global__ void bitonic_global_kernel(u64* __restrict__ data, int64_t P, int64_t k, int64_t j)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (P >> 1)) return;
    u64* row = data + (int64_t)blockIdx.y * P;
    const int64_t i = ((t / j) * 2 * j) + (t % j);
    const u64 a = row[i], c = row[i + j];
    const bool desc = (i & k) == 0;
    if (desc ? (a < c) : (a > c)) { row[i] = c; row[i + j] = a; }
}

__global__ void topk_emit_kernel(const u64* __restrict__ comp, int64_t P, int64_t K, float* __restrict__ skey, int64_t* __restrict__ order)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int b = blockIdx.y;
    const u64 v = comp[(int64_t)b * P + k];
    if (skey) skey[(int64_t)b * K + k] = v ? key_unimage((unsigned)(v >> 32)) : -INFINITY;
    if (order) order[(int64_t)b * K + k] = v ? (int64_t)(~(unsigned)v) : (int64_t)-1;
}

static int64_t tk_pow2(int64_t n) { int64_t p = 64; while (p < n) p <<= 1; return p; }

static int tk_sort_rows(u64* comp, int rows, int64_t P, hipStream_t stream)
{
    static RyLdsAttr attr;
    if (ry_max_dynamic_lds(attr, reinterpret_cast<const void*>(bitonic_local_kernel), TK_CHUNK * 8)) return RY_ERR_LAUNCH;
    const int chunk = (int)(P < TK_CHUNK ? P : TK_CHUNK);
    const unsigned nchunks = (unsigned)(P / chunk);
    hipLaunchKernelGGL(bitonic_local_kernel, dim3(nchunks, rows), dim3(TK_THREADS), (size_t)chunk * 8, stream, comp, P, chunk, (int64_t)2, (int64_t)chunk);
    for (int64_t k = 2 * (int64_t)chunk; k <= P; k <<= 1) {
        for (int64_t j = k >> 1; j >= chunk; j >>= 1)
            hipLaunchKernelGGL(bitonic_global_kernel, dim3((unsigned)ry_cdiv(P >> 1, 256), rows), dim3(256), 0, stream, comp, P, k, j);
        hipLaunchKernelGGL(bitonic_local_kernel, dim3(nchunks, rows), dim3(TK_THREADS), (size_t)chunk * 8, stream, comp, P, chunk, k, k);
    }
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_sort_workspace_bytes(int rows, int64_t n_sorted, size_t* bytes)
{
    if (!bytes || rows < 0 || n_sorted < 0 || n_sorted > (1ll << 24)) return RY_ERR_ARG;
    *bytes = (size_t)(rows > 0 ? rows : 1) * (size_t)tk_pow2(n_sorted) * sizeof(u64);
    return RY_OK;
}

extern "C" int ryolo_topk_desc(const float* key, int batch, int64_t M, int K, float* skey, int64_t* order, int32_t* nsel, void* ws,
                               size_t ws_bytes, hipStream_t stream)
{
    if (batch < 0 || M < 0 || K < 0 || K > M || M >= (1ll << 32)) return RY_ERR_ARG;
    if (batch == 0 || K == 0) return RY_OK;
    if (K > TK_CHUNK) return RY_ERR_UNSUPPORTED;                     // (the reference caps at 5000, lib/general.py:148)
    if (!key || !ws || (!skey && !order)) return RY_ERR_ARG;
    const int64_t P = tk_pow2(K);
    if (ws_bytes < (size_t)batch * P * sizeof(u64)) return RY_ERR_WORKSPACE;
    u64* comp = reinterpret_cast<u64*>(ws);
    hipLaunchKernelGGL(topk_select_kernel, dim3(batch), dim3(TK_THREADS), 0, stream, key, M, K, (int)P, comp, nsel);
    const int rc = tk_sort_rows(comp, batch, P, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(topk_emit_kernel, dim3((unsigned)ry_cdiv(K, 256), batch), dim3(256), 0, stream, comp, P, (int64_t)K, skey, order);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_argsort_desc(const float* scores, int64_t N, int64_t* order, void* ws, size_t ws_bytes, hipStream_t stream)
{
    if (N < 0 || N > (1ll << 24)) return RY_ERR_ARG;
    if (N == 0) return RY_OK;
    if (!scores || !order || !ws) return RY_ERR_ARG;
    const int64_t P = tk_pow2(N);
    if (ws_bytes < (size_t)P * sizeof(u64)) return RY_ERR_WORKSPACE;
    u64* comp = reinterpret_cast<u64*>(ws);
    hipLaunchKernelGGL(compose_kernel, dim3((unsigned)ry_cdiv(P, 256)), dim3(256), 0, stream, scores, N, P, comp);
    const int rc = tk_sort_rows(comp, 1, P, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(topk_emit_kernel, dim3((unsigned)ry_cdiv(N, 256), 1), dim3(256), 0, stream, comp, P, N, (float*)nullptr, order);
    RY_CHECK_LAUNCH();
    return RY_OK;
}
