// post_process device stages for gfx950 — replaces the per-image Python loop of
// lib/general.py:153-181 (B sequential iterations, each with boolean-mask host syncs) by three batched launches
// around the rotated-NMS kernels of nms.hip; the only host read-back is the final per-image detection count.
//
//   pp_score_kernel   cls *= obj IN PLACE (lib/general.py:155 mutates its input — kept), max over classes with the
//                     first-max rule, conf > conf_thres filter (strict, :161) encoded as key = -inf, per-image count.
//   (top-K of the keys by (score desc, index asc): csrc/topk.hip)
//   pp_gather_kernel  top-K rows in sorted order -> dets[B,K,7] and the NMS boxes (class offset 4096 px, rad->deg).
//   pp_emit_kernel    dets[keep] -> compact output rows.
// HBM-bound: algorithmic bytes = 4*B*M*(nc+6) read + 4*B*M*nc written back (the in-place product) + O(K).
#include "common.h"

// r05: a workgroup stages R consecutive rows (one contiguous block of R * (nc + 6) floats) in LDS with coalesced loads, one thread per row works
// on the LDS copy, and the block goes back with coalesced stores.  (One lane per row on global memory — nc + 1 strided scalar loads and nc strided
// stores per lane — ran at 0.55 TB/s: 2.4 ms for the 1.33 GB of a batch of 64.)  Same arithmetic per row.
__global__ __launch_bounds__(256) void pp_score_kernel(float* __restrict__ pred /*[B,M,nc+6] mutated*/, int B, int64_t M, int nc, float conf_thres,
                                                       float* __restrict__ key /*[B,M]*/, float* __restrict__ cls_out /*[B,M]*/,
                                                       int32_t* __restrict__ count /*[B], zeroed by caller*/, int R, int LD)
{
    extern __shared__ float ps_rows[];                           // [R][LD], LD odd (conflict-free row walks)
    const int A = nc + 6;
    const int64_t i0 = (int64_t)blockIdx.x * R;
    const int b = blockIdx.y;
    const int nrows = (int)(M - i0 < R ? M - i0 : R);
    float* const base = pred + ((int64_t)b * M + i0) * A;
    const int total = nrows * A;
    const float rA = 1.0f / (float)A;
    for (int e = threadIdx.x; e < total; e += 256) {
        int r = (int)((float)e * rA);
        if (r * A > e) r--;
        if ((r + 1) * A <= e) r++;
        ps_rows[r * LD + (e - r * A)] = base[e];
    }
    __syncthreads();
    bool pass = false;
    if ((int)threadIdx.x < nrows) {
        float* p = ps_rows + threadIdx.x * LD;
        const float obj = p[5];
        float best = 0.f;
        int bi = 0;
        for (int k = 0; k < nc; k++) {
            const float v = p[6 + k] * obj;
            p[6 + k] = v;
            if (k == 0 || v > best) { best = v; bi = k; }
        }
        pass = nc > 0 && best > conf_thres;
        key[(int64_t)b * M + i0 + threadIdx.x] = pass ? best : -INFINITY;
        cls_out[(int64_t)b * M + i0 + threadIdx.x] = (float)bi;
    }
    __shared__ int wave_pass[4];
    const unsigned long long m = __ballot(pass);
    if ((threadIdx.x & 63) == 0) wave_pass[threadIdx.x >> 6] = (int)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {                                       // one atomic per workgroup (per wave: 3 700 serialized atomics per image counter)
        const int n = wave_pass[0] + wave_pass[1] + wave_pass[2] + wave_pass[3];
        if (n) atomicAdd(&count[b], n);
    }
    for (int e = threadIdx.x; e < total; e += 256) {
        int r = (int)((float)e * rA);
        if (r * A > e) r--;
        if ((r + 1) * A <= e) r++;
        const int k = e - r * A;
        if (k >= 6) base[e] = ps_rows[r * LD + k];               // (the box / objectness columns are not changed)
    }
}

__global__ void pp_gather_kernel(const float* __restrict__ pred, const float* __restrict__ sorted_key /*[B,M] desc*/,
                                 const int64_t* __restrict__ order /*[B,M]*/, const float* __restrict__ cls, int B, int64_t M, int nc,
                                 int64_t K, int64_t stride /*row stride of sorted_key / order*/, float max_wh, float* __restrict__ dets /*[B,K,7]*/, float* __restrict__ rboxes /*[B,K,5]*/,
                                 int32_t* __restrict__ count /*[B] in: #pass, out: min(#pass,K)*/)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (k == 0) { const int c = count[b]; if (c > K) count[b] = (int32_t)K; }   // readers use min(count,K) themselves
    if (k >= K) return;
    const float s = sorted_key[(int64_t)b * stride + k];
    float* d = dets + ((int64_t)b * K + k) * 7;
    float* r = rboxes + ((int64_t)b * K + k) * 5;
    if (!(s > -INFINITY)) {
        for (int j = 0; j < 7; j++) d[j] = 0.f;
        for (int j = 0; j < 5; j++) r[j] = 0.f;
        return;
    }
    const int64_t src = order[(int64_t)b * stride + k];
    const float* p = pred + ((int64_t)b * M + src) * (nc + 6);
    const float c = cls[(int64_t)b * M + src];
    const float x = p[0], y = p[1], w = p[2], h = p[3], th = p[4];
    d[0] = x; d[1] = y; d[2] = w; d[3] = h; d[4] = th; d[5] = s; d[6] = c;
    const float off = c * max_wh;                                   // lib/general.py:171-173
    r[0] = x + off; r[1] = y + off; r[2] = w; r[3] = h;
    r[4] = th / 3.14159265358979323846f * 180.f;                    // lib/general.py:174 (true division, CPU semantics)
}

__global__ void pp_emit_kernel(const float* __restrict__ dets, const int64_t* __restrict__ keep, const int32_t* __restrict__ num_keep,
                               int64_t K, int64_t keep_stride, float* __restrict__ out /*[B,keep_stride,7]*/)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= keep_stride) return;
    float* o = out + ((int64_t)b * keep_stride + j) * 7;
    if (j < num_keep[b]) {
        const float* d = dets + ((int64_t)b * K + keep[(int64_t)b * keep_stride + j]) * 7;
        for (int t = 0; t < 7; t++) o[t] = d[t];
    } else {
        for (int t = 0; t < 7; t++) o[t] = 0.f;
    }
}

extern "C" int ryolo_pp_score(float* pred, int batch, int64_t M, int nc, float conf_thres, float* key, float* cls, int32_t* count,
                              hipStream_t stream)
{
    if (batch < 0 || M < 0 || nc < 0) return RY_ERR_ARG;
    if (batch == 0) return RY_OK;
    if (!count) return RY_ERR_ARG;
    if (hipMemsetAsync(count, 0, sizeof(int32_t) * batch, stream) != hipSuccess) return RY_ERR_LAUNCH;
    if (M == 0) return RY_OK;
    if (!pred || !key || !cls) return RY_ERR_ARG;
    const int LD = (nc + 6) | 1;
    int R = 12288 / LD;                                          // <= 48 KiB of LDS
    R = R > 256 ? 256 : (R < 1 ? 1 : R);
    if ((int64_t)R * (nc + 6) >= (1 << 24)) return RY_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pp_score_kernel, dim3((unsigned)ry_cdiv(M, R), batch), dim3(256), (size_t)R * LD * sizeof(float), stream, pred, batch, M, nc,
                       conf_thres, key, cls, count, R, LD);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_pp_gather(const float* pred, const float* sorted_key, const int64_t* order, const float* cls, int batch,
                               int64_t M, int nc, int64_t K, int64_t stride, float max_wh, float* dets, float* rboxes, int32_t* count,
                               hipStream_t stream)
{
    if (batch < 0 || M < 0 || K < 0 || K > M || stride < K) return RY_ERR_ARG;
    if (batch == 0 || K == 0) return RY_OK;
    if (!pred || !sorted_key || !order || !cls || !dets || !rboxes || !count) return RY_ERR_ARG;
    hipLaunchKernelGGL(pp_gather_kernel, dim3((unsigned)ry_cdiv(K, 256), batch), dim3(256), 0, stream, pred, sorted_key, order, cls,
                       batch, M, nc, K, stride, max_wh, dets, rboxes, count);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_pp_emit(const float* dets, const int64_t* keep, const int32_t* num_keep, int batch, int64_t K,
                             int64_t keep_stride, float* out, hipStream_t stream)
{
    if (batch < 0 || K < 0 || keep_stride < 0) return RY_ERR_ARG;
    if (batch == 0 || keep_stride == 0) return RY_OK;
    if (!dets || !keep || !num_keep || !out) return RY_ERR_ARG;
    hipLaunchKernelGGL(pp_emit_kernel, dim3((unsigned)ry_cdiv(keep_stride, 256), batch), dim3(256), 0, stream, dets, keep, num_keep, K,
                       keep_stride, out);
    RY_CHECK_LAUNCH();
    return RY_OK;
}
