// YoloLayer reshape + grid/anchor decode for gfx950.
// Replaces model/yololayer.py:15-56 (YoloCSLLayer.forward) and :66-105 (YoloKFIoULayer.forward) of the reference.
//
// HBM-bound streaming kernels.  The reference runs ~15 small launches per scale (view, permute+contiguous, sigmoid,
// arange/repeat, cat, ...); here one launch per scale reads the raw head logits once and writes the decoded rows
// straight into their slice of the concatenated [B, sum(na*gs*gs), nc+6] buffer (no per-scale cat).
//   kfiou: one lane per ELEMENT of the contiguous [rows][attrs] array (r05; one lane per 88-byte row before: 22 strided scalar accesses each way).
//   csl  : one wavefront per cell (nc+185 floats, coalesced 4-byte lanes); the 180-bin argmax is a wave64
//          (value, index) butterfly that keeps the FIRST maximal bin (torch.max tie rule, model/yololayer.py:48).
#include "common.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// full-precision variant for values that must match torch.sigmoid to ~1 ulp
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// [B, na*attrs, gs, gs] (NCHW conv output) -> [B, na, gs, gs, attrs]   (model/yololayer.py:25 / :76)
__global__ void head_permute_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int na, int attrs, int gs)
{
    // tile transpose through LDS: in is [.., attrs, cells], out is [.., cells, attrs]
    __shared__ float tile[32][33];
    const int cells = gs * gs;
    const int ba = blockIdx.z;                       // b*na + a
    const int c0 = blockIdx.x * 32, a0 = blockIdx.y * 32;
    const float* src = in + (int64_t)ba * attrs * cells;
    float* dst = out + (int64_t)ba * cells * attrs;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int at = a0 + j, ce = c0 + threadIdx.x;
        if (at < attrs && ce < cells) tile[j][threadIdx.x] = src[(int64_t)at * cells + ce];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int ce = c0 + j, at = a0 + threadIdx.x;
        if (at < attrs && ce < cells) dst[(int64_t)ce * attrs + at] = tile[threadIdx.x][j];
    }
}

struct AnchorSet { float v[18 * 3]; };   // up to 18 anchors x (w, h, angle) in grid units, passed by value

// kfiou: rows [x,y,w,h,a,obj,cls...] -> [x,y,w,h,theta,conf,cls...]
// Every output element depends on ONE input element (its own column) and on the row's (anchor, grid cell): one LANE PER ELEMENT of the contiguous
// [rows][attrs] array — coalesced 4-byte loads and stores (r05; the r01-r04 form gave a lane a whole 88-byte row: 22 strided scalar loads and 22
// strided stores per lane, 0.6 TB/s, 10 % of the batch-64 inference forward).  A workgroup owns DK_EPW consecutive elements; its first row is
// decomposed once (scalar 64-bit divisions), the rows inside by small exact divisions.
#define DK_EPW 2048
__device__ __forceinline__ int dk_div(int n, int d, float rd)
{
    int q = (int)((float)n * rd);
    if (q * d > n) q--;
    if ((q + 1) * d <= n) q++;
    return q;
}
__global__ __launch_bounds__(256) void decode_kfiou_kernel(const float* __restrict__ t /*[B,na,gs,gs,attrs]*/, float* __restrict__ out, int B, int na,
                                                           int gs, int nc, float stride, AnchorSet an, int64_t row_offset, int64_t rows_per_image)
{
    __shared__ float anl[18 * 3];
    const int attrs = nc + 6, cells = gs * gs;
    const int64_t per_img = (int64_t)na * cells;
    const int64_t total = (int64_t)B * per_img * attrs;
    if (threadIdx.x < na * 3) anl[threadIdx.x] = an.v[threadIdx.x];
    __syncthreads();
    const int64_t f0 = (int64_t)blockIdx.x * DK_EPW;             // (wave-uniform: scalar arithmetic)
    const int64_t row0 = f0 / attrs;
    const int k0 = (int)(f0 - row0 * attrs);
    const int b0 = (int)(row0 / per_img);
    const int64_t r0 = row0 - (int64_t)b0 * per_img;
    const int a0 = (int)(r0 / cells), cell0 = (int)(r0 - (int64_t)a0 * cells);
    const float rattrs = 1.0f / (float)attrs, rcells = 1.0f / (float)cells, rgs = 1.0f / (float)gs, rna = 1.0f / (float)na;
#pragma unroll
    for (int j = 0; j < DK_EPW / 256; j++) {
        const int off = j * 256 + threadIdx.x;
        const int64_t f = f0 + off;
        if (f >= total) break;
        const int local = k0 + off;                              // < DK_EPW + attrs
        const int drow = dk_div(local, attrs, rattrs), k = local - drow * attrs;
        const float x = t[f];
        // (image, anchor, cell) of row0 + drow
        int cell = cell0 + drow;
        const int da = dk_div(cell, cells, rcells);
        cell -= da * cells;
        int a = a0 + da;
        const int db = dk_div(a, na, rna);
        a -= db * na;
        const int b = b0 + db;
        const float s = sigmoid_acc(x);
        float v = s;                                            // columns >= 5: objectness and class scores
        if (k < 5) {
            if (k < 2) {
                const int gy = dk_div(cell, gs, rgs), gx = cell - gy * gs;
                v = (s * 2.f - 0.5f + (float)(k == 0 ? gx : gy)) * stride;
            } else if (k < 4) {
                const float w2 = s * 2.f;
                v = w2 * w2 * anl[a * 3 + (k - 2)] * stride;
            } else {
                v = (s - 0.5f) * 0.5236f + anl[a * 3 + 2];      // model/yololayer.py:96 (no norm_angle)
            }
        }
        out[((int64_t)b * rows_per_image + row_offset + (int64_t)a * cells + cell) * attrs + k] = v;
    }
}

// csl: rows [x,y,w,h,obj,cls x nc,theta x 180] -> [x,y,w,h,theta,conf,cls...]; one wave per cell
__global__ __launch_bounds__(256) void decode_csl_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int na, int gs,
                                                         int nc, float stride, AnchorSet an, int64_t row_offset,
                                                         int64_t rows_per_image)
{
    const int attrs = nc + 185, oattrs = nc + 6;
    const int64_t per_img = (int64_t)na * gs * gs;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= (int64_t)B * per_img) return;
    const int b = (int)(i / per_img);
    const int64_t r = i - (int64_t)b * per_img;
    const int a = (int)(r / (gs * gs));
    const int cell = (int)(r - (int64_t)a * gs * gs);
    const int gy = cell / gs, gx = cell - gy * gs;
    const float* p = t + i * attrs;
    float* o = out + ((int64_t)b * rows_per_image + row_offset + r) * oattrs;

    // argmax over the 180 sigmoid values, first maximal index
    float best = -1.f;
    int bidx = 0x7fffffff;
    const float* bins = p + 5 + nc;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int j = lane + 64 * k;
        if (j < 180) {
            const float v = sigmoid_acc(bins[j]);
            if (v > best) { best = v; bidx = j; }     // ascending j per lane: strict > keeps the first
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bidx, off, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane == 0) {
        const float sx = sigmoid_acc(p[0]), sy = sigmoid_acc(p[1]), sw = sigmoid_acc(p[2]), sh = sigmoid_acc(p[3]);
        o[0] = (sx * 2.f - 0.5f + (float)gx) * stride;
        o[1] = (sy * 2.f - 0.5f + (float)gy) * stride;
        const float w2 = sw * 2.f, h2 = sh * 2.f;
        o[2] = w2 * w2 * an.v[a * 3 + 0] * stride;
        o[3] = h2 * h2 * an.v[a * 3 + 1] * stride;
        o[4] = ((float)(bidx - 90)) / 180.f * 3.14159265358979323846f;      // model/yololayer.py:49
        o[5] = sigmoid_acc(p[4]);
    }
    if (lane < nc) o[6 + lane] = sigmoid_acc(p[5 + lane]);
    for (int k = 64 + lane; k < nc; k += 64) o[6 + k] = sigmoid_acc(p[5 + k]);
}

extern "C" int ryolo_head_permute(const float* in, float* out, int batch, int na, int attrs, int gs, hipStream_t stream)
{
    if (batch < 0 || na <= 0 || attrs <= 0 || gs <= 0) return RY_ERR_ARG;
    if (batch == 0) return RY_OK;
    if (!in || !out) return RY_ERR_ARG;
    dim3 grid((unsigned)ry_cdiv((int64_t)gs * gs, 32), (unsigned)ry_cdiv(attrs, 32), (unsigned)(batch * na));
    hipLaunchKernelGGL(head_permute_kernel, grid, dim3(32, 8), 0, stream, in, out, batch, na, attrs, gs);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_decode(int mode /*0 csl, 1 kfiou*/, const float* head /*[B,na,gs,gs,attrs]*/, float* infer_out, int batch,
                            int na, int gs, int nc, float stride, const float* anchors_host /*[na,3] w,h,angle*/,
                            int64_t row_offset, int64_t rows_per_image, hipStream_t stream)
{
    if (batch < 0 || na <= 0 || na > 18 || gs <= 0 || nc < 0 || (mode != 0 && mode != 1)) return RY_ERR_ARG;
    if (batch == 0) return RY_OK;
    if (!head || !infer_out || !anchors_host) return RY_ERR_ARG;
    AnchorSet an;
    for (int i = 0; i < na * 3; i++) an.v[i] = anchors_host[i];
    const int64_t cells = (int64_t)batch * na * gs * gs;
    if (mode == 1)
        hipLaunchKernelGGL(decode_kfiou_kernel, dim3((unsigned)ry_cdiv(cells * (nc + 6), DK_EPW)), dim3(256), 0, stream, head, infer_out, batch,
                           na, gs, nc, stride, an, row_offset, rows_per_image);
    else
        hipLaunchKernelGGL(decode_csl_kernel, dim3((unsigned)ry_cdiv(cells, 4)), dim3(256), 0, stream, head, infer_out, batch, na,
                           gs, nc, stride, an, row_offset, rows_per_image);
    RY_CHECK_LAUNCH();
    return RY_OK;
}
