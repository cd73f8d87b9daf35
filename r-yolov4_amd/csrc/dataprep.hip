// dataprep.hip — the batch-finalisation end of the reference's data pipeline and the detect path's box geometry, on the device
// (SURVEY.md §8(f) N2 slice + N4).  Everything here is HBM-bound byte/float movement; there is nothing to contract on MFMA.
//
//   ryolo_to_tensor       uint8 HWC BGR batch -> fp32 CHW RGB / 255 with per-image fliplr / flipud
//                         (datasets/base_dataset.py:131-136,155-157 + lib/augmentations.py:33-42 + collate stack :166)
//   ryolo_encode_labels   polygon targets [nt,10] -> filtering (:340-352), normalize (:354-361), flips (augmentations.py:35,41),
//                         xyxyxyxy2xywha (lib/general.py:70-104), CSL gaussian labels (base_dataset.py:13-31,143-149), collate
//                         sample index (:161-164); order-preserving compaction, count on the device
//   ryolo_dets_to_polys   rescale_boxes (lib/plot.py:9-31) + xywha2xyxyxyxy (lib/general.py:41-67) for a whole batch of detections
//
// fp32 arithmetic follows the reference's torch-CPU op sequence (python scalars enter float32 ops as float32); compiled with
// -ffp-contract=off so that no a*b+c is fused.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

// ------------------------------------------------------------------------------------------------ to_tensor
// One thread = 4 consecutive output pixels of one row: 12 source bytes (3 dwords when W % 4 == 0) -> one float4 per colour plane.
__global__ __launch_bounds__(256) void to_tensor_kernel(const uint8_t* __restrict__ src, int B, int H, int W, const uint8_t* __restrict__ flags,
                                                        float* __restrict__ dst)
{
    const int wq = (W + 3) >> 2;
    const int64_t total = (int64_t)B * H * wq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % wq);
        const int64_t r = i / wq;
        const int y = (int)(r % H), b = (int)(r / H);
        const int f = flags ? flags[b] : 0;
        const int ys = (f & 2) ? H - 1 - y : y;
        const int x0 = q * 4;
        const int n = min(4, W - x0);
        const uint8_t* row = src + ((int64_t)b * H + ys) * W * 3;
        const int xs0 = (f & 1) ? W - x0 - n : x0;                      // first source pixel of the (possibly mirrored) group
        const uint8_t* s = row + (int64_t)xs0 * 3;
        float* o0 = dst + (((int64_t)b * 3) * H + y) * W + x0;
        const int64_t plane = (int64_t)H * W;
        if (n == 4 && ((reinterpret_cast<uintptr_t>(s) & 3) == 0) && ((reinterpret_cast<uintptr_t>(o0) & 15) == 0) && (plane & 3) == 0) {
            const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
            const uint32_t w0 = s4[0], w1 = s4[1], w2 = s4[2];
            uint8_t px[12];
            px[0] = w0; px[1] = w0 >> 8; px[2] = w0 >> 16; px[3] = w0 >> 24;
            px[4] = w1; px[5] = w1 >> 8; px[6] = w1 >> 16; px[7] = w1 >> 24;
            px[8] = w2; px[9] = w2 >> 8; px[10] = w2 >> 16; px[11] = w2 >> 24;
#pragma unroll
            for (int c = 0; c < 3; c++) {                                 // BGR -> RGB, .float() / 255
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = (float)((f & 1) ? px[(3 - k) * 3 + (2 - c)] : px[k * 3 + (2 - c)]) / 255.0f;
                *reinterpret_cast<float4*>(o0 + c * plane) = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {
            for (int k = 0; k < n; k++) {
                const int ks = (f & 1) ? n - 1 - k : k;
                for (int c = 0; c < 3; c++) o0[c * plane + k] = (float)s[ks * 3 + (2 - c)] / 255.0f;
            }
        }
    }
}

extern "C" int ryolo_to_tensor(const uint8_t* src, int B, int H, int W, const uint8_t* flags, float* dst, hipStream_t stream)
{
    if (B < 0 || H < 0 || W < 0) return RY_ERR_ARG;
    const int64_t total = (int64_t)B * H * ((W + 3) >> 2);
    if (total == 0) return RY_OK;
    if (!src || !dst) return RY_ERR_ARG;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(to_tensor_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream, src, B, H, W, flags, dst);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

// ------------------------------------------------------------------------------------------------ labels
__device__ __forceinline__ float norm2(float dx, float dy) { return sqrtf(dx * dx + dy * dy); }

// lib/general.py:70-104 on one polygon (already normalised / flipped): (x, y, w, h, theta), h >= w, theta in [-pi/2, pi/2)
__device__ __forceinline__ void poly_to_xywha(const float* q, float* o)
{
    const float x1 = q[0], y1 = q[1], x2 = q[2], y2 = q[3], x3 = q[4], y3 = q[5], x4 = q[6], y4 = q[7];
    const float x = (((x1 + x2) + x3) + x4) / 4.0f;
    const float y = (((y1 + y2) + y3) + y4) / 4.0f;
    float w = (norm2(x2 - x3, y2 - y3) + norm2(x1 - x4, y1 - y4)) / 2.0f;
    float h = (norm2(x1 - x2, y1 - y2) + norm2(x4 - x3, y4 - y3)) / 2.0f;
    float th = -(atan2f(y1 - y2, x1 - x2) + atan2f(y4 - y3, x4 - x3)) / 2.0f;
    const float HALF_PI = 1.57079632679489661923f, PI = 3.14159265358979323846f;       // np.pi / 2, np.pi entering float32 ops
    if (w >= h) {                                                                       // lib/general.py:92-99
        const float t = w; w = h; h = t;
        th = th > 0.0f ? th - HALF_PI : th + HALF_PI;
    }
    if (th >= HALF_PI) th = th - PI;                                                     // norm_angle (lib/general.py:14-15)
    if (th < -HALF_PI) th = th + PI;
    o[0] = x; o[1] = y; o[2] = w; o[3] = h; o[4] = th;
}

// One 1024-thread workgroup walks the targets in order (ballot/popcount compaction keeps the reference's row order).
__global__ __launch_bounds__(1024) void encode_labels_kernel(const float* __restrict__ tg, int64_t nt, int H, int W,
                                                             const uint8_t* __restrict__ flags, const int* __restrict__ sample_of_img,
                                                             int ncols, float* __restrict__ out, int* __restrict__ bin0, int* __restrict__ count)
{
    __shared__ int wave_cnt[16];
    __shared__ int running;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = 0; base < nt; base += 1024) {
        const int64_t i = base + threadIdx.x;
        bool ok = false;
        float q[8];
        int img = 0;
        float cls = 0.f;
        if (i < nt) {
            const float* t = tg + i * 10;
            img = (int)t[0];
            cls = t[1];
#pragma unroll
            for (int k = 0; k < 8; k++) q[k] = t[2 + k];
            const float mx = (((q[0] + q[2]) + q[4]) + q[6]) / 4.0f;                     // filtering (base_dataset.py:345-352), border (0, W, 0, H)
            const float my = (((q[1] + q[3]) + q[5]) + q[7]) / 4.0f;
            ok = (mx > 0.0f) && (mx < (float)W) && (my > 0.0f) && (my < (float)H);
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = running;
        for (int w = 0; w < wave; w++) before += wave_cnt[w];
        if (ok) {
            const int e = before + __popcll(m & ((1ull << lane) - 1ull));
            const int f = flags ? flags[img] : 0;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                q[k] = q[k] / (float)W;                                                   // normalize (:357-359)
                q[k + 1] = q[k + 1] / (float)H;
                if (f & 1) q[k] = 1.0f - q[k];                                            // horizontal_flip (augmentations.py:41)
                if (f & 2) q[k + 1] = 1.0f - q[k + 1];                                    // vertical_flip (:35)
            }
            float o[5];
            poly_to_xywha(q, o);
            float* r = out + (int64_t)e * ncols;
            r[0] = (float)(sample_of_img ? sample_of_img[img] : img);                     // collate_fn: boxes[:, 0] = i
            r[1] = cls;
            r[2] = o[0]; r[3] = o[1]; r[4] = o[2]; r[5] = o[3]; r[6] = o[4];
            if (bin0) {
                // angle = theta * 180 / np.pi + 90 (float32 ops); index = int(180 / 2 - angle): truncation toward zero (base_dataset.py:30,145)
                const float ang = (o[4] * 180.0f) / 3.14159265358979323846f + 90.0f;
                bin0[e] = (int)(90.0f - ang);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { int tot = 0; for (int w = 0; w < 16; w++) tot += wave_cnt[w]; running += tot; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = running;
}

// CSL rows: out[e][7 + j] = y_sig[(j + index) mod 180], y_sig[k] = exp(-(k - 90)^2 / (2 * 6^2)) (float64 -> float32 like the numpy original)
__global__ __launch_bounds__(256) void csl_rows_kernel(float* __restrict__ out, const int* __restrict__ bin0, const int* __restrict__ count, int ncols)
{
    __shared__ float ysig[180];
    if (threadIdx.x < 180) {
        const double x = (double)((int)threadIdx.x - 90);
        ysig[threadIdx.x] = (float)exp(-(x * x) / 72.0);
    }
    __syncthreads();
    const int64_t total = (int64_t)(*count) * 180;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int e = (int)(i / 180), j = (int)(i - (int64_t)e * 180);
        int k = (j + bin0[e]) % 180;
        if (k < 0) k += 180;
        out[(int64_t)e * ncols + 7 + j] = ysig[k];
    }
}

extern "C" int ryolo_encode_labels(const float* targets, int64_t nt, int H, int W, const uint8_t* flags, const int* sample_of_img, int csl,
                                   float* out, int* count, int* workspace, hipStream_t stream)
{
    if (nt < 0 || H <= 0 || W <= 0 || !count) return RY_ERR_ARG;
    if (nt > 0 && (!targets || !out)) return RY_ERR_ARG;
    if (csl && nt > 0 && !workspace) return RY_ERR_ARG;
    const int ncols = csl ? 187 : 7;
    hipLaunchKernelGGL(encode_labels_kernel, dim3(1), dim3(1024), 0, stream, targets, nt, H, W, flags, sample_of_img, ncols, out,
                       csl ? workspace : (int*)nullptr, count);
    if (csl && nt > 0) {
        const int64_t blocks = (nt * 180 + 255) / 256;
        hipLaunchKernelGGL(csl_rows_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, out, workspace, count, ncols);
    }
    RY_CHECK_LAUNCH();
    return RY_OK;
}

__global__ __launch_bounds__(256) void polys_to_xywha_kernel(const float* __restrict__ polys, int64_t n, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float q[8], o[5];
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = polys[i * 8 + k];
    poly_to_xywha(q, o);
#pragma unroll
    for (int k = 0; k < 5; k++) out[i * 5 + k] = o[k];
}

extern "C" int ryolo_polys_to_xywha(const float* polys, int64_t n, float* out, hipStream_t stream)
{
    if (n < 0) return RY_ERR_ARG;
    if (n == 0) return RY_OK;
    if (!polys || !out) return RY_ERR_ARG;
    hipLaunchKernelGGL(polys_to_xywha_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, polys, n, out);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

// ------------------------------------------------------------------------------------------------ detect path
// dets [n][7] = (x, y, w, h, theta, conf, cls) in network pixels, img_of_det [n], shapes [B][2] = original (h, w).
// rescale_boxes (lib/plot.py:9-31) in place on columns 0..3, then xywha2xyxyxyxy (lib/general.py:41-67) -> polys [n][4][2].
__global__ __launch_bounds__(256) void dets_to_polys_kernel(float* __restrict__ dets, const int* __restrict__ img_of_det, const int* __restrict__ shapes,
                                                            int current_dim, int rescale, int64_t n, float* __restrict__ polys)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float* d = dets + i * 7;
    float x = d[0], y = d[1], w = d[2], h = d[3];
    const float th = d[4];
    if (rescale) {
        const int b = img_of_det ? img_of_det[i] : 0;
        const int oh = shapes[2 * b], ow = shapes[2 * b + 1];
        const double ratio = (double)current_dim / (double)(oh > ow ? oh : ow);       // python floats (double) ...
        const double pad_x = (double)(oh - ow > 0 ? oh - ow : 0) * ratio;
        const double pad_y = (double)(ow - oh > 0 ? ow - oh : 0) * ratio;
        const float unpad_h = (float)((double)current_dim - pad_y);                   // ... entering float32 tensor ops as float32
        const float unpad_w = (float)((double)current_dim - pad_x);
        const float hx = (float)floor(pad_x / 2.0), hy = (float)floor(pad_y / 2.0);   // pad // 2
        float x1 = x - w / 2.0f, y1 = y - h / 2.0f, x2 = x + w / 2.0f, y2 = y + h / 2.0f;   // xywh2xyxy (lib/general.py:33-37)
        x1 = ((x1 - hx) / unpad_w) * (float)ow;
        y1 = ((y1 - hy) / unpad_h) * (float)oh;
        x2 = ((x2 - hx) / unpad_w) * (float)ow;
        y2 = ((y2 - hy) / unpad_h) * (float)oh;
        x = (x1 + x2) / 2.0f; y = (y1 + y2) / 2.0f; w = x2 - x1; h = y2 - y1;
        d[0] = x; d[1] = y; d[2] = w; d[3] = h;
    }
    // cv.getRotationMatrix2D((x, y), theta * 180 / pi, 1) in double, stored as float32 (lib/general.py:55-57)
    const float deg = (th * 180.0f) / 3.14159265358979323846f;
    const double a = (double)deg * 3.14159265358979323846 / 180.0;
    const double al = cos(a), be = sin(a);
    const double cx = (double)x, cy = (double)y;
    const float r00 = (float)al, r01 = (float)be, r02 = (float)((1.0 - al) * cx - be * cy);
    const float r10 = (float)(-be), r11 = (float)al, r12 = (float)(be * cx + (1.0 - al) * cy);
    const float px[4] = {x - h / 2.0f, x + h / 2.0f, x + h / 2.0f, x - h / 2.0f};      // NOTE: h runs along x before the rotation (:59-62)
    const float py[4] = {y - w / 2.0f, y - w / 2.0f, y + w / 2.0f, y + w / 2.0f};
    float* o = polys + i * 8;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        o[2 * k] = (px[k] * r00 + py[k] * r01) + r02;                                   // bmm(p, Rs^T), p = (x, y, 1)
        o[2 * k + 1] = (px[k] * r10 + py[k] * r11) + r12;
    }
}

extern "C" int ryolo_dets_to_polys(float* dets, const int* img_of_det, const int* shapes, int current_dim, int rescale, int64_t n, float* polys,
                                   hipStream_t stream)
{
    if (n < 0 || (rescale && (!shapes || current_dim <= 0))) return RY_ERR_ARG;
    if (n == 0) return RY_OK;
    if (!dets || !polys) return RY_ERR_ARG;
    hipLaunchKernelGGL(dets_to_polys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dets, img_of_det, shapes, current_dim, rescale, n,
                       polys);
    RY_CHECK_LAUNCH();
    return RY_OK;
}
