// Rotated NMS + pairwise / element-wise SkewIoU for gfx950.
// Replaces torch.ops.detectron2.nms_rotated / box_iou_rotated (third-party CUDA, call sites
// lib/general.py:177 and test.py:135 of the reference).  Compiled with -ffp-contract=off (bit-exact keep sets).
//
// Pipeline per image (all on device, batched over images with grid.z / grid.x, no host round trip):
//   1. nms_prep_kernel    per-box prologue -> BoxPrep (48 B): hoists the double-precision trig out of the N^2 loop.
//   2. nms_mask_kernel    one workgroup (4 waves) per 64x64 tile of the upper triangle:
//                           phase 1  4096 cheap circle-reject tests, survivors compacted into an LDS queue
//                                    (wave-aggregated LDS atomics), so the expensive pair function runs on
//                                    dense wavefronts instead of 1-2 live lanes per wave;
//                           phase 2  exact rotated IoU on the queue, bits OR-ed into an LDS 64x64 bit tile;
//                           phase 3  one 8-byte word per row to the [n][ceil(n/64)] u64 mask.
//   3. nms_reduce_kernel  the sequential greedy pass, on device (detectron2 copies the mask to the host):
//                           one 1024-thread workgroup per image; per 64-box chunk one wave resolves the diagonal
//                           word with ballot/readlane (loop length = number of rows that suppress anything, not 64),
//                           then all 16 waves OR the kept rows into an LDS-resident `removed` bitmap;
//                           stops as soon as max_keep boxes are kept (post_process keeps only max_det=1500).
#include "common.h"
#include "rotated_iou.h"

#define TILE 64

__global__ void nms_prep_kernel(const float* __restrict__ boxes, int64_t total, BoxPrep* __restrict__ prep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    BoxPrep p;
    box_prep(boxes + 5 * i, p);
    float4* o = reinterpret_cast<float4*>(prep + i);
    o[0] = make_float4(p.x, p.y, p.w, p.h);
    o[1] = make_float4(p.sh, p.cw, p.ch, p.sw);
    o[2] = make_float4(p.area, p.rad, 0.f, 0.f);
}

__device__ __forceinline__ void load_tile(const BoxPrep* __restrict__ src, int64_t base, int nvalid, BoxPrep* dst, int tid)
{
    // 64 boxes x 3 float4 = 192 float4, coalesced
    if (tid < 192) {
        const int box = tid / 3;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (box < nvalid) v = reinterpret_cast<const float4*>(src + base)[tid];
        reinterpret_cast<float4*>(dst)[tid] = v;
    }
}

__global__ __launch_bounds__(256) void nms_mask_kernel(const BoxPrep* __restrict__ prep, const int32_t* __restrict__ counts,
                                                       int64_t nmax, int nw, float thr, int gt_only, int prune,
                                                       unsigned long long* __restrict__ mask)
{
    const int cb = blockIdx.x, rb = blockIdx.y, img = blockIdx.z;
    if (rb > cb) return;
    int64_t n = nmax;
    if (counts) { n = counts[img]; if (n > nmax) n = nmax; }
    if ((int64_t)cb * TILE >= n) return;

    __shared__ BoxPrep rowb[TILE];
    __shared__ BoxPrep colb[TILE];
    __shared__ unsigned short queue[TILE * TILE];
    __shared__ unsigned mbits[TILE][2];
    __shared__ int qcount;

    const int tid = threadIdx.x;
    const int64_t img_base = (int64_t)img * nmax;
    const int nrow = (int)min((int64_t)TILE, n - (int64_t)rb * TILE);
    const int ncol = (int)min((int64_t)TILE, n - (int64_t)cb * TILE);
    load_tile(prep, img_base + (int64_t)rb * TILE, nrow, rowb, tid);
    if (tid == 0) qcount = 0;
    if (tid < TILE) { mbits[tid][0] = 0u; mbits[tid][1] = 0u; }
    __syncthreads();           // rowb complete before colb reuses the same loader threads' registers
    load_tile(prep, img_base + (int64_t)cb * TILE, ncol, colb, tid);
    __syncthreads();

    // phase 1: cheap reject + compaction
    {
        const int r = tid & 63, q = tid >> 6;
        const bool rvalid = r < nrow;
        const BoxPrep A = rowb[r];
#pragma unroll 4
        for (int cc = 0; cc < 16; cc++) {
            const int c = q * 16 + cc;
            bool live = rvalid && c < ncol && (rb < cb || c > r);
            if (live && prune) live = !boxes_far_apart(A, colb[c]);
            if (live) {
                const int slot = atomicAdd(&qcount, 1);
                queue[slot] = (unsigned short)((r << 6) | c);
            }
        }
    }
    __syncthreads();

    // phase 2: exact IoU on the compacted pairs
    const int nq = qcount;
    for (int k = tid; k < nq; k += 256) {
        const int pr = queue[k];
        const int r = pr >> 6, c = pr & 63;
        const float iou = rotated_iou_pair(rowb[r], colb[c]);
        const bool sup = gt_only ? (iou > thr) : (iou >= thr);
        if (sup) atomicOr(&mbits[r][c >> 5], 1u << (c & 31));
    }
    __syncthreads();

    if (tid < nrow) {
        const unsigned long long w = ((unsigned long long)mbits[tid][1] << 32) | mbits[tid][0];
        mask[(img_base + (int64_t)rb * TILE + tid) * nw + cb] = w;
    }
}

__global__ __launch_bounds__(1024) void nms_reduce_kernel(const unsigned long long* __restrict__ mask,
                                                          const int32_t* __restrict__ counts, int64_t nmax, int nw,
                                                          int64_t max_keep, int64_t* __restrict__ keep, int64_t keep_stride,
                                                          int32_t* __restrict__ num_keep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];   // nw words (+2 control words)
    const int img = blockIdx.x, tid = threadIdx.x;
    int64_t n = nmax;
    if (counts) { n = counts[img]; if (n > nmax) n = nmax; }
    mask += (int64_t)img * nmax * nw;
    keep += (int64_t)img * keep_stride;
    unsigned long long* ctl = remv + nw;      // ctl[0] = keep bits of the current chunk, ctl[1] = kept so far

    for (int i = tid; i < nw + 2; i += 1024) remv[i] = 0ull;
    __syncthreads();

    const int nchunks = (int)((n + TILE - 1) / TILE);
    // The diagonal word of chunk c + 1 does not depend on the decisions of chunk c: it is loaded at the top of chunk c and sits in a
    // register when its turn comes.  The kept rows of a chunk are known only after its diagonal is resolved; their words for all the
    // later chunks a thread covers are then requested TOGETHER (one memory round trip, not one per 64 chunks).  The first version paid
    // two dependent round trips per chunk (diagonal, then rows): 2 us x 157 chunks at 10 k boxes.  (Loading all 64 rows of the next chunk
    // ahead of time, kept or not, hides the second trip too but moves 12 words per thread and chunk through ONE CU's L2 port: slower on
    // clustered sets, where a tenth of the rows is kept.)
    constexpr int NPF = 3;                                     // words per row and thread requested together: covers nw <= 192 (12 288 boxes)
    const int g = tid >> 6, l = tid & 63;
    auto load_diag = [&](int c) -> unsigned long long {
        const int64_t row = (int64_t)c * TILE + tid;
        return (c < nchunks && tid < 64 && row < n) ? mask[row * nw + c] : 0ull;
    };
    unsigned long long dcur = load_diag(0);
    for (int c = 0; c < nchunks; c++) {
        const unsigned long long dnext = load_diag(c + 1);
        if (tid < 64) {
            const int64_t row = (int64_t)c * TILE + tid;
            const int valid = (int)min((int64_t)TILE, n - (int64_t)c * TILE);
            const unsigned long long d = dcur;
            unsigned long long cur = remv[c];
            const unsigned dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
            unsigned long long nz = __ballot(d != 0ull);
            // only rows that suppress something can change `cur`; visit them in ascending order
            while (nz) {
                const int j = __builtin_ctzll(nz);
                nz &= nz - 1;
                if (!((cur >> j) & 1ull)) {
                    const unsigned lo = __builtin_amdgcn_readlane(dlo, j);
                    const unsigned hi = __builtin_amdgcn_readlane(dhi, j);
                    cur |= ((unsigned long long)hi << 32) | lo;
                }
            }
            const unsigned long long vmask = valid >= 64 ? ~0ull : ((1ull << valid) - 1ull);
            const unsigned long long kb = ~cur & vmask;
            const int64_t base = (int64_t)ctl[1];
            if ((kb >> tid) & 1ull) {
                const int64_t pos = base + __popcll(kb & ((1ull << tid) - 1ull));
                if (pos < max_keep) keep[pos] = row;
            }
            if (tid == 0) { ctl[0] = kb; ctl[1] = (unsigned long long)(base + __popcll(kb)); }
        }
        __syncthreads();
        const unsigned long long kb = ctl[0];
        if ((int64_t)ctl[1] >= max_keep) break;
        // OR the kept rows of this chunk into the removed-bitmap of all later chunks
        unsigned long long pv[NPF][4];
#pragma unroll
        for (int it = 0; it < NPF; it++) {
            const int w = c + 1 + l + 64 * it;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int j = g + 16 * jj;
                pv[it][jj] = (w < nchunks && ((kb >> j) & 1ull)) ? mask[((int64_t)c * TILE + j) * nw + w] : 0ull;
            }
        }
#pragma unroll
        for (int it = 0; it < NPF; it++) {
            const int w = c + 1 + l + 64 * it;
            const unsigned long long v = pv[it][0] | pv[it][1] | pv[it][2] | pv[it][3];
            if (v) atomicOr(&remv[w], v);
        }
        for (int w = c + 1 + l + 64 * NPF; w < nchunks; w += 64) {          // very long rows (> 12 288 boxes): the tail, 64 chunks at a time
            unsigned long long v = 0ull;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int j = g + 16 * jj;
                if ((kb >> j) & 1ull) v |= mask[((int64_t)c * TILE + j) * nw + w];
            }
            if (v) atomicOr(&remv[w], v);
        }
        __syncthreads();
        dcur = dnext;
    }
    __syncthreads();
    if (tid == 0) {
        int64_t k = (int64_t)ctl[1];
        num_keep[img] = (int32_t)(k < max_keep ? k : max_keep);
    }
}

// IoU[N,M] (row-major) — test.py:135 of the reference (mAP matching).  64x64 tiles, same prune-then-compute idea.
__global__ __launch_bounds__(256) void pairwise_iou_kernel(const BoxPrep* __restrict__ p1, int n, const BoxPrep* __restrict__ p2, int m,
                                                           float* __restrict__ out)
{
    __shared__ BoxPrep rowb[TILE];
    __shared__ BoxPrep colb[TILE];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * TILE, c0 = blockIdx.x * TILE;
    const int nrow = min(TILE, n - r0), ncol = min(TILE, m - c0);
    load_tile(p1, r0, nrow, rowb, tid);
    __syncthreads();
    load_tile(p2, c0, ncol, colb, tid);
    __syncthreads();
    for (int k = tid; k < TILE * TILE; k += 256) {
        const int r = k >> 6, c = k & 63;
        if (r < nrow && c < ncol) {
            float v = 0.f;
            if (!boxes_far_apart(rowb[r], colb[c])) v = rotated_iou_pair(rowb[r], colb[c]);
            out[(int64_t)(r0 + r) * m + (c0 + c)] = v;
        }
    }
}

// element-wise (diagonal) SkewIoU: iou[i] = IoU(b1[i], b2[i]) — the dead-code tconf variant lib/loss.py:233-245
__global__ void diag_iou_kernel(const float* __restrict__ b1, const float* __restrict__ b2, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BoxPrep A, B;
    box_prep(b1 + 5 * i, A);
    box_prep(b2 + 5 * i, B);
    out[i] = rotated_iou_pair(A, B);
}

// -------------------------------------------------------------------------------------------------- C ABI
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int ryolo_nms_workspace_bytes(int batch, int64_t nmax, size_t* bytes)
{
    if (!bytes || batch < 0 || nmax < 0) return RY_ERR_ARG;
    const int64_t nw = ry_cdiv(nmax, TILE);
    *bytes = align256((size_t)batch * nmax * sizeof(BoxPrep)) + align256((size_t)batch * nmax * nw * 8) + 256;
    return RY_OK;
}

extern "C" int ryolo_nms_rotated_batched(const float* boxes, const int32_t* counts, int batch, int64_t nmax, float iou_thr,
                                         int gt_only, int64_t max_keep, void* ws, size_t ws_bytes, int64_t* keep,
                                         int64_t keep_stride, int32_t* num_keep, hipStream_t stream)
{
    if (batch < 0 || nmax < 0 || !num_keep) return RY_ERR_ARG;
    if (batch == 0) return RY_OK;
    if (nmax == 0) return hipMemsetAsync(num_keep, 0, sizeof(int32_t) * batch, stream) == hipSuccess ? RY_OK : RY_ERR_LAUNCH;
    if (!boxes || !keep || !ws) return RY_ERR_ARG;
    if (nmax > 65536 * 8) return RY_ERR_UNSUPPORTED;
    size_t need;
    ryolo_nms_workspace_bytes(batch, nmax, &need);
    if (ws_bytes < need) return RY_ERR_WORKSPACE;
    const int nw = (int)ry_cdiv(nmax, TILE);
    BoxPrep* prep = reinterpret_cast<BoxPrep*>(ws);
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(
        reinterpret_cast<char*>(ws) + align256((size_t)batch * nmax * sizeof(BoxPrep)));
    const int64_t total = (int64_t)batch * nmax;
    hipLaunchKernelGGL(nms_prep_kernel, dim3((unsigned)ry_cdiv(total, 256)), dim3(256), 0, stream, boxes, total, prep);
    const int prune = iou_thr > 1e-6f ? 1 : 0;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nw, nw, batch), dim3(256), 0, stream, prep, counts, nmax, nw, iou_thr, gt_only,
                       prune, mask);
    if (max_keep <= 0 || max_keep > nmax) max_keep = nmax;
    if (max_keep > keep_stride) max_keep = keep_stride;
    const size_t lds = (size_t)(nw + 2) * 8;
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(batch), dim3(1024), lds, stream, mask, counts, nmax, nw, max_keep, keep,
                       keep_stride, num_keep);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_box_iou_rotated(const float* b1, int n, const float* b2, int m, void* ws, size_t ws_bytes, float* out,
                                     hipStream_t stream)
{
    if (n < 0 || m < 0) return RY_ERR_ARG;
    if (n == 0 || m == 0) return RY_OK;
    if (!b1 || !b2 || !out || !ws) return RY_ERR_ARG;
    if (ws_bytes < align256((size_t)n * sizeof(BoxPrep)) + (size_t)m * sizeof(BoxPrep)) return RY_ERR_WORKSPACE;
    BoxPrep* p1 = reinterpret_cast<BoxPrep*>(ws);
    BoxPrep* p2 = reinterpret_cast<BoxPrep*>(reinterpret_cast<char*>(ws) + align256((size_t)n * sizeof(BoxPrep)));
    hipLaunchKernelGGL(nms_prep_kernel, dim3((unsigned)ry_cdiv(n, 256)), dim3(256), 0, stream, b1, (int64_t)n, p1);
    hipLaunchKernelGGL(nms_prep_kernel, dim3((unsigned)ry_cdiv(m, 256)), dim3(256), 0, stream, b2, (int64_t)m, p2);
    hipLaunchKernelGGL(pairwise_iou_kernel, dim3((unsigned)ry_cdiv(m, TILE), (unsigned)ry_cdiv(n, TILE)), dim3(256), 0, stream,
                       p1, n, p2, m, out);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

extern "C" int ryolo_diag_iou_rotated(const float* b1, const float* b2, int n, float* out, hipStream_t stream)
{
    if (n < 0) return RY_ERR_ARG;
    if (n == 0) return RY_OK;
    if (!b1 || !b2 || !out) return RY_ERR_ARG;
    hipLaunchKernelGGL(diag_iou_kernel, dim3((unsigned)ry_cdiv(n, 128)), dim3(128), 0, stream, b1, b2, n, out);
    RY_CHECK_LAUNCH();
    return RY_OK;
}
