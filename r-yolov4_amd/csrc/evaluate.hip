// mAP evaluation, device side: the per-image true-positive matching of the reference's
// `get_batch_statistics` (test.py:102-149).  The reference loops over images and classes on the host, calls detectron2's
// pairwise_iou_rotated per (image, class) and walks the matches in Python with `.item()` syncs; here one workgroup per image
// does it in a single launch for the whole batch:
//   phase 1  every prediction finds its best-IoU label of the SAME class (first maximum, like torch.max) with the exact
//            pair function shared with NMS (rotated_iou.h; this file is compiled without FMA contraction like nms.hip);
//   phase 2  one thread per class walks that class's predictions in score order (post_process order) and claims labels:
//            a prediction is a TP at threshold k iff its best label is still free and IoU > iouv[k]; a prediction whose best
//            label is taken stays a false positive (the reference never falls back to the second-best label, :138-141).
// Side effect kept (test.py:126): theta of the predictions becomes DEGREES in place when the image has labels.
#include "common.h"
#include "rotated_iou.h"

#define MAP_MAX_CLASSES 256

__global__ __launch_bounds__(256) void map_match_kernel(float* __restrict__ preds, const int64_t* __restrict__ poff, const float* __restrict__ tg,
                                                        const int64_t* __restrict__ toff, const float* __restrict__ iouv, int niou,
                                                        unsigned char* __restrict__ tp, float* __restrict__ best_iou, int* __restrict__ best_t,
                                                        unsigned char* __restrict__ taken)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t p0 = poff[b], t0 = toff[b];
    const int n = (int)(poff[b + 1] - p0), nl = (int)(toff[b + 1] - t0);
    for (int k = tid; k < n * niou; k += 256) tp[p0 * niou + k] = 0;
    if (n == 0 || nl == 0) return;
    const float PI_F = 3.14159274f;                                   // float32(np.pi): `pred_boxes[:, 4] / np.pi * 180` on a float32 tensor
    for (int i = tid; i < n; i += 256) preds[(p0 + i) * 7 + 4] = preds[(p0 + i) * 7 + 4] / PI_F * 180.f;
    for (int t = tid; t < nl; t += 256) taken[t0 + t] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const float* pr = preds + (p0 + i) * 7;
        const float cls = pr[6];
        BoxPrep P;
        box_prep(pr, P);
        float bi = -1.f;
        int bt = -1;
        for (int t = 0; t < nl; t++) {
            const float* tr = tg + (t0 + t) * 7;
            if (tr[1] != cls) continue;
            const float tb[5] = {tr[2], tr[3], tr[4], tr[5], tr[6] / PI_F * 180.f};
            BoxPrep T;
            box_prep(tb, T);
            const float v = boxes_far_apart(P, T) ? 0.f : rotated_iou_pair(P, T);
            if (v > bi) { bi = v; bt = t; }
        }
        best_iou[p0 + i] = bi;
        best_t[p0 + i] = bt;
    }
    __syncthreads();
    // labels of different classes are disjoint, so the class threads never touch the same `taken` byte
    const float c = (float)tid, thr0 = iouv[0];
    for (int i = 0; i < n; i++) {
        if (preds[(p0 + i) * 7 + 6] != c) continue;
        const int bt = best_t[p0 + i];
        const float bi = best_iou[p0 + i];
        if (bt < 0 || !(bi > thr0) || taken[t0 + bt]) continue;
        taken[t0 + bt] = 1;
        for (int k = 0; k < niou; k++) tp[(p0 + i) * niou + k] = bi > iouv[k] ? 1 : 0;
    }
}

extern "C" int ryolo_map_match_workspace_bytes(int64_t npred, int64_t ntgt, size_t* bytes)
{
    if (!bytes || npred < 0 || ntgt < 0) return RY_ERR_ARG;
    *bytes = (size_t)npred * 8 + (size_t)ntgt + 256;
    return RY_OK;
}

extern "C" int ryolo_map_match(float* preds, const int64_t* pred_off, const float* targets, const int64_t* tgt_off, int batch, int64_t npred,
                               int64_t ntgt, const float* iouv, int niou, int num_classes, unsigned char* tp, void* ws, size_t ws_bytes,
                               hipStream_t stream)
{
    if (batch < 0 || npred < 0 || ntgt < 0 || niou < 1) return RY_ERR_ARG;
    if (num_classes > MAP_MAX_CLASSES) return RY_ERR_UNSUPPORTED;
    if (batch == 0 || npred == 0) return RY_OK;
    if (!preds || !pred_off || !tgt_off || !iouv || !tp || !ws || (ntgt > 0 && !targets)) return RY_ERR_ARG;
    if (ws_bytes < (size_t)npred * 8 + (size_t)ntgt + 256) return RY_ERR_WORKSPACE;
    float* best_iou = reinterpret_cast<float*>(ws);
    int* best_t = reinterpret_cast<int*>(best_iou + npred);
    unsigned char* taken = reinterpret_cast<unsigned char*>(best_t + npred);
    hipLaunchKernelGGL(map_match_kernel, dim3((unsigned)batch), dim3(256), 0, stream, preds, pred_off, targets, tgt_off, iouv, niou, tp, best_iou,
                       best_t, taken);
    RY_CHECK_LAUNCH();
    return RY_OK;
}

// ------------------------------------------------------------------------------------------------ AP from the statistics
// `ap_per_class` / `compute_ap` of the reference (test.py:16-99) on the device: the confidence sort (csrc/topk.hip), the per-class
// cumulative hit counts, recall / precision curves, the precision envelope, the 101-point interpolated integral for every IoU threshold
// and the 1000-point precision / recall curves over confidence.  What stays on the host is O(classes x 1000): F1, its mean, the argmax.
// The floating-point expressions are numpy's, restated: float64 throughout, `hits / (n_labels + 1e-16)`, `hits / rank`, np.interp
// (last knot <= x, slope * (x - xp[j]) + fp[j], exact hit returns fp[j]), np.trapz = pairwise-summed d * (y1 + y0) / 2 — compiled
// without FMA contraction like the rest of this file, so the numbers equal the host path's (lib/evaluate.py) bit for bit on the fixture.
#define AP_MAX_T 16
#define AP_THREADS 1024

__global__ __launch_bounds__(AP_THREADS) void ap_count_kernel(const float* __restrict__ pcls, int64_t n, const float* __restrict__ tcls, int64_t nl, int nc,
                                                              int64_t* __restrict__ cnt, int64_t* __restrict__ nlab, int64_t* __restrict__ off)
{
    __shared__ int c_p[MAP_MAX_CLASSES], c_l[MAP_MAX_CLASSES];
    const int tid = threadIdx.x;
    for (int k = tid; k < nc; k += AP_THREADS) { c_p[k] = 0; c_l[k] = 0; }
    __syncthreads();
    for (int64_t i = tid; i < n; i += AP_THREADS) {
        const float c = pcls[i];
        const int k = (int)c;
        if (k >= 0 && k < nc && (float)k == c) atomicAdd(&c_p[k], 1);
    }
    for (int64_t i = tid; i < nl; i += AP_THREADS) {
        const float c = tcls[i];
        const int k = (int)c;
        if (k >= 0 && k < nc && (float)k == c) atomicAdd(&c_l[k], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int k = 0; k < nc; k++) { cnt[k] = c_p[k]; nlab[k] = c_l[k]; off[k] = run; run += c_p[k]; }
    }
}

// one workgroup per class: walk the detections in confidence order, scan the class's hits per threshold, write the curves
__global__ __launch_bounds__(AP_THREADS) void ap_curves_kernel(const unsigned char* __restrict__ tp, const float* __restrict__ conf,
                                                               const float* __restrict__ pcls, const int64_t* __restrict__ order, int64_t n, int T,
                                                               const int64_t* __restrict__ cnt, const int64_t* __restrict__ nlab,
                                                               const int64_t* __restrict__ off, double* __restrict__ negc,
                                                               double* __restrict__ rec, double* __restrict__ prec)
{
    __shared__ unsigned wt[16][AP_MAX_T + 1], wp[16][AP_MAX_T + 1], tot[AP_MAX_T + 1];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (cnt[k] == 0 || nlab[k] == 0) return;
    const double denom = (double)nlab[k] + 1e-16;
    const int64_t o0 = off[k];
    const float kc = (float)k;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int64_t run[AP_MAX_T + 1];
#pragma unroll
    for (int q = 0; q <= AP_MAX_T; q++) run[q] = 0;
    for (int64_t i0 = 0; i0 < n; i0 += AP_THREADS) {
        const int64_t i = i0 + tid;
        const int64_t idx = i < n ? order[i] : 0;
        const bool f0 = i < n && pcls[idx] == kc;
        unsigned pre[AP_MAX_T + 1];
        bool fl[AP_MAX_T + 1];
#pragma unroll
        for (int q = 0; q <= AP_MAX_T; q++) {
            fl[q] = q == 0 ? f0 : (q <= T && f0 && tp[idx * T + (q - 1)] != 0);
            const unsigned long long b = __ballot(fl[q]);
            pre[q] = (unsigned)__popcll(b & lt);
            if (lane == 0) wt[wave][q] = (unsigned)__popcll(b);
        }
        __syncthreads();
        if (tid <= T) {
            unsigned r = 0;
            for (int w = 0; w < 16; w++) { wp[w][tid] = r; r += wt[w][tid]; }
            tot[tid] = r;
        }
        __syncthreads();
        if (f0) {
            const int64_t pos = run[0] + wp[wave][0] + pre[0];
            negc[o0 + pos] = (double)(-conf[idx]);
#pragma unroll
            for (int q = 1; q <= AP_MAX_T; q++)
                if (q <= T) {
                    const int64_t hits = run[q] + wp[wave][q] + pre[q] + (fl[q] ? 1 : 0);
                    rec[(int64_t)(q - 1) * n + o0 + pos] = (double)hits / denom;
                    prec[(int64_t)(q - 1) * n + o0 + pos] = (double)hits / (double)(pos + 1);
                }
        }
#pragma unroll
        for (int q = 0; q <= AP_MAX_T; q++)
            if (q <= T) run[q] += tot[q];
        __syncthreads();
    }
}

// np.interp(x, xp, fp, left, right) for ONE x over virtual knot arrays given as functors (length L >= 1, xp non-decreasing)
template <class XP, class FP>
__device__ __forceinline__ double np_interp1(double x, int64_t L, XP xp, FP fp, double left, double right)
{
    if (x > xp(L - 1)) return right;
    if (x < xp(0)) return left;
    int64_t lo = 0, hi = L;                                       // j = (number of knots <= x) - 1
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (x >= xp(mid)) lo = mid + 1; else hi = mid;
    }
    const int64_t j = lo - 1;
    if (j == L - 1) return fp(j);
    const double xj = xp(j);
    if (xj == x) return fp(j);
    const double slope = (fp(j + 1) - fp(j)) / (xp(j + 1) - xj);
    double r = slope * (x - xj) + fp(j);
    if (r != r) {
        r = slope * (x - xp(j + 1)) + fp(j + 1);
        if (r != r && fp(j) == fp(j + 1)) r = fp(j);
    }
    return r;
}

// numpy's pairwise summation of a contiguous double vector, 8 <= n <= 128 (one block of the recursion) or n < 8
__device__ __forceinline__ double np_pairwise_sum(const double* a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

// grid (T, classes): precision envelope (running maximum from the right), 101-point interpolation, trapezoid
__global__ __launch_bounds__(AP_THREADS) void ap_integrate_kernel(int64_t n, int T, const int64_t* __restrict__ cnt, const int64_t* __restrict__ nlab,
                                                                  const int64_t* __restrict__ off, const double* __restrict__ rec,
                                                                  const double* __restrict__ prec, double* __restrict__ env,
                                                                  const double* __restrict__ grid101, double* __restrict__ ap)
{
    __shared__ double wmax[16], carry_s, y[101], term[100];
    const int t = blockIdx.x, k = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m = cnt[k];
    if (m == 0 || nlab[k] == 0) { if (tid == 0) ap[(int64_t)k * T + t] = 0.0; return; }
    const double* r_ = rec + (int64_t)t * n + off[k];
    const double* p_ = prec + (int64_t)t * n + off[k];
    double* e_ = env + (int64_t)t * n + off[k];
    if (tid == 0) carry_s = 0.0;                                   // the appended knot (recall[-1] + 0.01, precision 0)
    __syncthreads();
    for (int64_t hi = m; hi > 0; hi -= AP_THREADS) {               // chunks from the right end
        const int64_t i = hi - AP_THREADS + tid;                   // this chunk covers [hi - 1024, hi)
        double v = i >= 0 ? p_[i] : 0.0;
        for (int d = 1; d < 64; d <<= 1) {                         // suffix maximum inside the wave
            const double o = __shfl_down(v, d, 64);
            if (lane + d < 64) v = fmax(v, o);
        }
        if (lane == 0) wmax[wave] = v;
        __syncthreads();
        double later = carry_s;
        for (int w = wave + 1; w < 16; w++) later = fmax(later, wmax[w]);
        v = fmax(v, later);
        if (i >= 0) e_[i] = v;
        __syncthreads();
        if (tid == 0) {
            double c = carry_s;
            for (int w = 0; w < 16; w++) c = fmax(c, wmax[w]);
            carry_s = c;
        }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    const double env0 = fmax(1.0, carry_s);                        // leading knot (recall 0, precision 1)
    const double rlast = r_[m - 1] + 0.01;
    auto xp = [&](int64_t j) -> double { return j == 0 ? 0.0 : (j <= m ? r_[j - 1] : rlast); };
    auto fp = [&](int64_t j) -> double { return j == 0 ? env0 : (j <= m ? e_[j - 1] : 0.0); };
    if (tid < 101) y[tid] = np_interp1(grid101[tid], m + 2, xp, fp, fp(0), fp(m + 1));
    __syncthreads();
    if (tid < 100) term[tid] = (grid101[tid + 1] - grid101[tid]) * (y[tid + 1] + y[tid]) / 2.0;
    __syncthreads();
    if (tid == 0) ap[(int64_t)k * T + t] = np_pairwise_sum(term, 100);
}

// precision / recall at the first IoU threshold as functions of confidence: np.interp(-grid, -conf, curve, left = 0 / 1)
__global__ __launch_bounds__(AP_THREADS) void ap_pr_kernel(int64_t n, const int64_t* __restrict__ cnt, const int64_t* __restrict__ nlab,
                                                           const int64_t* __restrict__ off, const double* __restrict__ negc, const double* __restrict__ rec,
                                                           const double* __restrict__ prec, const double* __restrict__ cgrid, int ng,
                                                           double* __restrict__ prec_at, double* __restrict__ rec_at)
{
    const int k = blockIdx.x;
    const int64_t m = cnt[k];
    const bool dead = m == 0 || nlab[k] == 0;
    const double* x_ = negc + off[k];
    const double* r_ = rec + off[k];
    const double* p_ = prec + off[k];
    auto xp = [&](int64_t j) -> double { return x_[j]; };
    auto fr = [&](int64_t j) -> double { return r_[j]; };
    auto fq = [&](int64_t j) -> double { return p_[j]; };
    for (int i = threadIdx.x; i < ng; i += AP_THREADS) {
        double pr = 0.0, rc = 0.0;
        if (!dead) {
            const double x = -cgrid[i];
            rc = np_interp1(x, m, xp, fr, 0.0, r_[m - 1]);
            pr = np_interp1(x, m, xp, fq, 1.0, p_[m - 1]);
        }
        prec_at[(int64_t)k * ng + i] = pr;
        rec_at[(int64_t)k * ng + i] = rc;
    }
}

static inline size_t ap_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int ryolo_sort_workspace_bytes(int batch, int64_t n, size_t* bytes);
extern "C" int ryolo_argsort_desc(const float* scores, int64_t N, int64_t* order, void* ws, size_t ws_bytes, hipStream_t stream);

extern "C" int ryolo_ap_workspace_bytes(int64_t n, int niou, int nc, size_t* bytes)
{
    if (!bytes || n < 0 || niou < 1 || niou > AP_MAX_T || nc < 1 || nc > MAP_MAX_CLASSES) return RY_ERR_ARG;
    size_t sort = 0;
    if (n > 0 && ryolo_sort_workspace_bytes(1, n, &sort) != RY_OK) return RY_ERR_ARG;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    *bytes = ap_align(sort) + ap_align(nn * 8) + ap_align((size_t)3 * nc * 8) + ap_align(nn * 8) + 3 * ap_align(nn * niou * 8);
    return RY_OK;
}

extern "C" int ryolo_ap_per_class(const unsigned char* tp, const float* conf, const float* pred_cls, int64_t n, const float* target_cls, int64_t nl,
                                  int nc, int niou, const double* recall_grid, const double* conf_grid, int nconf, void* ws, size_t ws_bytes,
                                  double* ap, double* prec_at, double* rec_at, int64_t* n_labels, int64_t* n_pred, hipStream_t stream)
{
    size_t need = 0;
    if (ryolo_ap_workspace_bytes(n, niou, nc, &need) != RY_OK) return RY_ERR_ARG;
    if (!ws || ws_bytes < need) return RY_ERR_WORKSPACE;
    if (!ap || !prec_at || !rec_at || !n_labels || !n_pred || !recall_grid || !conf_grid || nconf < 1 || nl < 0) return RY_ERR_ARG;
    if (n > 0 && (!tp || !conf || !pred_cls)) return RY_ERR_ARG;
    if (nl > 0 && !target_cls) return RY_ERR_ARG;
    size_t sort = 0;
    if (n > 0) ryolo_sort_workspace_bytes(1, n, &sort);
    const size_t nn = (size_t)(n > 0 ? n : 1);
    unsigned char* base = reinterpret_cast<unsigned char*>(ws);
    void* sort_ws = base;                              base += ap_align(sort);
    int64_t* order = reinterpret_cast<int64_t*>(base); base += ap_align(nn * 8);
    int64_t* off = reinterpret_cast<int64_t*>(base);   base += ap_align((size_t)3 * nc * 8);
    double* negc = reinterpret_cast<double*>(base);    base += ap_align(nn * 8);
    double* rec = reinterpret_cast<double*>(base);     base += ap_align(nn * niou * 8);
    double* prec = reinterpret_cast<double*>(base);    base += ap_align(nn * niou * 8);
    double* env = reinterpret_cast<double*>(base);
    if (n > 0) {
        const int rc = ryolo_argsort_desc(conf, n, order, sort_ws, sort, stream);       // (confidence desc, index asc): np.argsort(-conf) with ties fixed
        if (rc) return rc;
    }
    hipLaunchKernelGGL(ap_count_kernel, dim3(1), dim3(AP_THREADS), 0, stream, pred_cls, n, target_cls, nl, nc, n_pred, n_labels, off);
    hipLaunchKernelGGL(ap_curves_kernel, dim3(nc), dim3(AP_THREADS), 0, stream, tp, conf, pred_cls, order, n, niou, n_pred, n_labels, off, negc, rec, prec);
    hipLaunchKernelGGL(ap_integrate_kernel, dim3(niou, nc), dim3(AP_THREADS), 0, stream, n, niou, n_pred, n_labels, off, rec, prec, env, recall_grid, ap);
    hipLaunchKernelGGL(ap_pr_kernel, dim3(nc), dim3(AP_THREADS), 0, stream, n, n_pred, n_labels, off, negc, rec, prec, conf_grid, nconf, prec_at, rec_at);
    RY_CHECK_LAUNCH();
    return RY_OK;
}
