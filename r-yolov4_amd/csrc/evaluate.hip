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
