"""Drop-in for the reference's lib/general.py hot-path functions, on MI355X HIP kernels.

Same names / arguments / return types as /root/reference/lib/general.py:
    post_process(predictions, conf_thres=0.5, iou_thres=0.4) -> list[Tensor[n_i, 7]]      (lib/general.py:136-183)
    norm_angle(theta), xywh2xyxy(x), xywhr2xywhrsigma(xywhr)                               (lib/general.py:7-133)
plus the two detectron2 ops the reference imports (lib/general.py:4, test.py:7):
    nms_rotated(boxes[N,5] deg, scores[N], iou_threshold) -> int64 keep indices
    pairwise_iou_rotated(boxes1[N,5], boxes2[M,5]) -> [N, M]
and the two polygon converters of the data / detect side, batched on the device:
    xyxyxyxy2xywha(boxes[N,8]) -> [N,5]                                                    (lib/general.py:70-104)
    xywha2xyxyxyxy(boxes[N,5]) -> [N,4,2]                                                  (lib/general.py:41-67)
"""
import numpy as np
import torch

from .. import hip

MAX_WH = 4096.0      # lib/general.py:147
MAX_NMS = 5000       # lib/general.py:148
MAX_DET = 1500       # lib/general.py:149


# ------------------------------------------------------------------------------------------ small tensor helpers
def norm_angle(theta):
    """lib/general.py:7-20.  The reference's range assert is a device->host sync; the two selects below already
    guarantee the post-condition for any |theta| < 3*pi/2, so it is dropped (SURVEY §8b 'sync points to remove')."""
    theta = torch.where(theta >= np.pi / 2, theta - np.pi, theta)
    theta = torch.where(theta < -np.pi / 2, theta + np.pi, theta)
    return theta


def xywh2xyxy(x):
    """lib/general.py:23-38"""
    assert isinstance(x, torch.Tensor), "Input should be torch.tensors."
    half = x[..., 2:4] / 2
    return torch.cat((x[..., 0:2] - half, x[..., 0:2] + half), -1)


def xywhr2xywhrsigma(xywhr):
    """lib/general.py:107-133 — (xy, wh clamped to [1e-4,1e4], r, Sigma = R diag((wh/2)^2) R^T).
    API-surface helper only: the KFIoU loss kernel (csrc/loss.hip) computes the closed form in registers."""
    shape = xywhr.size()
    assert shape[-1] == 5
    xy = xywhr[..., :2]
    wh = xywhr[..., 2:4].clamp(min=1e-4, max=1e4)
    r = xywhr[..., 4]
    c, s = torch.cos(r), torch.sin(r)
    a2, b2 = (0.5 * wh[..., 0]) ** 2, (0.5 * wh[..., 1]) ** 2
    sigma = torch.stack((c * c * a2 + s * s * b2, c * s * (a2 - b2), c * s * (a2 - b2), s * s * a2 + c * c * b2), -1)
    return xy, wh, r, sigma.reshape(shape[0], 2, 2)


# ------------------------------------------------------------------------------------------ rotated NMS / IoU ops
def _nms_sorted_batched(rboxes, counts, iou_thres, gt_only, max_keep):
    """rboxes [B,K,5] score-sorted; counts int32[B] or None.  Returns (keep int64[B,max_keep], num_keep int32[B])."""
    B, K = rboxes.shape[0], rboxes.shape[1]
    dev = rboxes.device
    need = hip._Z()
    hip.call("ryolo_nms_workspace_bytes", B, K, need)
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    mk = K if max_keep is None else min(max_keep, K)
    keep = torch.empty((B, max(mk, 1)), dtype=torch.int64, device=dev)
    num = torch.empty(B, dtype=torch.int32, device=dev)
    hip.call("ryolo_nms_rotated_batched", hip.ptr(rboxes), hip.ptr(counts), B, K, float(iou_thres), 1 if gt_only else 0, mk,
             hip.ptr(ws), need.value, hip.ptr(keep), keep.shape[1], hip.ptr(num), hip.stream())
    return keep, num


def argsort_desc(scores):
    """Stable descending argsort of a score vector on the device (csrc/topk.hip: LDS bitonic sort of (score, ~index) composites,
    global merge steps beyond 16 384): what detectron2's nms_rotated does first; equal scores keep ascending index order."""
    hip.require_device(scores, "argsort_desc")
    sc = scores.float().contiguous()
    n = sc.shape[0]
    order = torch.empty(n, dtype=torch.int64, device=sc.device)
    if n:
        need = hip._Z()
        hip.call("ryolo_sort_workspace_bytes", 1, n, need)
        ws = torch.empty(need.value, dtype=torch.uint8, device=sc.device)
        hip.call("ryolo_argsort_desc", hip.ptr(sc), n, hip.ptr(order), hip.ptr(ws), need.value, hip.stream())
    return order


def topk_desc(key, K):
    """key [B, M] fp32 -> (skey [B, K], order [B, K] int64): per row the K largest in (key desc, index asc) order, -inf entries never
    selected (padding: -inf / -1).  Radix select + LDS sort, one workgroup per row (csrc/topk.hip)."""
    hip.require_device(key, "topk_desc")
    B, M = key.shape
    skey = torch.empty((B, K), dtype=torch.float32, device=key.device)
    order = torch.empty((B, K), dtype=torch.int64, device=key.device)
    if B and K:
        need = hip._Z()
        hip.call("ryolo_sort_workspace_bytes", B, K, need)
        ws = torch.empty(need.value, dtype=torch.uint8, device=key.device)
        hip.call("ryolo_topk_desc", hip.ptr(key), B, M, K, hip.ptr(skey), hip.ptr(order), None, hip.ptr(ws), need.value, hip.stream())
    return skey, order


def nms_rotated(boxes, scores, iou_threshold, gt_only=True):
    """Drop-in for detectron2.layers.nms.nms_rotated (call site lib/general.py:177).
    gt_only=True suppresses iff IoU > thr (detectron2's CUDA kernel, what the reference runs on GPU);
    False gives the `>=` of detectron2's CPU kernel.  Ties in `scores` are broken by ascending index."""
    hip.require_device(boxes, "nms_rotated")
    if boxes.dim() != 2 or boxes.shape[1] != 5 or scores.shape[0] != boxes.shape[0]:
        raise RuntimeError("nms_rotated: boxes must be [N,5] and scores [N]")
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    order = argsort_desc(scores)
    sb = boxes.float()[order].contiguous().unsqueeze(0)
    keep, num = _nms_sorted_batched(sb, None, iou_threshold, gt_only, None)
    k = int(num.item())
    return order[keep[0, :k]]


def pairwise_iou_rotated(boxes1, boxes2):
    """Drop-in for detectron2.layers.rotated_boxes.pairwise_iou_rotated (call site test.py:135)."""
    hip.require_device(boxes1, "pairwise_iou_rotated")
    b1 = boxes1.float().contiguous()
    b2 = boxes2.float().contiguous()
    n, m = b1.shape[0], b2.shape[0]
    out = torch.empty((n, m), dtype=torch.float32, device=b1.device)
    if n and m:
        wsz = ((n * 48 + 255) // 256) * 256 + m * 48
        ws = torch.empty(wsz, dtype=torch.uint8, device=b1.device)
        hip.call("ryolo_box_iou_rotated", hip.ptr(b1), n, hip.ptr(b2), m, hip.ptr(ws), wsz, hip.ptr(out), hip.stream())
    return out


def diag_iou_rotated(boxes1, boxes2):
    """iou[i] = SkewIoU(boxes1[i], boxes2[i]) — element-wise variant for the loss-side score of lib/loss.py:233-245."""
    hip.require_device(boxes1, "diag_iou_rotated")
    b1, b2 = boxes1.float().contiguous(), boxes2.float().contiguous()
    out = torch.empty(b1.shape[0], dtype=torch.float32, device=b1.device)
    hip.call("ryolo_diag_iou_rotated", hip.ptr(b1), hip.ptr(b2), b1.shape[0], hip.ptr(out), hip.stream())
    return out


# ------------------------------------------------------------------------------------------ post_process
class PostProcessPlan:
    """post_process (lib/general.py:136-183) for a FIXED shape [B, M, nc + 6] on static, worst-case sized buffers: six C-ABI launches
    (score + filter, radix-select top-K, gather, NMS mask + on-device greedy reduce, emit), no allocation, no host read — the whole
    sequence can be captured in a hipGraph behind the forward (Yolo.capture_inference(..., post=...), BASELINE config C5).
    run(predictions) -> (out [B, max_det, 7] zero padded, num [B] int32 on the device)."""

    def __init__(self, B, M, nc, device, conf_thres=0.5, iou_thres=0.4, gt_only=True):
        self.B, self.M, self.nc, self.dev = B, M, nc, device
        self.conf_thres, self.iou_thres, self.gt_only = float(conf_thres), float(iou_thres), bool(gt_only)
        f32, i32 = torch.float32, torch.int32
        self.K = K = min(M, MAX_NMS)
        self.key = torch.empty((B, M), dtype=f32, device=device)
        self.cls = torch.empty((B, M), dtype=f32, device=device)
        self.count = torch.empty(B, dtype=i32, device=device)
        need = hip._Z()
        hip.call("ryolo_sort_workspace_bytes", B, K, need)
        self.sort_ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device=device)
        self.skey = torch.empty((B, K), dtype=f32, device=device)
        self.order = torch.empty((B, K), dtype=torch.int64, device=device)
        self.dets = torch.empty((B, K, 7), dtype=f32, device=device)
        self.rboxes = torch.empty((B, K, 5), dtype=f32, device=device)
        hip.call("ryolo_nms_workspace_bytes", B, K, need)
        self.nms_ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device=device)
        self.mk = max(min(MAX_DET, K), 1)
        self.keep = torch.empty((B, self.mk), dtype=torch.int64, device=device)
        self.num = torch.empty(B, dtype=i32, device=device)
        self.out = torch.empty((B, self.mk, 7), dtype=f32, device=device)

    def run(self, predictions, conf_thres=None, iou_thres=None, gt_only=None):
        """Thresholds default to the constructor's; per-call values do not touch the plan (a captured graph keeps what it was captured
        with).  `predictions[:, :, 6:]` is multiplied by the objectness IN PLACE, as lib/general.py:155 does."""
        B, M, nc, K, st = self.B, self.M, self.nc, self.K, hip.stream()
        conf = self.conf_thres if conf_thres is None else float(conf_thres)
        iou = self.iou_thres if iou_thres is None else float(iou_thres)
        gt = self.gt_only if gt_only is None else bool(gt_only)
        hip.call("ryolo_pp_score", hip.ptr(predictions), B, M, nc, conf, hip.ptr(self.key), hip.ptr(self.cls), hip.ptr(self.count), st)
        # the max_nms best candidates in (score desc, candidate index asc) order — the reference's argsort(descending=True)[:max_nms] at
        # lib/general.py:166-168 leaves tie order undefined; SURVEY §7 fixes it.  Radix select + LDS sort on the device (csrc/topk.hip)
        hip.call("ryolo_topk_desc", hip.ptr(self.key), B, M, K, hip.ptr(self.skey), hip.ptr(self.order), None, hip.ptr(self.sort_ws),
                 self.sort_ws.numel(), st)
        hip.call("ryolo_pp_gather", hip.ptr(predictions), hip.ptr(self.skey), hip.ptr(self.order), hip.ptr(self.cls), B, M, nc, K, K, MAX_WH,
                 hip.ptr(self.dets), hip.ptr(self.rboxes), hip.ptr(self.count), st)
        hip.call("ryolo_nms_rotated_batched", hip.ptr(self.rboxes), hip.ptr(self.count), B, K, iou, 1 if gt else 0, self.mk,
                 hip.ptr(self.nms_ws), self.nms_ws.numel(), hip.ptr(self.keep), self.mk, hip.ptr(self.num), st)
        hip.call("ryolo_pp_emit", hip.ptr(self.dets), hip.ptr(self.keep), hip.ptr(self.num), B, K, self.mk, hip.ptr(self.out), st)
        return self.out, self.num


_pp_plans = {}      # (B, M, nc, device, stream, thread) -> PostProcessPlan, insertion order = least recently used first
_PP_PLANS_MAX = 8


def post_process(predictions, conf_thres=0.5, iou_thres=0.4, gt_only=True):
    """lib/general.py:136-183.  predictions [B, M, nc+6] (x,y,w,h,theta_rad,obj,cls...) on the HIP device;
    predictions[:, :, 6:] is multiplied by the objectness IN PLACE exactly like the reference (:155).
    Returns a list of B tensors [n_i, 7] = (x, y, w, h, theta_rad, score, class).  One device->host read (the per-image counts: the
    list of variable-length tensors is the reference's return type); PostProcessPlan.run is the read-free form."""
    hip.require_device(predictions, "post_process")
    if predictions.dim() != 3 or predictions.shape[2] < 6:
        raise RuntimeError("post_process: predictions must be [B, M, nc+6]")
    if predictions.dtype != torch.float32 or not predictions.is_contiguous():
        raise RuntimeError("post_process: predictions must be contiguous float32")
    B, M, A = predictions.shape
    nc = A - 6
    dev = predictions.device
    empty = torch.zeros((0, 7), device=dev)
    if B == 0:
        return []
    if M == 0 or nc == 0:
        return [empty] * B
    # the static buffers of a plan belong to ONE stream of ONE host thread: launches of the same shape from another stream or thread get
    # their own plan instead of racing on key / dets / keep / out
    import threading
    key = (B, M, nc, dev.index, torch.cuda.current_stream(dev).cuda_stream, threading.get_ident())
    plan = _pp_plans.pop(key, None)
    if plan is None:
        while len(_pp_plans) >= _PP_PLANS_MAX:
            old_key = next(iter(_pp_plans))
            torch.cuda.synchronize(torch.device("cuda", old_key[3]))    # launches still reading the evicted buffers finish first
            del _pp_plans[old_key]
        plan = PostProcessPlan(B, M, nc, dev)
    _pp_plans[key] = plan                                                # most recently used last
    out, num = plan.run(predictions, conf_thres, iou_thres, gt_only)
    n_host = num.cpu().tolist()          # the single device->host read of the whole batch
    return [out[b, :n].clone() if n > 0 else empty for b, n in enumerate(n_host)]


# ------------------------------------------------------------------------------------------ polygon <-> rotated box
def xyxyxyxy2xywha(boxes):
    """lib/general.py:70-104: clockwise polygons [N, 8] -> (x, y, w, h, theta) [N, 5], h = long side, theta in [-pi/2, pi/2).
    One launch for the batch (the reference loops over boxes on the host); the range assert of norm_angle (a host sync) is dropped."""
    hip.require_device(boxes, "xyxyxyxy2xywha")
    src = boxes.float().contiguous()
    out = torch.empty((src.shape[0], 5), dtype=torch.float32, device=boxes.device)
    hip.call("ryolo_polys_to_xywha", hip.ptr(src), src.shape[0], hip.ptr(out), hip.stream())
    return out


def xywha2xyxyxyxy(boxes):
    """lib/general.py:41-67: (x, y, w, h, theta rad) [N, 5] -> vertices [N, 4, 2] (h spans x, w spans y before the rotation, as
    in the reference); cv2.getRotationMatrix2D restated in double on the device."""
    hip.require_device(boxes, "xywha2xyxyxyxy")
    n = boxes.shape[0]
    d = torch.zeros((n, 7), dtype=torch.float32, device=boxes.device)
    d[:, :5] = boxes[:, :5].float()
    polys = torch.empty((n, 4, 2), dtype=torch.float32, device=boxes.device)
    hip.call("ryolo_dets_to_polys", hip.ptr(d), None, None, 0, 0, n, hip.ptr(polys), hip.stream())
    return polys
