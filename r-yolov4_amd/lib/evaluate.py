"""mAP evaluation of the reference's test.py on the HIP path.

`get_batch_statistics` (test.py:102-149) — the per-image / per-class Python loops with a detectron2 `pairwise_iou_rotated`
call and `.item()` syncs each — is ONE kernel launch for the batch (csrc/evaluate.hip) and one device->host read.
`ap_per_class` / `compute_ap` / `calculate_eval_stats` (test.py:16-99,152-164) are host-side numpy in the reference (they run
once per evaluation on a few thousand rows) and are kept as host code with the same signatures and return values.
"""
import ctypes

import numpy as np
import torch

from .. import hip


def get_batch_statistics(outputs, targets, iouv, niou):
    """Same contract as test.py:102: `outputs` = post_process result (list of [n_i, 7] score-descending tensors on the HIP
    device), `targets` [nt, 7] = (img, cls, x, y, w, h, theta_rad) in pixels, `iouv` [niou] IoU thresholds.  Returns the list of
    (true_positives bool [n_i, niou] cpu, scores cpu, labels cpu, target-class list); images with neither predictions nor labels
    are skipped (:112-115).  The reference's in-place radians -> degrees conversion of outputs[i][:, 4] (:126) is kept."""
    B = len(outputs)
    dev = None
    for o in outputs:
        if len(o):
            hip.require_device(o, "get_batch_statistics")
            dev = o.device
    tg = targets.float()
    timg = tg[:, 0].long().cpu() if tg.numel() else torch.zeros(0, dtype=torch.long)
    tcls_all = tg[:, 1].cpu() if tg.numel() else torch.zeros(0)
    order = torch.argsort(timg, stable=True)                   # group by image, keep the caller's order inside an image
    tcnt = torch.bincount(timg, minlength=B)[:B] if timg.numel() else torch.zeros(B, dtype=torch.long)
    toff = torch.zeros(B + 1, dtype=torch.long)
    toff[1:] = torch.cumsum(tcnt, 0)
    pcnt = torch.tensor([len(o) for o in outputs], dtype=torch.long)
    poff = torch.zeros(B + 1, dtype=torch.long)
    poff[1:] = torch.cumsum(pcnt, 0)
    npred, ntgt = int(poff[-1]), int(toff[-1])
    tp_host = scores_host = labels_host = None
    if npred:
        preds = torch.cat([o.float().reshape(-1, 7) for o in outputs if len(o)], 0).contiguous()
        tgs = tg.to(dev)[order.to(dev)].contiguous() if ntgt else torch.zeros((0, 7), device=dev)
        iou_d = torch.as_tensor(iouv, dtype=torch.float32).to(dev).contiguous()
        tp = torch.empty((npred, niou), dtype=torch.uint8, device=dev)
        need = ctypes.c_size_t()
        hip.call("ryolo_map_match_workspace_bytes", npred, ntgt, need)
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        ncls = int(max(float(preds[:, 6].max()), float(tgs[:, 1].max()) if ntgt else 0.0)) + 1
        poff_d, toff_d = poff.to(dev), toff.to(dev)            # named: the device copies must outlive the launch
        hip.call("ryolo_map_match", hip.ptr(preds), hip.ptr(poff_d), hip.ptr(tgs) if ntgt else None, hip.ptr(toff_d), B, npred, ntgt,
                 hip.ptr(iou_d), int(niou), ncls, hip.ptr(tp), hip.ptr(ws), need.value, hip.stream())
        # side effect of the reference: theta in degrees, in place, for images that have labels (and predictions)
        for b, o in enumerate(outputs):
            if len(o) and int(tcnt[b]):
                o[:, 4] = preds[int(poff[b]):int(poff[b + 1]), 4].to(o.dtype)
        host = torch.cat([tp.float(), preds[:, 5:7]], 1).cpu()  # the single device -> host read
        tp_host, scores_host, labels_host = host[:, :niou].bool(), host[:, niou], host[:, niou + 1]
    stats = []
    order_l = order.tolist()
    for b in range(B):
        tcls = [float(tcls_all[j]) for j in order_l[int(toff[b]):int(toff[b + 1])]]
        if int(pcnt[b]) == 0:
            if tcls:
                stats.append((np.zeros((0, niou), dtype=bool), np.empty(0), np.empty(0), tcls))
            continue
        s, e = int(poff[b]), int(poff[b + 1])
        stats.append((tp_host[s:e], scores_host[s:e], labels_host[s:e], tcls))
    return stats


def compute_ap(recall, precision):
    """test.py:73-99 (YOLOv7 metric: sentinels, precision envelope, 101-point interpolation)."""
    mrec = np.concatenate(([0.0], recall, [recall[-1] + 0.01]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    ap = np.trapz(np.interp(x, mrec, mpre), x)
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls):
    """test.py:16-70.  (p, r, ap [nc, niou], f1, classes) at the confidence of maximum mean F1."""
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes = np.unique(target_cls)
    nc = unique_classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap, p, r = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        i = pred_cls == c
        n_l = (target_cls == c).sum()
        n_p = i.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[i]).cumsum(0)
        tpc = tp[i].cumsum(0)
        recall = tpc / (n_l + 1e-16)
        r[ci] = np.interp(-px, -conf[i], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p[ci] = np.interp(-px, -conf[i], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], _, _ = compute_ap(recall[:, j], precision[:, j])
    f1 = 2 * p * r / (p + r + 1e-16)
    i = f1.mean(0).argmax()
    return p[:, i], r[:, i], ap, f1[:, i], unique_classes.astype("int32")


def calculate_eval_stats(stats, num_classes):
    """test.py:152-164: (nt, p, r, ap50, ap, f1, ap_class, mp, mr, map50, map) from the concatenated statistics."""
    p, r, f1, mp, mr, map50, map_ = 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0
    ap50, ap, ap_class = [], [], []
    if len(stats) and stats[0].any():
        p, r, ap, f1, ap_class = ap_per_class(*stats)
        ap50, ap = ap[:, 0], ap.mean(1)
        mp, mr, map50, map_ = p.mean(), r.mean(), ap50.mean(), ap.mean()
        nt = np.bincount(stats[3].astype(np.int64), minlength=num_classes)
    else:
        nt = torch.zeros(1)
    return nt, p, r, ap50, ap, f1, ap_class, mp, mr, map50, map_
