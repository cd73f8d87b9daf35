"""mAP evaluation of the reference's test.py on the HIP path.

`get_batch_statistics` (test.py:102-149) — the per-image / per-class Python loops with a detectron2 `pairwise_iou_rotated`
call and `.item()` syncs each — is ONE kernel launch for the batch (csrc/evaluate.hip) and one device->host read.
`ap_per_class` / `compute_ap` / `calculate_eval_stats` (test.py:16-99,152-164): `calculate_eval_stats` runs the AP computation on the
device (`ap_per_class_device` -> ryolo_ap_per_class: confidence sort, cumulative curves, envelope, 101-point integral, 1000-point
precision / recall curves; numpy's float64 expressions restated, bit-identical on fixture G8); `ap_per_class` / `compute_ap` are the
host-side numpy restatement with the reference's signatures and return values, written independently (class-grouped, all IoU
thresholds at once) — what the CPU tests pin to the fixture and the device path is compared with.
"""
import ctypes

import numpy as np
import torch

from .. import hip

MAP_MAX_CLASSES = 256      # csrc/evaluate.hip


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


# ------------------------------------------------------------------------------------------------ AP from the statistics
# Same signatures and return values as test.py:16-99,152-164, written independently: predictions are grouped by class with ONE
# stable sort (confidence order survives inside a class), the cumulative hit / miss counts come from a single cumsum (misses =
# rank - hits, both exact integers), and the precision envelope + 101-point integral are evaluated for all IoU thresholds of a
# class at once.  The floating-point expressions that define the metric (hits / (n_labels + 1e-16), hits / rank, np.interp on the
# negated confidences, trapezoid over 101 points) are the reference's, so the results are bit-identical (fixture G8).
_RECALL_GRID = np.linspace(0.0, 1.0, 101)
_CONF_GRID = np.linspace(0.0, 1.0, 1000)


def _envelope_ap(recall, precision):
    """[n, T] recall / precision curves (one column per IoU threshold) -> (ap [T], envelope [n + 2, T], recall knots [n + 2, T])."""
    n, T = recall.shape
    knots = np.empty((n + 2, T))
    knots[0], knots[1:-1], knots[-1] = 0.0, recall, recall[-1] + 0.01
    env = np.empty((n + 2, T))
    env[0], env[1:-1], env[-1] = 1.0, precision, 0.0
    env = np.maximum.accumulate(env[::-1], axis=0)[::-1]            # running maximum from the right = precision envelope
    ap = np.array([np.trapz(np.interp(_RECALL_GRID, knots[:, j], env[:, j]), _RECALL_GRID) for j in range(T)])
    return ap, env, knots


def compute_ap(recall, precision):
    """test.py:73-99 for one curve: (ap, envelope, recall knots)."""
    ap, env, knots = _envelope_ap(np.asarray(recall, dtype=np.float64)[:, None], np.asarray(precision, dtype=np.float64)[:, None])
    return ap[0], env[:, 0], knots[:, 0]


def ap_per_class(tp, conf, pred_cls, target_cls):
    """test.py:16-70.  (p, r, ap [nc, niou], f1, classes), p / r / f1 read at the confidence that maximises the mean F1."""
    by_conf = np.argsort(-conf)
    tp, conf, pred_cls = tp[by_conf], conf[by_conf], pred_cls[by_conf]
    classes, n_labels = np.unique(target_cls, return_counts=True)
    by_cls = np.argsort(pred_cls, kind="stable")
    lo = np.searchsorted(pred_cls[by_cls], classes, side="left")
    hi = np.searchsorted(pred_cls[by_cls], classes, side="right")
    niou = tp.shape[1]
    ap = np.zeros((len(classes), niou))
    prec_at, rec_at = np.zeros((len(classes), _CONF_GRID.size)), np.zeros((len(classes), _CONF_GRID.size))
    for k in range(len(classes)):
        rows = by_cls[lo[k]:hi[k]]
        if rows.size == 0 or n_labels[k] == 0:
            continue
        hits = tp[rows].cumsum(0)
        rank = np.arange(1, rows.size + 1)[:, None]                  # hits + misses
        recall = hits / (n_labels[k] + 1e-16)
        precision = hits / rank
        neg = -conf[rows]
        rec_at[k] = np.interp(-_CONF_GRID, neg, recall[:, 0], left=0)
        prec_at[k] = np.interp(-_CONF_GRID, neg, precision[:, 0], left=1)
        ap[k] = _envelope_ap(recall, precision)[0]
    f1 = 2 * prec_at * rec_at / (prec_at + rec_at + 1e-16)
    best = f1.mean(0).argmax()
    return prec_at[:, best], rec_at[:, best], ap, f1[:, best], classes.astype("int32")


def ap_per_class_device(tp, conf, pred_cls, target_cls, num_classes=None, device=None):
    """ap_per_class with the sort, the cumulative curves, the envelope, the 101-point integral and the 1000-point precision / recall
    curves computed on the HIP device (csrc/evaluate.hip: ryolo_ap_per_class); inputs numpy arrays or tensors as ap_per_class takes
    them.  Only the O(classes x 1000) tail (F1, its mean over classes, the argmax) runs on the host.  Same return tuple.  Equal
    confidences are ordered by index (np.argsort(-conf) leaves their order to the sort implementation)."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    tp_t = torch.as_tensor(np.ascontiguousarray(np.asarray(tp)).astype(np.uint8)).to(dev)
    conf_t = torch.as_tensor(np.asarray(conf, dtype=np.float32)).to(dev).contiguous()
    pcls_t = torch.as_tensor(np.asarray(pred_cls, dtype=np.float32)).to(dev).contiguous()
    tcls_np = np.asarray(target_cls, dtype=np.float32)
    tcls_t = torch.as_tensor(tcls_np).to(dev).contiguous()
    n, niou = int(tp_t.shape[0]), int(tp_t.shape[1])
    nc = int(num_classes) if num_classes else int(max(tcls_np.max(initial=0), np.asarray(pred_cls, dtype=np.float32).max(initial=0))) + 1
    if nc > MAP_MAX_CLASSES:
        raise RuntimeError(f"ap_per_class_device: {nc} classes > {MAP_MAX_CLASSES} (one workgroup per class, class table in LDS); use ap_per_class / host=True")
    if tcls_np.size and (tcls_np.max() >= nc or tcls_np.min() < 0):
        # the numpy path counts every label class it meets (np.unique); dropping out-of-range labels here would silently change nt / recall
        raise ValueError(f"ap_per_class_device: target class outside [0, {nc})")
    grid = torch.as_tensor(_RECALL_GRID).to(dev)
    cgrid = torch.as_tensor(_CONF_GRID).to(dev)
    need = hip._Z()
    hip.call("ryolo_ap_workspace_bytes", n, niou, nc, need)
    ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device=dev)
    ap = torch.empty((nc, niou), dtype=torch.float64, device=dev)
    prec_at = torch.empty((nc, _CONF_GRID.size), dtype=torch.float64, device=dev)
    rec_at = torch.empty_like(prec_at)
    nlab = torch.empty(nc, dtype=torch.int64, device=dev)
    npred = torch.empty(nc, dtype=torch.int64, device=dev)
    hip.call("ryolo_ap_per_class", hip.ptr(tp_t) if n else None, hip.ptr(conf_t) if n else None, hip.ptr(pcls_t) if n else None, n,
             hip.ptr(tcls_t) if tcls_t.numel() else None, int(tcls_t.numel()), nc, niou, hip.ptr(grid), hip.ptr(cgrid), int(_CONF_GRID.size),
             hip.ptr(ws), ws.numel(), hip.ptr(ap), hip.ptr(prec_at), hip.ptr(rec_at), hip.ptr(nlab), hip.ptr(npred), hip.stream())
    present = np.nonzero(nlab.cpu().numpy() > 0)[0]               # np.unique(target_cls): the classes that have labels
    ap_h, p_h, r_h = ap.cpu().numpy()[present], prec_at.cpu().numpy()[present], rec_at.cpu().numpy()[present]
    f1 = 2 * p_h * r_h / (p_h + r_h + 1e-16)
    best = f1.mean(0).argmax()
    return p_h[:, best], r_h[:, best], ap_h, f1[:, best], present.astype("int32")


def calculate_eval_stats(stats, num_classes, host=False):
    """test.py:152-164: (nt, p, r, ap50, ap, f1, ap_class, mp, mr, map50, map) from the concatenated statistics; the all-zero
    tuple (with nt = zeros(1)) when there is not a single true positive at the first threshold.  The AP computation runs on the HIP
    device (ap_per_class_device); host=True selects the numpy restatement above (what the CPU tests pin to the reference fixture; also
    the form for more than MAP_MAX_CLASSES classes).  There is no silent switch between the two: without a HIP device the default raises.
    Tie rule of the device path: confidences are compared as fp32 and equal confidences keep ascending detection index, i.e. a STABLE
    descending sort; the reference's np.argsort(-conf) (quicksort) leaves tie order unspecified, so with duplicated confidences the
    numpy path may differ in the last digits while both are valid readings of test.py:38."""
    if not (len(stats) and stats[0].any()):
        return torch.zeros(1), 0.0, 0.0, [], [], 0.0, [], 0.0, 0.0, 0.0, 0.0
    if host:
        p, r, ap_all, f1, ap_class = ap_per_class(*stats)
    else:
        if not torch.cuda.is_available():
            raise RuntimeError("ryolov4_amd.lib.evaluate.calculate_eval_stats: no HIP device (pass host=True for the numpy restatement)")
        p, r, ap_all, f1, ap_class = ap_per_class_device(*stats, num_classes=num_classes)
    ap50, ap = ap_all[:, 0], ap_all.mean(1)
    nt = np.bincount(stats[3].astype(np.int64), minlength=num_classes)
    return nt, p, r, ap50, ap, f1, ap_class, p.mean(), r.mean(), ap50.mean(), ap.mean()
