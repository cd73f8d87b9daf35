"""mAP-parity experiment (BASELINE north_star: "... at reference mAP parity"): the HIP path and the fp32 torch-CPU
oracle are TRAINED on the same rendered rotated-rectangle task with identical initial weights, batches and optimizer, then EVALUATED the
way test.py:167-222 does (eval forward, post_process conf 0.001 / iou 0.65, get_batch_statistics over IoU 0.5:0.95, ap_per_class).

Task: 96 x 96 images with 1-3 filled rotated rectangles on a noisy background, 2 classes told apart by colour, sizes around the
stride-8 anchors; 32 images, trained for `steps` SGD-nesterov steps (lr 0.01, momentum 0.937, train.py:156) in batches of 16, evaluated
on the same 32 images (a memorisation task: what a few hundred steps can reach).

Reported (gpurun_out/r03_map_parity.json):
  hip_trained / oracle_trained     mAP@0.5, mAP@0.5:0.95, P, R of each path's OWN training + evaluation          -> band |d mAP@0.5| stated
  cross                            the ORACLE-trained weights evaluated by the HIP path (model, post_process, NMS, matching, AP on the
                                   device): isolates inference + evaluation parity from the chaotic training trajectory -> measured 1e-4 (300 steps) / 0.025 (600 steps): 65 labels, one flipped TP = 0.015-0.03
usage: python tools/map_parity.py [steps] [ver] [mode]"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_data, ref_model, ref_ops
from ryolov4_amd.lib import evaluate as EV
from ryolov4_amd.lib.general import post_process
from ryolov4_amd.lib.loss import ComputeCSLLoss, ComputeKFIoULoss
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, HYP

S, NC, NIMG, BATCH = 96, 2, 32, 16
DEV = "cuda:0"


def weights_init_normal(m):                                      # train.py:28-33
    if isinstance(m, torch.nn.Conv2d):
        torch.nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif isinstance(m, torch.nn.BatchNorm2d):
        torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
        torch.nn.init.constant_(m.bias.data, 0.0)


def render(rs, csl):
    """One image [3, S, S] in [0, 1] and its labels [n, 7 | 187] = (0, cls, x, y, w, h, theta[, csl x 180]) normalised, in the repo's box
    convention (lib/general.py:70-104 through the oracle's xyxyxyxy2xywha: h = long side, theta in [-pi/2, pi/2))."""
    img = rs.rand(S, S, 3).astype(np.float32) * 0.25
    ys, xs = np.mgrid[0:S, 0:S].astype(np.float32) + 0.5
    rows = []
    for _ in range(rs.randint(1, 4)):
        cx, cy = rs.uniform(18, S - 18, size=2)
        long_, short = rs.uniform(22, 34), rs.uniform(10, 16)
        th = rs.uniform(-math.pi / 2, math.pi / 2)
        cls = rs.randint(0, NC)
        c, s = math.cos(th), math.sin(th)
        u, v = (xs - cx) * c + (ys - cy) * s, -(xs - cx) * s + (ys - cy) * c
        inside = (np.abs(u) <= long_ / 2) & (np.abs(v) <= short / 2)
        colour = np.array([0.9, 0.5, 0.1] if cls == 0 else [0.1, 0.5, 0.9], dtype=np.float32)
        img[inside] = colour + rs.rand(int(inside.sum()), 3).astype(np.float32) * 0.1
        hx, hy = np.array([c, s]) * long_ / 2, np.array([-s, c]) * short / 2
        ctr = np.array([cx, cy])
        poly = np.concatenate([ctr - hx - hy, ctr + hx - hy, ctr + hx + hy, ctr - hx + hy]) / S
        rows.append((cls, poly))
    polys = torch.tensor(np.stack([p for _, p in rows]), dtype=torch.float32)
    rb = ref_data.xyxyxyxy2xywha(polys)
    lab = torch.cat((torch.zeros(len(rows), 1), torch.tensor([[float(c)] for c, _ in rows]), rb), 1)
    if csl:
        g = np.stack([ref_data.gaussian_label(float(r[4]) * 180 / np.pi + 90, 180, 0, 6) for r in rb])
        lab = torch.cat((lab, torch.from_numpy(g).float()), 1)
    return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))), lab


def dataset(csl):
    rs = np.random.RandomState(1)
    imgs, labs = zip(*[render(rs, csl) for _ in range(NIMG)])
    batches = []
    for a in range(0, NIMG, BATCH):
        tg = []
        for k in range(BATCH):
            t = labs[a + k].clone()
            t[:, 0] = k
            tg.append(t)
        batches.append((torch.stack(imgs[a:a + BATCH]), torch.cat(tg)))
    return batches


def evaluate(forward, loss_fn, pp, stats_fn, batches, to_dev, host_ap):
    """test.py:167-222 with conf 0.001, iou 0.65."""
    iouv = torch.linspace(0.5, 0.95, 10)
    stats, seen = [], 0
    for imgs, targets in batches:
        imgs, targets = to_dev(imgs), to_dev(targets.clone())
        seen += len(imgs)
        with torch.no_grad():
            outputs, infer = forward(imgs)
            infer = pp(infer, conf_thres=0.001, iou_thres=0.65)
        targets[:, 2:6] *= S
        stats += stats_fn(infer, targets, iouv, iouv.numel())
    cat = [np.concatenate(x, 0) for x in list(zip(*stats))]
    nt, p, r, ap50, ap, f1, ap_class, mp, mr, map50, map_ = EV.calculate_eval_stats(cat, NC, host=host_ap)
    return dict(images=seen, labels=int(np.sum(nt)), detections=int(len(cat[1])), P=float(mp), R=float(mr), mAP50=float(map50), mAP=float(map_))


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 1))          # torch-CPU oversubscribes on the 256-thread GPU hosts (bench.py's cpu_baseline notes)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ver = sys.argv[2] if len(sys.argv) > 2 else "yolov7"
    mode = sys.argv[3] if len(sys.argv) > 3 else "kfiou"
    csl = mode == "csl"
    batches = dataset(csl)
    torch.manual_seed(42)
    orc = ref_model.Yolo(NC, CFG, mode, ver)
    orc.apply(weights_init_normal)
    sd0 = {k: v.clone() for k, v in orc.state_dict().items()}
    net = Yolo(NC, CFG, mode, ver)
    net.load_state_dict(sd0)
    net.to(DEV)
    crit = (ComputeCSLLoss if csl else ComputeKFIoULoss)(net, HYP)
    t0 = time.time()
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.937, nesterov=True)
    dbatches = [(i.to(DEV), t.to(DEV)) for i, t in batches]
    hip_curve = []
    for it in range(steps):
        imgs, tg = dbatches[it % len(dbatches)]
        loss, items = crit(net(imgs, training=True), tg)
        loss.backward()
        opt.step()
        opt.zero_grad()
        hip_curve.append(float(items["total_loss"]))
    t_hip = time.time() - t0
    t0 = time.time()
    orc.train()
    oopt = torch.optim.SGD(orc.parameters(), lr=0.01, momentum=0.937, nesterov=True)
    orc_curve = []
    for it in range(steps):
        imgs, tg = batches[it % len(batches)]
        loss, items = ref_ops.compute_loss(orc(imgs, True), tg, orc.anchors, NC, mode, HYP)
        loss.backward()
        oopt.step()
        oopt.zero_grad()
        orc_curve.append(float(items["total_loss"]))
    t_orc = time.time() - t0
    net.eval()
    orc.eval()
    res = {}
    res["hip_trained"] = evaluate(lambda x: net(x, training=False), None, post_process, EV.get_batch_statistics, batches, lambda t: t.to(DEV), False)
    res["oracle_trained"] = evaluate(lambda x: orc(x, False), None, ref_ops.post_process, ref_ops.get_batch_statistics, batches, lambda t: t, True)
    net2 = Yolo(NC, CFG, mode, ver)
    net2.load_state_dict(orc.state_dict())
    net2.to(DEV).eval()
    res["cross_oracle_weights_on_hip_path"] = evaluate(lambda x: net2(x, training=False), None, post_process, EV.get_batch_statistics, batches,
                                                       lambda t: t.to(DEV), False)
    out = dict(task=f"{NIMG} rendered {S}x{S} images, 1-3 rotated rectangles, {NC} classes; {ver} {mode}; {steps} SGD-nesterov steps lr 0.01 batch {BATCH}; "
                    "evaluated on the training images as test.py:167-222 (conf 0.001, iou 0.65)",
               loss_first_last=dict(hip=[hip_curve[0], hip_curve[-1]], oracle=[orc_curve[0], orc_curve[-1]]),
               loss_every_20=dict(hip=[round(v, 4) for v in hip_curve[::20]], oracle=[round(v, 4) for v in orc_curve[::20]]),
               seconds=dict(hip=round(t_hip, 1), oracle_cpu=round(t_orc, 1)), **res)
    out["delta_mAP50_trained"] = abs(res["hip_trained"]["mAP50"] - res["oracle_trained"]["mAP50"])
    out["delta_mAP50_cross"] = abs(res["cross_oracle_weights_on_hip_path"]["mAP50"] - res["oracle_trained"]["mAP50"])
    os.makedirs("gpurun_out", exist_ok=True)
    path = "gpurun_out/r03_map_parity.json"
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[f"{ver}_{mode}_{steps}"] = out
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "loss_every_20"}))


if __name__ == "__main__":
    main()
