"""Ad-hoc robustness sweep: one training step (forward, loss, backward, fused SGD) for every ver x mode at unusual batch / image sizes;
prints the loss and whether every gradient is finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ryolov4_amd.lib.loss import ComputeCSLLoss, ComputeKFIoULoss
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, HYP, synth_batch
bad = 0
for ver in ("yolov4", "yolov5", "yolov7"):
    for mode in ("csl", "kfiou"):
        for B, S in ((1, 416), (3, 608), (5, 320), (2, 1024)):
            torch.manual_seed(0)
            m = Yolo(2, CFG, mode, ver); m.apply(bench.weights_init_normal); m.cuda().train()
            crit = (ComputeCSLLoss if mode == "csl" else ComputeKFIoULoss)(m, HYP)
            imgs, tg = synth_batch(B, S, 2, mode == "csl", seed=1, per_image=7)
            losses = []
            for step in range(2):
                loss, items = crit(m(imgs.cuda(), training=True), tg.cuda()); loss.backward()
                ok = all(torch.isfinite(p.grad).all() for p in m.parameters())
                m.runtime().sgd_step(0.01); losses.append(round(float(loss), 4))
            fin = ok and all(l == l and abs(l) < 1e6 for l in losses)
            bad += 0 if fin else 1
            print(ver, mode, B, S, losses, "OK" if fin else "FAIL", flush=True)
            del m, crit
            torch.cuda.empty_cache()
print("failures", bad)
