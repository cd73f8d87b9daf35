"""Evidence that the whole training path is numerically sound at the bench size: SGD (nesterov 0.937, lr 0.01, train.py:156) on ONE fixed
synthetic batch (yolov7, 800x800, batch 64, N(0, 0.02) init) must overfit it — the loss falls monotonically towards its floor.
Writes the curve (every 10th step) to gpurun_out/overfit_<mode>.json."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from ryolov4_amd.lib.loss import ComputeCSLLoss, ComputeKFIoULoss
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, HYP, synth_batch

mode = sys.argv[1] if len(sys.argv) > 1 else "kfiou"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
torch.manual_seed(42)
m = Yolo(16, CFG, mode, "yolov7"); m.apply(bench.weights_init_normal); m.to(dev).train()
rt = m.runtime(); crit = (ComputeCSLLoss if mode == "csl" else ComputeKFIoULoss)(m, HYP)
imgs, tg = synth_batch(64, 800, 16, mode == "csl", seed=42); imgs, tg = imgs.to(dev), tg.to(dev)
curve = []
for i in range(steps):
    loss, items = crit(m(imgs, training=True), tg, sync_items=(i % 10 == 0 or i == steps - 1))
    loss.backward(); rt.sgd_step(0.01)
    if i % 10 == 0 or i == steps - 1:
        curve.append({"step": i, **{k: round(float(v), 4) for k, v in items.items()}})
print(json.dumps({"mode": mode, "steps": steps, "finite": bool(torch.isfinite(rt.flat).all()), "curve": curve}))
