"""Whole training step (forward, loss, backward on two streams, fused SGD) replayed from ONE hipGraph vs enqueued launch by launch.
B from the environment (default 8: the latency-bound regime, ~640 launches of 5-50 us per step)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ryolov4_amd.lib.loss import ComputeKFIoULoss
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, HYP, synth_batch

dev = torch.device("cuda:0")
B, SZ, K = int(os.environ.get("B", 8)), 800, int(os.environ.get("K", 40))
torch.manual_seed(42)
m = Yolo(16, CFG, "kfiou", "yolov7")
m.apply(bench.weights_init_normal)
m.to(dev)
rt = m.runtime(dev)
crit = ComputeKFIoULoss(m, HYP)
imgs, tg = synth_batch(B, SZ, 16, False, seed=42, per_image=64)
imgs, tg = imgs.to(dev), tg.to(dev)


def step():
    outs = m(imgs, training=True)
    loss, _ = crit(outs, tg, sync_items=False)
    loss.backward()
    rt.sgd_step(0.001, 0.937, zero_grad=True)
    return loss


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    step()
print("eager   ms/step", round(timed(step, K), 3), flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = step()
except Exception:
    import traceback
    traceback.print_exc()
    print("capture FAILED", flush=True)
    sys.exit(1)
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
print("loss after replay", float(loss), flush=True)
print("graph   ms/step", round(timed(g.replay, K), 3), flush=True)
print("eager   ms/step", round(timed(step, K), 3), flush=True)
