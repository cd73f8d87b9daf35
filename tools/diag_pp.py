"""Where the captured post_process spends its time: eager PostProcessPlan.run vs the same six launches replayed from a hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ryolov4_amd.lib.general import PostProcessPlan
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, synth_batch

dev = torch.device("cuda:0")
B, SZ = int(os.environ.get("B", 1)), 800
model = Yolo(16, CFG, "kfiou", "yolov7"); model.apply(bench.weights_init_normal); model.to(dev).eval()
imgs, _ = synth_batch(B, SZ, 16, False, seed=42); imgs = imgs.to(dev)
with torch.no_grad():
    _, inf = model(imgs, training=False)
inf0 = inf.clone()
plan = PostProcessPlan(B, inf.shape[1], inf.shape[2] - 6, dev, 0.25, 0.45)
work = inf0.clone()


def timed(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def eager():
    work.copy_(inf0)
    plan.run(work)

print("eager plan.run ms", timed(eager), "count", plan.count.tolist(), "num", plan.num.tolist())
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    eager()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g):
    eager()
print("graph plan.run ms", timed(g.replay))

# forward + decode (+ post) graphs
capp = model.capture_inference(B, SZ, post=(0.25, 0.45))
print("captured fwd+pp ms", timed(lambda: capp(imgs)))
cap = model.capture_inference(B, SZ)
print("captured fwd ms", timed(lambda: cap(imgs)))
print("captured fwd+pp again ms", timed(lambda: capp(imgs)))
_, inf_s = cap(imgs)
plan2 = PostProcessPlan(B, inf.shape[1], inf.shape[2] - 6, dev, 0.25, 0.45)
plan2.run(inf_s); torch.cuda.synchronize()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    plan2.run(inf_s)
def two():
    cap(imgs); g2.replay()
print("two graphs fwd, pp ms", timed(two))
