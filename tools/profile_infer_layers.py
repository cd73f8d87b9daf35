"""Per-launch table of the EVAL forward tape (folded BatchNorm + activation epilogues): shape, time, TFLOP/s, algorithmic GB/s, workgroups.
B / SZ from the environment (C5: B=8 SZ=1024; bench block: B=64 SZ=800)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ryolov4_amd.engine import structs as S
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG

dev = torch.device("cuda:0")
B, SZ = int(os.environ.get("B", 8)), int(os.environ.get("SZ", 1024))
torch.manual_seed(42)
model = Yolo(16, CFG, "kfiou", "yolov7")
model.apply(bench.weights_init_normal)
model.to(dev).eval()
imgs = torch.rand(B, 3, SZ, SZ, device=dev)
with torch.no_grad():
    for _ in range(3):
        model(imgs, False)
rt = model.runtime()
g = rt.graph(B, SZ, SZ, False)
g.static_weights = True
torch.cuda.synchronize()
best = collections.defaultdict(lambda: 1e9)
st = torch.cuda.current_stream().cuda_stream
for rep in range(5):
    evs = []
    for i, (fn, args, name) in enumerate(g.fwd):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(*args, st); e1.record(); evs.append((i, name, e0, e1))
    torch.cuda.synchronize()
    for i, name, e0, e1 in evs:
        best[(i, name)] = min(best[(i, name)], e0.elapsed_time(e1))
tot = collections.defaultdict(float)
cnt = collections.Counter()
detail = []
for (i, name), ms in best.items():
    kind, fl, by = (tuple(g.meta.get((id(g.fwd), i), (name, 0, 0))) + (0, 0))[:3]
    tot[kind] += ms
    cnt[kind] += 1
    a0 = g.fwd[i][1][0] if g.fwd[i][1] else None
    st_ = getattr(a0, "_obj", None)
    shape = ""
    if isinstance(st_, S.ConvGemmParams):
        shape = f"{st_.Cin}->{st_.Nout} taps{st_.cls[0].ntaps}x{st_.nclasses} {st_.OH}x{st_.OW} s{st_.sh} epi{st_.epi} M{st_.NB * st_.OH * st_.OW}"
    detail.append((ms, i, kind, fl, by, shape))
print(f"== eval forward tape, B={B} SZ={SZ}: totals per kernel class (ms, isolated launches) ==")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{k:38s} {v:8.3f} ms  {cnt[k]:4d} launches")
print("sum", round(sum(tot.values()), 3), "launches", len(best))
print("== launches ==")
for ms, i, kind, fl, by, shape in sorted(detail, reverse=True)[:int(os.environ.get("TOP", 120))]:
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0
    gb = by / (ms * 1e-3) / 1e9 if ms > 0 else 0
    print(f"#{i:4d} {kind:36s} {ms * 1e3:8.1f} us {tf:8.1f} TF/s {gb:8.1f} GB/s  flops {fl / 1e9:8.2f} G  bytes {by / 1e6:8.1f} MB  {shape}")
