"""Per-launch table of the conv kernels of one training step (HIP events): shape, time, TFLOP/s, algorithmic GB/s."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ryolov4_amd.lib.loss import ComputeKFIoULoss
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, HYP, synth_batch
from ryolov4_amd.engine import structs as S
import ctypes as C
dev = torch.device("cuda:0")
B, SZ = int(os.environ.get("B", 8)), int(os.environ.get("SZ", 800))
model = Yolo(16, CFG, "kfiou", "yolov7"); model.apply(bench.weights_init_normal); model.to(dev).train()
rt = model.runtime(); crit = ComputeKFIoULoss(model, HYP)
imgs, tg = synth_batch(B, SZ, 16, False, seed=42); imgs, tg = imgs.to(dev), tg.to(dev)
def step():
    outs = model(imgs, training=True); loss, _ = crit(outs, tg, sync_items=False); loss.backward(); rt.sgd_step(0.01)
for _ in range(3): step()
g = rt.graph(B, SZ, SZ, True)
torch.cuda.synchronize()
rows = []
for tape_name, tape in (("fwd", g.fwd), ("bwd", g.bwd)):
    pass
# time every launch of both tapes individually (events), 3 repetitions, take the min
import collections
best = collections.defaultdict(lambda: 1e9)
for rep in range(3):
    model(imgs, training=True)   # refresh forward state
    for tname, tape in (("fwd", g.fwd), ("bwd", g.bwd)):
        st = torch.cuda.current_stream().cuda_stream
        evs = []
        for i, (fn, args, name) in enumerate(tape):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(*args, st); e1.record(); evs.append((i, name, e0, e1))
        torch.cuda.synchronize()
        for i, name, e0, e1 in evs:
            best[(tname, i, name)] = min(best[(tname, i, name)], e0.elapsed_time(e1))
tot = collections.defaultdict(float)
detail = []
for (tname, i, name), ms in best.items():
    tape = g.fwd if tname == "fwd" else g.bwd
    kind, fl, by = (tuple(g.meta.get((id(tape), i), (name, 0, 0))) + (0, 0))[:3]
    tot[(tname, kind)] += ms
    shape = ""
    a0 = tape[i][1][0] if tape[i][1] else None
    st_ = getattr(a0, "_obj", None)
    if isinstance(st_, S.ConvGemmParams):
        shape = f"{st_.Cin}->{st_.Nout} taps{st_.cls[0].ntaps}x{st_.nclasses} {st_.OH}x{st_.OW} s{st_.sh} epi{st_.epi}"
    elif isinstance(st_, S.WgradParams):
        shape = f"{st_.Cin}->{st_.Cout} taps{st_.ntaps} {st_.OH}x{st_.OW} s{st_.sh}"
    elif isinstance(st_, S.BnActParams):
        shape = f"M{st_.M} C{st_.C}"
    detail.append((ms, tname, i, kind, fl, by, shape))
print("== totals per kernel class (ms per step, isolated launches) ==")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{k[0]:4s} {k[1]:34s} {v:8.3f} ms")
print("sum", sum(tot.values()))
print("== top launches ==")
# recover shapes from the kept structs: walk keep list in order of creation for gemm/wgrad structs
for ms, tname, i, kind, fl, by, shape in sorted(detail, reverse=True)[:int(os.environ.get("TOP", 70))]:
    print(f"{tname} #{i:4d} {kind:34s} {ms*1e3:9.1f} us  {fl/ms/1e9 if fl else 0:8.1f} TF/s  {by/ms/1e6 if by else 0:8.1f} GB/s  flops {fl/1e9:8.2f} G  bytes {by/1e6:8.1f} MB  {shape}")
json.dump(detail, open("gpurun_out/layers.json", "w"))
