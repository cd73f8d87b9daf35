"""Split-K slab traffic of the weight gradients of one training step: per layer splitk, |dW|, slab bytes (plan only, no launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ryolov4_amd import hip
from ryolov4_amd.engine import structs as S
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG
B, SZ = int(os.environ.get("B", 64)), int(os.environ.get("SZ", 800))
m = Yolo(16, CFG, "kfiou", "yolov7").to("cuda:0").train()
g = m.runtime().graph(B, SZ, SZ, True)
tot = 0
rows = []
for p in g._wgrads:
    sk, need, k = S.I(), S.Z(), S.I()
    hip.call("ryolo_conv_wgrad_plan", p, sk, need)
    hip.call("ryolo_conv_wgrad_kernel", p, k)
    w = p.Cout * p.Cin * p.ntaps
    rows.append((need.value, sk.value, p.Cout, p.Cin, p.ntaps, p.OH, k.value))
    tot += need.value
print("layers", len(rows), "total slab bytes %.1f MB" % (tot / 1e6), "max %.1f MB" % (max(r[0] for r in rows) / 1e6))
for r in sorted(rows, reverse=True)[:25]:
    print("slab %.1f MB splitk %d Cout %d Cin %d taps %d OH %d ring %d" % (r[0] / 1e6, *r[1:]))
import collections
h = collections.Counter(r[1] for r in rows)
print(sorted(h.items()))
