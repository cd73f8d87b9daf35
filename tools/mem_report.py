"""What a training plan holds in HBM: the arena, and every other tensor the plan keeps (statistics partials, pooling indices, head
staging, split-K workspace), largest first.  B / SZ from the environment."""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG

dev = torch.device("cuda:0")
B, SZ = int(os.environ.get("B", 64)), int(os.environ.get("SZ", 800))
m = Yolo(16, CFG, "kfiou", "yolov7")
m.apply(bench.weights_init_normal)
m.to(dev)
rt = m.runtime(dev)
torch.cuda.synchronize()
base = torch.cuda.memory_allocated()
g = rt.graph(B, SZ, SZ, True)
gc.collect()
torch.cuda.synchronize()
print("model + runtime %.2f GB, plan %.2f GB (arena %.2f GB of %.2f GB one-tensor-each), peak during planning %.2f GB" % (
    base / 1e9, (torch.cuda.memory_allocated() - base) / 1e9, g.layout.total / 1e9 if g.layout else 0, g.layout.sum_bytes / 1e9 if g.layout else 0,
    torch.cuda.max_memory_allocated() / 1e9))
seen, rows = set(), []
for t in g.keep + [g.img] + [h[k] for h in g.heads for k in h if isinstance(h[k], torch.Tensor)]:
    if isinstance(t, torch.Tensor) and t.untyped_storage().data_ptr() not in seen:
        seen.add(t.untyped_storage().data_ptr())
        rows.append((t.untyped_storage().nbytes(), tuple(t.shape), str(t.dtype)))
rows.sort(reverse=True)
print("other plan tensors: %d, %.2f GB" % (len(rows), sum(r[0] for r in rows) / 1e9))
for r in rows[:25]:
    print("  %8.1f MB  %s %s" % (r[0] / 1e6, r[1], r[2]))
