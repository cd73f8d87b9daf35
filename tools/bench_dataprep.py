"""Data-side kernels (csrc/dataprep.hip) at the training batch shape: HIP-event timings and algorithmic GB/s.
to_tensor: 3 B read + 12 B written per pixel; encode_labels: 40 B read + 28 / 748 B written per target (kfiou / csl)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ryolov4_amd.datasets.base_dataset import finalize_batch
from ryolov4_amd.lib.plot import detections_to_polys
from ryolov4_amd import hip
from tests.golden.make_golden_data import synth_polys

dev = "cuda:0"
B, S, per = int(os.environ.get("B", 64)), int(os.environ.get("SZ", 800)), 64
g = np.random.default_rng(0)
imgs = torch.from_numpy(g.integers(0, 256, (B, S, S, 3), dtype=np.uint8)).to(dev)
t = np.zeros((B * per, 10), np.float32)
t[:, 0] = np.repeat(np.arange(B), per); t[:, 1] = g.integers(0, 16, B * per); t[:, 2:] = synth_polys(g, B * per, S)
tg = torch.from_numpy(t).to(dev)
flags = torch.from_numpy(g.integers(0, 4, B).astype(np.uint8)).to(dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = torch.empty((B, 3, S, S), dtype=torch.float32, device=dev)
ms = timed(lambda: hip.call("ryolo_to_tensor", imgs.data_ptr(), B, S, S, flags.data_ptr(), out.data_ptr(), hip.stream()))
res = {"batch": B, "size": S, "to_tensor_ms": round(ms, 4), "to_tensor_GBps": round(B * S * S * 15 / ms / 1e6, 1)}
for csl in (False, True):
    small = torch.zeros((B, 8, 8, 3), dtype=torch.uint8, device=dev)     # labels only: H = W = 8 would filter, so scale the polygons
    tg8 = tg.clone(); tg8[:, 2:] *= 8.0 / S
    ms = timed(lambda: finalize_batch(small, tg8, flags, csl))
    res["finalize_labels_%s_ms_incl_host_count_read" % ("csl" if csl else "kfiou")] = round(ms, 4)
dets = [torch.rand((1500, 7), device=dev) * 100 + 50 for _ in range(B)]
ms = timed(lambda: detections_to_polys(dets, S, [(1024, 1024)] * B))
res["detections_to_polys_ms_96k_boxes"] = round(ms, 4)
print(json.dumps(res))
