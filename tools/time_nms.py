"""Device-time rotated NMS at the SURVEY §8(d) sizes (hipEvent, median of 50)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ryolov4_amd.lib import general
from ryolov4_amd.synth import synth_nms_boxes
dev = "cuda:0"
res = []
for n in (1000, 5000, 10000, 50000):
    for dist in ("U", "C"):
        for thr in (0.65, 0.2):
            b, s = synth_nms_boxes(n, dist, seed=0)
            tb = torch.from_numpy(b).to(dev).unsqueeze(0).contiguous()
            for mk in (None, 1500):
                for _ in range(3):
                    keep, num = general._nms_sorted_batched(tb, None, thr, True, mk)
                torch.cuda.synchronize()
                ts = []
                for _ in range(30):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); keep, num = general._nms_sorted_batched(tb, None, thr, True, mk); e1.record()
                    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
                r = dict(n=n, dist=dist, thr=thr, max_keep=mk, ms=float(np.median(ts)), kept=int(num.item()))
                print(json.dumps(r)); res.append(r)
json.dump(res, open("gpurun_out/nms_times.json", "w"))
