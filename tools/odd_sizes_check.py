"""Ad-hoc robustness sweep: eval-mode whole-network forward against the fp32 oracle at odd batch sizes / image sizes, with the
3x3 patch and ring kernels forced on (RYOLO_GEMM_PIPE=0x601 RYOLO_W3_FORCE=1) or off.  Prints rel-L2 per head map."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_model
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, fill_state

def rel(a, b): return float((a - b).norm() / (b.norm() + 1e-12))
worst = 0.0
for ver, mode, B, S in [("yolov7", "kfiou", 3, 96), ("yolov7", "kfiou", 5, 160), ("yolov4", "csl", 3, 96), ("yolov5", "kfiou", 2, 224), ("yolov7", "csl", 1, 320)]:
    net = Yolo(2, CFG, mode, ver)
    sd = fill_state(net.state_dict())
    net.load_state_dict(sd, strict=True)
    net.cuda().eval()
    orc = ref_model.Yolo(2, CFG, mode, ver)
    orc.load_state_dict(sd, strict=True)
    orc.eval()
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(B * S))
    na = 3 if mode == "csl" else 18
    with torch.no_grad():
        outs, _ = net(x.cuda(), training=False)
        hm = orc.head_maps(x)
    errs = []
    for a, b in zip(outs, hm):
        Bb, _, gs, _ = b.shape
        errs.append(rel(a.cpu(), b.view(Bb, na, -1, gs, gs).permute(0, 1, 3, 4, 2)))
    worst = max(worst, max(errs))
    print(ver, mode, B, S, ["%.2e" % e for e in errs], flush=True)
print("worst", worst, "OK" if worst < 1e-2 else "FAIL")
