"""Isolated timing of the first-layer kernels at the benchmark size (64 x 800^2): statistics pass, fused forward, fused backward
(LDS-patch kernels), and the register-gather forward / two-kernel backward they replace."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ryolov4_amd import hip
from ryolov4_amd.engine import structs as S
hip.lib(); S.check_layouts()
B, H, W = int(os.environ.get("B", 64)), int(os.environ.get("SZ", 800)), int(os.environ.get("SZ", 800))
img = torch.rand(B, 3, H, W, device="cuda")
wf = torch.zeros(32, 32); wf[:, :27] = torch.randn(32, 27) * 0.2; wf = wf.to(torch.bfloat16).cuda()
M = B * H * W
z = torch.empty(M, 32, dtype=torch.bfloat16, device="cuda")
dz = (torch.randn(M, 32, device="cuda") * 0.1).to(torch.bfloat16)
rows, wsb = S.I(), S.Z(); hip.call("ryolo_stem3x3_plan", B, H, W, 32, rows, wsb)
stats = torch.zeros(rows.value + 64, 2, 32, device="cuda")
co = torch.rand(4, 32, device="cuda") + 0.5
def fwd(epi, out):
    p = S.StemParams(); p.img, p.NB, p.H, p.W = img.data_ptr(), B, H, W
    p.wf, p.Cout, p.epi, p.out, p.ldC = wf.data_ptr(), 32, epi, out.data_ptr() if out is not None else None, 32
    p.stats, p.scale, p.shift, p.act = stats.data_ptr(), co.data_ptr() + 256, co.data_ptr() + 384, 3
    return lambda: hip.call("ryolo_stem3x3_fwd", p, hip.stream())
bw = S.Z(); hip.call("ryolo_stem3x3_bwd_plan", B, H, W, 32, bw)
ws = torch.empty(bw.value // 4, device="cuda"); dW = torch.zeros(32, 3, 3, 3, device="cuda"); dg = torch.zeros(32, device="cuda"); db = torch.zeros(32, device="cuda")
q = S.StemBwdParams(); q.img, q.NB, q.H, q.W = img.data_ptr(), B, H, W
q.dz, q.lddz, q.wf, q.co, q.act, q.frozen = dz.data_ptr(), 32, wf.data_ptr(), co.data_ptr(), 3, 0
q.workspace, q.dW, q.dgamma, q.dbeta = ws.data_ptr(), dW.data_ptr(), dg.data_ptr(), db.data_ptr()
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)
res = {"stats_only_ms": t(fwd(1, None)), "fwd_stats_store_ms": t(fwd(1, z)), "fwd_fused_ms": t(fwd(5, z)), "fwd_raw_ms": t(fwd(0, z)),
       "bwd_fused_ms": t(lambda: hip.call("ryolo_stem3x3_bwd", q, hip.stream()))}
print(json.dumps(res))
