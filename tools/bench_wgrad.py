"""Single-layer microbenchmark of ryolo_conv_wgrad through the C ABI (split-K kernel + deterministic reduce), with a check
against torch's fp32 conv weight gradient.  usage: python tools/bench_wgrad.py [B H Cin Cout k stride reps]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ryolov4_amd import hip
from ryolov4_amd.engine import structs as S
args = [int(a, 0) for a in sys.argv[1:]] + [None] * 7
B, H, Cin, Cout, k, stride, reps = [a if a is not None else d for a, d in zip(args[:7], (64, 100, 128, 128, 3, 1, 10))]
dev = "cuda:0"
hip.lib(); S.check_layouts()
pad = (k - 1) // 2
OH = (H + 2 * pad - k) // stride + 1
x = torch.randn(B * H * H, Cin, device=dev).to(torch.bfloat16)
dy = (torch.randn(B * OH * OH, Cout, device=dev) * 0.1).to(torch.bfloat16)
dw = torch.zeros(Cout, Cin, k * k, device=dev)
p = S.WgradParams()
p.dY, p.ldY, p.Cout, p.CoutPad = dy.data_ptr(), Cout, Cout, Cout
p.X, p.NB, p.IH, p.IW, p.Cin, p.ldX = x.data_ptr(), B, H, H, Cin, Cin
p.OH, p.OW, p.sh, p.sw, p.ntaps = OH, OH, stride, stride, k * k
for r in range(k):
    for s in range(k):
        p.dh[r * k + s], p.dw[r * k + s] = r - pad, s - pad
p.dW = dw.data_ptr()
zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
if os.environ.get('W3', '1') == '1': p.zeros = zeros.data_ptr()
sk, ws = S.I(), S.Z()
hip.call("ryolo_conv_wgrad_plan", p, sk, ws)
work = torch.zeros(ws.value + (1 << 22), dtype=torch.uint8, device=dev)     # + room for the W3_TIMING debug stamps
p.partial = work.data_ptr()
st = hip.stream()
for _ in range(2): hip.call("ryolo_conv_wgrad", p, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): hip.call("ryolo_conv_wgrad", p, st)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 2 * B * OH * OH * Cout * k * k * Cin
print(f"wgrad B{B} H{H} Cin{Cin} Cout{Cout} k{k} s{stride} splitk{sk.value}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s (kernel + reduce)")
if os.environ.get("CHECK", "1") == "1":
    dw.zero_(); hip.call("ryolo_conv_wgrad", p, st); torch.cuda.synchronize()
    nb = min(B, 4)                                    # reference on the first images only (fp32 torch conv backward)
    p.NB = nb; dw2 = torch.zeros_like(dw); p.dW = dw2.data_ptr(); hip.call("ryolo_conv_wgrad_plan", p, sk, ws); hip.call("ryolo_conv_wgrad", p, st); torch.cuda.synchronize()
    xr = x[: nb * H * H].float().view(nb, H, H, Cin).permute(0, 3, 1, 2).requires_grad_(False)
    w0 = torch.zeros(Cout, Cin, k, k, device=dev, requires_grad=True)
    y = torch.nn.functional.conv2d(xr, w0, stride=stride, padding=pad)
    y.backward(dy[: nb * OH * OH].float().view(nb, OH, OH, Cout).permute(0, 3, 1, 2))
    ref = w0.grad.reshape(Cout, Cin, k * k)
    print("rel err", float((dw2 - ref).norm() / ref.norm()))

if os.environ.get("W3_TIMING"):
    import numpy as np
    torch.cuda.synchronize()
    d = work[ws.value:].view(torch.int64).view(-1, 4).cpu().numpy()
    d = d[d[:, 3] != 0]
    print("blocks", len(d), "steps", d[:, 3].mean(), "loop per step", (d[:, 1] / d[:, 3]).mean(), "DMA wait per step (wave 0)", (d[:, 0] / d[:, 3]).mean(), "barrier wait per step", (d[:, 2] / d[:, 3]).mean())
