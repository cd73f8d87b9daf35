"""Single-layer microbenchmark of ryolo_conv_gemm (forward 3x3 / 1x1) through the C ABI — for PMC runs and A/B tests.
usage: python tools/bench_conv.py [B H Cin Cout k stride pipe reps]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ryolov4_amd import hip
from ryolov4_amd.engine import structs as S
args = [int(a, 0) for a in sys.argv[1:]] + [None] * 8
B, H, Cin, Cout, k, stride, pipe, reps = [a if a is not None else d for a, d in zip(args[:8], (8, 200, 128, 128, 3, 1, 1, 20))]
epi = int(os.environ.get("EPI", "0"))
dev = "cuda:0"
hip.lib(); S.check_layouts()
pad = (k - 1) // 2
OH = (H + 2 * pad - k) // stride + 1
x = torch.randn(B * H * H, Cin, device=dev).to(torch.bfloat16)
w = (torch.randn(Cout, k * k, Cin, device=dev) * 0.05).to(torch.bfloat16)
y = torch.empty(B * OH * OH, Cout, dtype=torch.bfloat16, device=dev)
zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
p = S.ConvGemmParams()
p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = x.data_ptr(), B, H, H, Cin, Cin
p.W, p.Nout, p.wtaps = w.data_ptr(), Cout, k * k
p.OH, p.OW, p.sh, p.sw = OH, OH, stride, stride
p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, OH, OH
p.nclasses = 1
tc = p.cls[0]; tc.ntaps = k * k
for r in range(k):
    for s in range(k):
        tc.dh[r * k + s], tc.dw[r * k + s], tc.widx[r * k + s] = r - pad, s - pad, r * k + s
p.epi, p.out, p.ldC = epi, y.data_ptr(), Cout
p.zeros, p.pipe = zeros.data_ptr(), pipe
p.a_bytes, p.w_bytes = x.numel() * 2, w.numel() * 2
co = torch.rand(4, Cout, device=dev) + 0.5                    # EPI=2: folded BatchNorm scale (row 2) / shift (row 3), SiLU
p.scale, p.shift, p.act = co.data_ptr() + 2 * Cout * 4, co.data_ptr() + 3 * Cout * 4, int(os.environ.get("ACT", "3"))
rows, kern = S.I(), S.I()
try:
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
except Exception:                                   # older library (A/B runs through RYOLO_LIB)
    hip.call("ryolo_conv_gemm_stats_rows", B * OH * OH, Cout, pipe & 0xff, rows)
stats = torch.zeros(rows.value, 2, Cout, device=dev)
p.stats = stats.data_ptr()
dbg = torch.zeros(4 * 70000, dtype=torch.int64, device=dev)
if os.environ.get('P3_TIMING') or os.environ.get('GEMM_TIMING') or os.environ.get('WS3_TIMING'): p.bias = dbg.data_ptr()
st = hip.stream()
for _ in range(3): hip.call("ryolo_conv_gemm", p, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): hip.call("ryolo_conv_gemm", p, st)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 2 * B * OH * OH * Cout * k * k * Cin
print(f"B{B} H{H} Cin{Cin} Cout{Cout} k{k} s{stride} pipe{pipe:#x} kernel{kern.value}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s   in {x.numel()*2/1e6:.0f} MB out {y.numel()*2/1e6:.0f} MB")
# reference check against torch (fp32): first two images and the last one (image borders matter for the flat-run tiles)
xr = x.float().view(B, H, H, Cin).permute(0, 3, 1, 2)
wr = w.float().view(Cout, k, k, Cin).permute(0, 3, 1, 2)
for im in sorted({0, min(1, B - 1), B - 1}) if epi != 2 else []:
    ref = torch.nn.functional.conv2d(xr[im:im + 1], wr, stride=stride, padding=pad).permute(0, 2, 3, 1).reshape(-1, Cout)
    got = y[im * OH * OH:(im + 1) * OH * OH].float()
    print("img", im, "rel err", float((got - ref).norm() / ref.norm()), "max abs", float((got - ref).abs().max()))
if epi == 1:
    yy = y.float()
    print("stats rel err", float((stats[:, 0].sum(0) - yy.sum(0)).norm() / yy.sum(0).norm()), float((stats[:, 1].sum(0) - (yy * yy).sum(0)).norm() / (yy * yy).sum(0).norm()))


if os.environ.get('GEMM_TIMING'):
    import numpy as np
    torch.cuda.synchronize()
    d = dbg.view(-1, 4).cpu().numpy(); d = d[d[:, 1] != 0]
    print("blocks", len(d), "prologue", np.mean(d[:, 0]), "loop", np.mean(d[:, 1]), "stage+store", np.mean(d[:, 2]), "stats", np.mean(d[:, 3]))
if os.environ.get('P3_TIMING'):
    import numpy as np
    torch.cuda.synchronize()
    d = dbg.view(-1, 4).cpu().numpy(); d = d[d[:, 0] != 0]
    print("blocks", len(d), "prologue", np.mean(d[:, 1] - d[:, 0]), "loop", np.mean(d[:, 2] - d[:, 1]), "epilogue", np.mean(d[:, 3] - d[:, 2]))
if os.environ.get('WS3_TIMING'):
    import numpy as np
    torch.cuda.synchronize()
    d = dbg.view(-1, 4).cpu().numpy(); d = d[d[:, 3] != 0]
    tiles = d[:, 3] & 4095
    total = d[:, 3] >> 12
    print("workgroups", len(d), "tiles/wg", tiles.mean(), "kernel cycles", total.mean(), "prologue", d[:, 0].mean(), "MFMA loop per tile", (d[:, 1] / tiles).mean(),
          "tail per tile", (d[:, 2] / tiles).mean(), "unaccounted per tile", ((total - d[:, 0] - d[:, 1] - d[:, 2]) / tiles).mean())
