"""Microbenchmark of the 32 -> 64 stride-2 layer's forward and space-to-depth data gradient through the C ABI (csrc/conv3x3s2_c32.hip against the
kernels it replaces: RYOLO_S2C32=0 / RYOLO_S2C32_DGRAD=0).  usage: python tools/bench_s2c32.py [B H reps] ; MODE=fwd|dgrad"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ryolov4_amd import hip
from ryolov4_amd.engine import structs as S

B, H, reps = [int(a) for a in sys.argv[1:4]] + [64, 800, 30][len(sys.argv) - 1:]
mode = os.environ.get("MODE", "dgrad")
dev = "cuda:0"
hip.lib()
S.check_layouts()
Cin, Cout, OH = 32, 64, H // 2
zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
p = S.ConvGemmParams()
if mode == "dgrad":
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.08)
    packed = torch.zeros(4 * Cin, 4, Cout, dtype=torch.bfloat16, device=dev)
    hip.call("ryolo_pack_s2d", w.data_ptr(), Cout, Cin, packed.data_ptr(), hip.stream())
    dy = (torch.randn(B * OH * OH, Cout, device=dev) * 0.5).to(torch.bfloat16)
    dx = torch.empty(B * H * H, Cin, dtype=torch.bfloat16, device=dev)
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = dy.data_ptr(), B, OH, OH, Cout, Cout
    p.W, p.Nout, p.wtaps = packed.data_ptr(), 4 * Cin, 4
    p.OH, p.OW, p.sh, p.sw = OH, OH, 1, 1
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 2, 2, H, H
    p.nclasses = 1
    p.cls[0].ntaps = 4
    for i in range(4):
        p.cls[0].dh[i], p.cls[0].dw[i], p.cls[0].widx[i] = i >> 1, i & 1, i
    p.epi, p.out, p.ldC = 0, dx.data_ptr(), Cin
    p.zeros, p.pipe, p.s2d_cin = zeros.data_ptr(), 0x301, Cin
    nbytes = dy.numel() * 2 + dx.numel() * 2
else:
    x = torch.randn(B * H * H, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 9, Cin, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty(B * OH * OH, Cout, dtype=torch.bfloat16, device=dev)
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = x.data_ptr(), B, H, H, Cin, Cin
    p.W, p.Nout, p.wtaps = w.data_ptr(), Cout, 9
    p.OH, p.OW, p.sh, p.sw = OH, OH, 2, 2
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, OH, OH
    p.nclasses = 1
    p.cls[0].ntaps = 9
    for i in range(9):
        p.cls[0].dh[i], p.cls[0].dw[i], p.cls[0].widx[i] = i // 3 - 1, i % 3 - 1, i
    p.epi, p.out, p.ldC = 1, y.data_ptr(), Cout
    p.zeros, p.pipe = zeros.data_ptr(), 0x301
    nbytes = x.numel() * 2 + y.numel() * 2
rows, kern = S.I(), S.I()
hip.call("ryolo_conv_gemm_plan", p, rows, kern)
stats = torch.zeros(rows.value + 64, 2, 128, device=dev)
p.stats = stats.data_ptr()
st = hip.stream()
for _ in range(3):
    hip.call("ryolo_conv_gemm", p, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    hip.call("ryolo_conv_gemm", p, st)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 2 * B * OH * OH * Cout * 9 * Cin
print(f"{mode} B{B} H{H} kernel {kern.value:#x}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  {nbytes / us / 1e3:7.1f} GB/s of algorithmic bytes")
