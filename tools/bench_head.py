"""Isolated timing of the detection-head forward at the benchmark shapes (yolov7 kfiou nc=16, batch 64 @ 800^2): row-major GEMM + finish pass
against the GEMM that writes the final layout (ConvGemmParams.head_attrs).  usage (GPU box): python tools/bench_head.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ryolov4_amd import hip                       # noqa: E402
from ryolov4_amd.engine import structs as S       # noqa: E402

DEV = "cuda:0"


def params(x, w, bias, B, gs, C, Cin):
    zeros = torch.zeros(256, dtype=torch.uint8, device=DEV)
    p = S.ConvGemmParams()
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = x.data_ptr(), B, gs, gs, Cin, Cin
    p.W, p.Nout, p.wtaps = w.data_ptr(), C, 1
    p.OH, p.OW, p.sh, p.sw = gs, gs, 1, 1
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, gs, gs
    p.nclasses = 1
    p.cls[0].ntaps = 1
    p.epi, p.bias = S.EPI_F32_BIAS, bias.data_ptr()
    p.zeros, p.pipe = zeros.data_ptr(), 0x301
    p._keep = zeros
    return p


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    S.check_layouts()
    B, na, attrs, och = 64, 18, 22, 5
    C = na * attrs
    ldp = (C + 31) // 32 * 32
    for gs, Cin in ((100, 256), (50, 512), (25, 1024)):
        M = B * gs * gs
        x = torch.randn(M, Cin, device=DEV).to(torch.bfloat16)
        w = (torch.randn(C, Cin, device=DEV) * 0.05).to(torch.bfloat16)
        bias, mul = torch.randn(C, device=DEV), 1 + 0.1 * torch.randn(C, device=DEV)
        out = torch.empty(B, na, gs, gs, attrs, device=DEV)
        pre = torch.empty(M, ldp, device=DEV)
        xobj = torch.empty(B, na, gs, gs, device=DEV)
        p0 = params(x, w, bias, B, gs, C, Cin)
        p0.out, p0.ldC = pre.data_ptr(), ldp
        t_gemm = timeit(lambda: hip.call("ryolo_conv_gemm", p0, hip.stream()))
        t_fin = timeit(lambda: hip.call("ryolo_head_finish_fwd", pre.data_ptr(), ldp, mul.data_ptr(), B, gs, na, attrs, out.data_ptr(), hip.stream()))
        res = {}
        for tag, flag in (("fused", 0),):
            p1 = params(x, w, bias, B, gs, C, Cin)
            p1.out, p1.ldC, p1.head_attrs, p1.head_och, p1.scale, p1.stats = out.data_ptr(), attrs, attrs, och | flag, mul.data_ptr(), xobj.data_ptr()
            res[tag] = timeit(lambda: hip.call("ryolo_conv_gemm", p1, hip.stream()))
        gb = (M * Cin * 2 + M * C * 4) / 1e9
        print(f"gs {gs:3d} Cin {Cin:4d}: row-major GEMM {t_gemm:7.1f} us + finish {t_fin:7.1f} us = {t_gemm + t_fin:7.1f};  fused {res['fused']:7.1f} us "
              f"({gb / res['fused'] * 1e3:.2f} TB/s of x + out)")


if __name__ == "__main__":
    main()
