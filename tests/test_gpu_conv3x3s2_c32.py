"""The streaming 3x3 stride-2 forward for 32 input channels (csrc/conv3x3s2_c32.hip, r06; the second layer of the yolov4 / yolov7 backbones:
model/backbone.py Conv(32, 64, 3, 2), model/utils.py:13-23) through the C ABI against a float64 convolution of the SAME bf16 operands.

Cases: even and odd maps (right / bottom padding taps only exist on odd ones), widths that are not a multiple of the 32-pixel column block (masked lanes),
fewer tiles than waves, Cout = 64 / 48 / 8, channel slices of wider buffers on both sides (ld > C), all three epilogues — raw, raw + BatchNorm batch
statistics (column sums of the STORED bf16 values, one row per workgroup, every row written), folded BatchNorm + activation (SiLU, Mish, leaky).
Tolerance: the stored value is the bf16 rounding of an fp32-accumulated sum of 288 products — |got - ref| <= 2^-8 |ref| + 1e-3 element-wise; the
statistics are compared with the sums of the kernel's own stored output (exact summands, fp32 accumulation order differs: 1e-4 relative).
Dispatch is asserted (kernel family 5) and the generic kernel is run on the same case behind RYOLO_S2C32=0 semantics by the network-level suites."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(B, H, W, Cout, epi=0, ldx_extra=0, ldc_extra=0, act=3, seed=0, expect=5):
    import torch.nn.functional as F
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    S.check_layouts()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(seed)
    Cin = 32
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    ldA, ldC = Cin + ldx_extra, Cout + ldc_extra
    x = torch.randn(B, H, W, ldA, generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, 9, Cin, generator=g) * 0.08).to(torch.bfloat16)            # packed forward image Wf[Cout][tap][Cin]
    out0 = (torch.randn(B, OH, OW, ldC, generator=g)).to(torch.bfloat16)
    xd, wd, od = x.to(dev), w.to(dev), out0.to(dev)
    zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
    p = S.ConvGemmParams()
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = xd.data_ptr(), B, H, W, Cin, ldA
    p.W, p.Nout, p.wtaps = wd.data_ptr(), Cout, 9
    p.OH, p.OW, p.sh, p.sw = OH, OW, 2, 2
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, OH, OW
    p.nclasses = 1
    p.cls[0].ntaps = 9
    for i in range(9):
        p.cls[0].dh[i], p.cls[0].dw[i], p.cls[0].widx[i] = i // 3 - 1, i % 3 - 1, i
    p.epi, p.out, p.ldC = epi, od.data_ptr(), ldC
    p.zeros, p.pipe = zeros.data_ptr(), 0x701          # 0x400: also on grids below the size gate
    co = torch.rand(4, Cout, generator=g) + 0.5
    co[3] -= 1.0
    cod = co.to(dev)
    p.scale, p.shift, p.act = cod.data_ptr() + 2 * Cout * 4, cod.data_ptr() + 3 * Cout * 4, act
    rows, kern = S.I(), S.I()
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    assert kern.value & 0xff == expect, f"dispatch picked kernel {kern.value:#x}"
    stats = torch.full((rows.value, 2, Cout), float("nan"), device=dev)
    p.stats = stats.data_ptr()
    hip.call("ryolo_conv_gemm", p, hip.stream())
    torch.cuda.synchronize()
    got_full = od.cpu()
    got = got_full[..., :Cout].double()
    # float64 reference on the same bf16 operands
    xr = x[..., :Cin].double().permute(0, 3, 1, 2)
    wr = w.double().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xr, wr, stride=2, padding=1).permute(0, 2, 3, 1)
    if epi == 2:
        u = ref * co[2].double() + co[3].double()
        if act == 3:
            ref = u * torch.sigmoid(u)
        elif act == 1:
            ref = u * torch.tanh(F.softplus(u))
        elif act == 2:
            ref = torch.where(u > 0, u, 0.1 * u)
        else:
            ref = u
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 1e-3
    assert bool((err <= tol).all()), f"max abs err {float(err.max()):.3e} at ref {float(ref.flatten()[err.argmax()]):.3e}"
    if ldc_extra:
        assert torch.equal(got_full[..., Cout:], out0[..., Cout:]), "columns outside the slice were touched"
    if epi == 1:
        assert bool(torch.isfinite(stats).all()), "a statistics row was not written"
        s = stats.double().sum(0).cpu()
        assert torch.allclose(s[0], got.sum((0, 1, 2)), rtol=1e-4, atol=1e-2)
        assert torch.allclose(s[1], (got * got).sum((0, 1, 2)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 96, 200), (3, 65, 47), (1, 34, 70), (5, 16, 16), (1, 400, 400)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_streaming_stride2_forward(B, H, W, epi):
    _run(B, H, W, 64, epi=epi, seed=B + H + epi)


def test_streaming_stride2_forward_narrow_outputs_and_slices():
    _run(2, 64, 64, 48, epi=1, seed=3)
    _run(2, 50, 70, 8, epi=0, seed=4)
    _run(2, 64, 64, 64, epi=1, ldx_extra=32, ldc_extra=64, seed=5)               # slices of concat buffers on both sides
    _run(1, 66, 130, 64, epi=0, ldx_extra=8, ldc_extra=24, seed=6)


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_streaming_stride2_forward_activations(act):
    _run(2, 48, 80, 64, epi=2, act=act, seed=7 + act)


def test_other_shapes_stay_on_the_other_kernels():
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    p = S.ConvGemmParams()
    z = torch.zeros(64, device="cuda:0")
    p.A = p.W = p.out = p.zeros = z.data_ptr()
    p.NB, p.IH, p.IW, p.Cin, p.ldA, p.Nout, p.wtaps, p.OH, p.OW, p.sh, p.sw = 2, 64, 64, 64, 64, 128, 9, 32, 32, 2, 2       # Cin = 64
    p.oh_mul, p.ow_mul, p.OHf, p.OWf, p.nclasses, p.ldC, p.pipe = 1, 1, 32, 32, 1, 128, 0x301
    p.cls[0].ntaps = 9
    for i in range(9):
        p.cls[0].dh[i], p.cls[0].dw[i], p.cls[0].widx[i] = i // 3 - 1, i % 3 - 1, i
    rows, kern = S.I(), S.I()
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    assert kern.value & 0xff != 5


def test_small_grids_stay_on_the_generic_kernel_by_default():
    """Below 8 192 wave tiles (about four per wave) the persistent kernels do not pay for their weight tile: one 800 x 800 image stays generic."""
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    p = S.ConvGemmParams()
    z = torch.zeros(4096, device="cuda:0")
    p.A = p.W = p.out = p.zeros = z.data_ptr()
    p.NB, p.IH, p.IW, p.Cin, p.ldA, p.Nout, p.wtaps, p.OH, p.OW, p.sh, p.sw = 1, 800, 800, 32, 32, 64, 9, 400, 400, 2, 2
    p.oh_mul, p.ow_mul, p.OHf, p.OWf, p.nclasses, p.ldC, p.pipe = 1, 1, 400, 400, 1, 64, 0x301
    p.cls[0].ntaps = 9
    for i in range(9):
        p.cls[0].dh[i], p.cls[0].dw[i], p.cls[0].widx[i] = i // 3 - 1, i % 3 - 1, i
    rows, kern = S.I(), S.I()
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    assert kern.value & 0xff == 0
    p.NB = 8
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    assert kern.value & 0xff == 5


# ---- the data gradient of the same layer in its space-to-depth form (conv3x3s2_c32_dgrad_kernel; dispatch code 6) ---------------------------------
def _run_dgrad(B, H, W, ldy_extra=0, ldx_extra=0, seed=0, expect=6):
    import torch.nn.functional as F
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    S.check_layouts()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(seed)
    Cin, Cout = 32, 64
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    ldY, ldX = Cout + ldy_extra, Cin + ldx_extra
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.08).to(torch.bfloat16).float()          # bf16-representable fp32 master weights
    dy = (torch.randn(B, OH, OW, ldY, generator=g) * 0.5).to(torch.bfloat16)
    dx0 = torch.randn(B, H, W, ldX, generator=g).to(torch.bfloat16)
    wd, dyd, dxd = w.to(dev), dy.to(dev), dx0.to(dev)
    packed = torch.zeros(4 * Cin, 4, Cout, dtype=torch.bfloat16, device=dev)
    hip.call("ryolo_pack_s2d", wd.data_ptr(), Cout, Cin, packed.data_ptr(), hip.stream())
    zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
    p = S.ConvGemmParams()
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = dyd.data_ptr(), B, OH, OW, Cout, ldY
    p.W, p.Nout, p.wtaps = packed.data_ptr(), 4 * Cin, 4
    p.OH, p.OW, p.sh, p.sw = OH, OW, 1, 1
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 2, 2, H, W
    p.nclasses = 1
    p.cls[0].ntaps = 4
    for i in range(4):
        p.cls[0].dh[i], p.cls[0].dw[i], p.cls[0].widx[i] = i >> 1, i & 1, i
    p.epi, p.out, p.ldC = 0, dxd.data_ptr(), ldX
    p.zeros, p.pipe, p.s2d_cin = zeros.data_ptr(), 0x701, Cin
    rows, kern = S.I(), S.I()
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    assert kern.value & 0xff == expect, f"dispatch picked kernel {kern.value:#x}"
    hip.call("ryolo_conv_gemm", p, hip.stream())
    torch.cuda.synchronize()
    got_full = dxd.cpu()
    got = got_full[..., :Cin].double()
    x = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w.double(), stride=2, padding=1)
    (y * dy[..., :Cout].double().permute(0, 3, 1, 2)).sum().backward()
    ref = x.grad.permute(0, 2, 3, 1)
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 1e-3
    assert bool((err <= tol).all()), f"max abs err {float(err.max()):.3e}"
    if ldx_extra:
        assert torch.equal(got_full[..., Cin:], dx0[..., Cin:]), "columns outside the slice were touched"


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 96, 200), (3, 66, 46), (1, 34, 70), (5, 16, 16), (1, 400, 400), (3, 65, 47)])
def test_streaming_stride2_dgrad(B, H, W):       # (the odd map: the last input row / column has no 2a + 1 partner — only this kernel takes it)
    _run_dgrad(B, H, W, seed=B + H)


def test_streaming_stride2_dgrad_slices():
    _run_dgrad(2, 64, 64, ldy_extra=64, ldx_extra=32, seed=11)
    _run_dgrad(1, 66, 130, ldy_extra=8, ldx_extra=24, seed=12)
