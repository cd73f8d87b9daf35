"""(cases of tests/test_gpu_gemm1x1.py; not collected on its own: the file name does not match test_*.py)
The weight-stationary persistent 1x1 kernel (csrc/gemm1x1.hip, off by default: RYOLO_GEMM_WS) through the C ABI against torch's fp32
matmul on the same bf16 inputs: ragged pixel counts (last tile partial), output widths that are not a multiple of the 128-channel tile
(whole quarters of a tile masked: the store count the wave's vmcnt arithmetic relies on changes), channel slices of wider buffers
(ld > C), more waves than tiles, every epilogue (raw, BatchNorm statistics accumulated over all tiles of a wave, folded BN + activation,
accumulate) and the reduction lengths 64 ... 256.  The knob is read once per process, so the cases run in a child process with
RYOLO_GEMM_WS=2 (every eligible launch).  Tolerance: bf16 output rounding (2^-7 relative) on fp32-accumulated sums."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
assert os.environ.get("RYOLO_GEMM_WS") == "2", "run through tests/test_gpu_gemm1x1.py (the knob is read once per process)"

SHAPES = [  # M, Cin, Cout
    (3 * 25 * 25, 256, 400),       # ragged M (1875 = 29 tiles + 19 rows), 4 n tiles, last one 16 channels wide
    (5 * 31 * 31, 128, 136),       # second n tile has ONE live 8-channel chunk: three quarters store nothing
    (2 * 40 * 40, 64, 128),        # shortest K (two stages: the ring never wraps inside a tile)
    (64, 96, 8),                   # one tile, one quarter, K = 3 stages
    (17 * 1000, 256, 256),         # 266 tiles over 2 x 128 workgroups: several tiles per wave, prefetch across tiles
    (40000, 192, 128),             # K = 6 stages, 625 tiles
]


def _run(M, Cin, Cout, epi, ld_extra=0, seed=0, act=3):
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    S.check_layouts()
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(seed)
    ldA, ldC = Cin + ld_extra, Cout + ld_extra
    xfull = torch.randn(M, ldA, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, Cin, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    yfull = (torch.randn(M, ldC, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    y0 = yfull.clone()
    zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
    p = S.ConvGemmParams()
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = xfull.data_ptr(), 1, 1, M, Cin, ldA        # one "image" of M x 1 pixels: a 1x1 conv is a plain GEMM
    p.W, p.Nout, p.wtaps = w.data_ptr(), Cout, 1
    p.OH, p.OW, p.sh, p.sw = 1, M, 1, 1
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, 1, M
    p.nclasses = 1
    p.cls[0].ntaps = 1
    p.epi, p.out, p.ldC = epi, yfull.data_ptr(), ldC
    p.zeros, p.pipe = zeros.data_ptr(), 0x201
    co = torch.rand(4, Cout, device=dev) + 0.5
    co[3] -= 1.0
    p.scale, p.shift, p.act = co.data_ptr() + 2 * Cout * 4, co.data_ptr() + 3 * Cout * 4, act
    rows, kern = S.I(), S.I()
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    assert kern.value & 0xff == 2, f"not routed to the persistent kernel (kernel {kern.value:#x})"
    stats = torch.full((rows.value, 2, Cout), float("nan"), device=dev)              # every row must be WRITTEN (waves without tiles write zeros)
    p.stats = stats.data_ptr()
    hip.call("ryolo_conv_gemm", p, hip.stream())
    torch.cuda.synchronize()
    ref = xfull[:, :Cin].float() @ w.float().t()
    got = yfull[:, :Cout].float()
    if epi == S.EPI_AFFINE_ACT:
        u = ref * co[2] + co[3]
        ref = {3: u * torch.sigmoid(u), 2: torch.where(u > 0, u, 0.1 * u), 1: u * torch.tanh(torch.nn.functional.softplus(u)), 0: u}[act]
    if epi == S.EPI_ACCUM:
        ref = ref.to(torch.bfloat16).float() + y0[:, :Cout].float()
    err = (got - ref).abs()
    tol = 2.0 ** -7 * ref.abs() + 2e-2
    assert bool((err <= tol).all()), f"max err {float(err.max())} at {int(err.argmax())}"
    assert float((got - ref).norm() / ref.norm()) < 4e-3
    if ld_extra:
        assert torch.equal(yfull[:, Cout:], y0[:, Cout:]), "wrote outside its channel slice"
    if epi == S.EPI_STATS:
        assert bool(torch.isfinite(stats).all()), "a partial-statistics row was not written"
        s1, s2 = stats[:, 0].double().sum(0), stats[:, 1].double().sum(0)
        assert torch.allclose(s1, got.double().sum(0), rtol=1e-4, atol=2e-2)
        assert torch.allclose(s2, (got.double() * got.double()).sum(0), rtol=1e-4, atol=2e-2)
    return yfull, stats


@pytest.mark.parametrize("epi", [0, 1, 2, 4])
@pytest.mark.parametrize("shape", SHAPES)
def test_shapes_and_epilogues(shape, epi):
    _run(*shape, epi=epi, seed=epi)


@pytest.mark.parametrize("epi", [0, 1, 2, 4])
def test_channel_slices(epi):
    _run(3 * 25 * 25, 128, 200, epi=epi, ld_extra=56)
    _run(9000, 256, 128, epi=epi, ld_extra=8)


def test_repeatable_bits():
    """Static tile assignment + fixed-order statistics: two launches give identical bits."""
    y1, s1 = _run(17 * 1000, 256, 256, epi=1, seed=3)
    y2, s2 = _run(17 * 1000, 256, 256, epi=1, seed=3)
    assert torch.equal(y1, y2) and torch.equal(s1, s2)


def _run_s2d(NB, OH, OW, Cout, epi, ld_extra=0, seed=0):
    """The space-to-depth instantiation: data gradient of Conv2d(32, Cout, 3, stride 2, pad 1) as ONE stride-1 GEMM over the dY grid
    (ryolo_pack_s2d weights, 2 x 2 taps, depth-to-space store), against torch autograd in fp32 on the same bf16 operands."""
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    S.check_layouts()
    dev = "cuda:0"
    Cin = 32
    g = torch.Generator(device="cpu").manual_seed(seed)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).to(dev)
    wp = torch.zeros(4 * Cin, 4, Cout, dtype=torch.bfloat16, device=dev)
    hip.call("ryolo_pack_s2d", w.data_ptr(), Cout, Cin, wp.data_ptr(), hip.stream())
    M = NB * OH * OW
    ldA, ldC = Cout + ld_extra, Cin + ld_extra
    dyfull = torch.randn(M, ldA, generator=g).to(torch.bfloat16).to(dev)
    dxfull = (torch.randn(NB * 4 * OH * OW, ldC, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dx0 = dxfull.clone()
    zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
    p = S.ConvGemmParams()
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = dyfull.data_ptr(), NB, OH, OW, Cout, ldA
    p.W, p.Nout, p.wtaps = wp.data_ptr(), 4 * Cin, 4
    p.OH, p.OW, p.sh, p.sw = OH, OW, 1, 1
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 2, 2, 2 * OH, 2 * OW
    p.nclasses = 1
    tc = p.cls[0]
    tc.ntaps = 4
    for t, (da, db) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        tc.dh[t], tc.dw[t], tc.widx[t] = da, db, 2 * da + db
    p.epi, p.out, p.ldC = epi, dxfull.data_ptr(), ldC
    p.zeros, p.pipe, p.s2d_cin = zeros.data_ptr(), 0x201, Cin
    rows, kern = S.I(), S.I()
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    assert kern.value & 0xff == 2, f"not routed to the persistent kernel (kernel {kern.value:#x})"
    hip.call("ryolo_conv_gemm", p, hip.stream())
    torch.cuda.synchronize()
    x = torch.zeros(NB, Cin, 2 * OH, 2 * OW, device=dev, requires_grad=True)
    wq = w.to(torch.bfloat16).float()                                          # the packed image holds bf16-rounded weights
    y = torch.nn.functional.conv2d(x, wq, stride=2, padding=1)
    y.backward(dyfull[:, :Cout].float().view(NB, OH, OW, Cout).permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1).reshape(-1, Cin)
    got = dxfull[:, :Cin].float()
    if epi == S.EPI_ACCUM:
        ref = ref.to(torch.bfloat16).float() + dx0[:, :Cin].float()
    err = (got - ref).abs()
    tol = 2.0 ** -7 * ref.abs() + 2e-2
    assert bool((err <= tol).all()), f"max err {float(err.max())} at {int(err.argmax())}"
    assert float((got - ref).norm() / ref.norm()) < 4e-3
    if ld_extra:
        assert torch.equal(dxfull[:, Cin:], dx0[:, Cin:]), "wrote outside its channel slice"
    return dxfull


@pytest.mark.parametrize("epi", [0, 4])
@pytest.mark.parametrize("geom", [(2, 32, 32, 64), (3, 25, 19, 64), (1, 7, 5, 64), (2, 40, 40, 32), (5, 16, 48, 64)])
def test_space_to_depth_data_gradient(geom, epi):
    _run_s2d(*geom, epi=epi, seed=epi + 1)


def test_space_to_depth_channel_slices_and_repeatability():
    a = _run_s2d(2, 20, 24, 64, epi=0, ld_extra=32, seed=5)
    b = _run_s2d(2, 20, 24, 64, epi=0, ld_extra=32, seed=5)
    assert torch.equal(a, b)


def _run_pool(NB, H, W, Cin, Cout, epi, seed=0):
    """The MaxPool-gradient instantiation: out[(h, w), n] = bf16(x @ w^T) + pool_dz[(h / 2, w / 2), n] where pool_idx == (h & 1) * 2 + (w & 1)
    (then the accumulate epilogue), against the same arithmetic in torch."""
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    S.check_layouts()
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(seed)
    M = NB * H * W
    x = torch.randn(M, Cin, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, Cin, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    y = (torch.randn(M, Cout, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    y0 = y.clone()
    pidx = torch.randint(0, 4, (NB, H // 2, W // 2, Cout), generator=g, dtype=torch.uint8).to(dev)
    pdz = torch.randn(NB, H // 2, W // 2, Cout, generator=g).to(torch.bfloat16).to(dev)
    zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
    p = S.ConvGemmParams()
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = x.data_ptr(), NB, H, W, Cin, Cin
    p.W, p.Nout, p.wtaps = w.data_ptr(), Cout, 1
    p.OH, p.OW, p.sh, p.sw = H, W, 1, 1
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, H, W
    p.nclasses = 1
    p.cls[0].ntaps = 1
    p.epi, p.out, p.ldC = epi, y.data_ptr(), Cout
    p.zeros, p.pipe = zeros.data_ptr(), 0x201
    p.pool_idx, p.pool_dz, p.pool_ldi, p.pool_ld = pidx.data_ptr(), pdz.data_ptr(), Cout, Cout
    rows, kern = S.I(), S.I()
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    assert kern.value & 0xff == 2, f"not routed to the persistent kernel (kernel {kern.value:#x})"
    hip.call("ryolo_conv_gemm", p, hip.stream())
    torch.cuda.synchronize()
    ref = (x.float() @ w.float().t()).to(torch.bfloat16).float().view(NB, H, W, Cout)
    hh = torch.arange(H, device=dev).view(1, H, 1, 1)
    ww = torch.arange(W, device=dev).view(1, 1, W, 1)
    want = ((hh & 1) * 2 + (ww & 1)).to(torch.uint8)
    up_idx = pidx.repeat_interleave(2, 1).repeat_interleave(2, 2)
    up_dz = pdz.float().repeat_interleave(2, 1).repeat_interleave(2, 2)
    ref = (ref + torch.where(up_idx == want, up_dz, torch.zeros_like(up_dz))).to(torch.bfloat16).float().view(M, Cout)
    if epi == S.EPI_ACCUM:
        ref = ref + y0.float()
    got = y.float()
    err = (got - ref).abs()
    tol = 2.0 ** -6 * ref.abs() + 3e-2           # one more bf16 rounding than the plain store
    assert bool((err <= tol).all()), f"max err {float(err.max())} at {int(err.argmax())}"
    assert float((got - ref).norm() / ref.norm()) < 6e-3


@pytest.mark.parametrize("epi", [0, 4])
@pytest.mark.parametrize("geom", [(2, 20, 20, 128, 256), (3, 10, 14, 256, 128), (1, 6, 6, 64, 136), (2, 50, 50, 128, 256)])
def test_maxpool_gradient_in_the_store(geom, epi):
    _run_pool(*geom, epi=epi, seed=epi + 2)
