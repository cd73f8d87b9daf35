"""(cases of tests/test_gpu_gemm256.py; not collected on its own)  The 256-wide pointwise GEMM (csrc/gemm256.hip) through the C ABI against torch's
fp32 matmul on the same bf16 inputs: ragged pixel counts (last 256-pixel tile partial), output widths that are not a multiple of the channel tile
(zero-page weight rows, masked column chunks), both tile widths (<= 128 columns: 256 x 128), channel slices of wider buffers, K = 64 ... 2048,
every epilogue it takes (raw, BatchNorm statistics, folded BatchNorm + activation, accumulate).  RYOLO_GEMM_256=2 puts it on every eligible launch (by default: Cin >= 512 and
grids of >= 600 tiles), read once per process — hence the child process.  Tolerance: bf16 output rounding (2^-7 relative) on fp32-accumulated sums."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
FORCED = os.environ.get("RYOLO_GEMM_256") == "2"

SHAPES = [  # M, Cin, Cout
    (3 * 25 * 25, 512, 400),       # ragged M (1875 = 7 tiles + 83 rows), 2 channel tiles, the last one 144 wide
    (5 * 31 * 31, 128, 136),       # K = 2 steps; 256 x 256 tile with 136 live columns
    (2 * 40 * 40, 64, 128),        # ONE K step (no second stage is ever requested); 256 x 128 tiles
    (300, 1024, 128),              # two pixel tiles, 16 K steps
    (17 * 1000, 256, 256),
    (40000, 2048, 512),            # the 2048 -> 512 layer of the SPPCSPC block at 25^2, batch 64
]


def _run(M, Cin, Cout, epi, ld_extra=0, seed=0, expect=4, act=3):
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
    co = torch.rand(4, Cout, generator=g).to(dev) + 0.5           # EPI_AFFINE_ACT: folded BatchNorm scale (row 2) / shift (row 3)
    co[3] -= 1.0
    p.scale, p.shift, p.act = co.data_ptr() + 2 * Cout * 4, co.data_ptr() + 3 * Cout * 4, act
    rows, kern = S.I(), S.I()
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    assert kern.value & 0xff == expect, f"routed to kernel family {kern.value & 0xff}, expected {expect} ({kern.value:#x})"
    if expect == 4:
        assert rows.value == (M + 255) // 256 and (kern.value >> 16) & 15 == (4 if Cout <= 128 else 8)
    stats = torch.full((rows.value, 2, Cout), float("nan"), device=dev)
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


@pytest.mark.skipif(not FORCED, reason="run through tests/test_gpu_gemm256.py (RYOLO_GEMM_256=2)")
@pytest.mark.parametrize("epi", [0, 1, 2, 4])
@pytest.mark.parametrize("shape", SHAPES)
def test_shapes_and_epilogues(shape, epi):
    _run(*shape, epi=epi, seed=epi)


@pytest.mark.skipif(not FORCED, reason="run through tests/test_gpu_gemm256.py (RYOLO_GEMM_256=2)")
@pytest.mark.parametrize("epi", [0, 1, 2, 4])
def test_channel_slices(epi):
    _run(3 * 25 * 25, 128, 200, epi=epi, ld_extra=56)
    _run(9000, 512, 128, epi=epi, ld_extra=8)


@pytest.mark.skipif(not FORCED, reason="run through tests/test_gpu_gemm256.py (RYOLO_GEMM_256=2)")
@pytest.mark.parametrize("act", [0, 1, 2])
def test_inference_epilogue_activations(act):
    """EPI_AFFINE_ACT (r06: the eval tape's long-K pointwise layers): folded BatchNorm + every activation on both tile widths."""
    _run(3 * 25 * 25, 512, 400, epi=2, act=act, seed=30 + act)
    _run(300, 1024, 128, epi=2, act=act, ld_extra=8, seed=33 + act)


@pytest.mark.skipif(not FORCED, reason="run through tests/test_gpu_gemm256.py (RYOLO_GEMM_256=2)")
def test_repeatable_bits_and_ineligible_shapes():
    y1, s1 = _run(17 * 1000, 512, 256, epi=1, seed=3)
    y2, s2 = _run(17 * 1000, 512, 256, epi=1, seed=3)
    assert torch.equal(y1, y2) and torch.equal(s1, s2)
    _run(4000, 96, 256, epi=0, expect=0)                # K not a multiple of 64: generic kernel
    _run(4000, 512, 64, epi=0, expect=0)                # fewer than 128 output columns
