"""The 3x3 stride-1 halo-patch kernel (csrc/conv3x3.hip) through the C ABI against torch's fp32 conv2d on the same bf16
inputs: both tile modes (2-D tiles, flat runs), image borders / wrap-around masks, partial last tiles, Cout not a multiple
of the N tile, every epilogue (raw, BatchNorm statistics, folded BN + activation, accumulate), forward and mirrored-tap
(data-gradient) weight order.  Tolerance: bf16 output rounding (rel 2^-8) on fp32-accumulated sums."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(B, H, W, Cin, Cout, epi=0, ld_extra=0, mirrored=False, seed=0, act=3, expect_kernel=1, pipe_extra=0, expect_cols=None):
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    S.check_layouts()
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(seed)
    ldA, ldC = Cin + ld_extra, Cout + ld_extra
    xfull = torch.randn(B * H * W, ldA, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, 9, Cin, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    yfull = (torch.randn(B * H * W, ldC, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    y0 = yfull.clone()
    zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
    p = S.ConvGemmParams()
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = xfull.data_ptr(), B, H, W, Cin, ldA
    p.W, p.Nout, p.wtaps = w.data_ptr(), Cout, 9
    p.OH, p.OW, p.sh, p.sw = H, W, 1, 1
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, H, W
    p.nclasses = 1
    tc = p.cls[0]
    tc.ntaps = 9
    order = list(range(9))
    if mirrored:
        order = order[::-1]                       # tap t reads offset (dh, dw) but weight slot 8 - t, listed in reverse order
    for t, k in enumerate(order):
        r, s = divmod(k, 3)
        tc.dh[t], tc.dw[t] = r - 1, s - 1
        tc.widx[t] = (8 - k) if mirrored else k
    p.epi, p.out, p.ldC = epi, yfull.data_ptr(), ldC
    p.zeros, p.pipe = zeros.data_ptr(), 0x1 | 0x200 | 0x400 | pipe_extra
    p.a_bytes, p.w_bytes = xfull.numel() * 2, w.numel() * 2
    co = torch.rand(4, Cout, device=dev) + 0.5
    co[3] -= 1.0
    p.scale, p.shift, p.act = co.data_ptr() + 2 * Cout * 4, co.data_ptr() + 3 * Cout * 4, act
    rows, kern = S.I(), S.I()
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    assert kern.value & 0xff == expect_kernel, f"layer routed to kernel family {kern.value & 0xff}, expected {expect_kernel}"
    if expect_cols is not None:
        assert ((kern.value >> 16) & 15) * 32 == expect_cols, f"tile columns {((kern.value >> 16) & 15) * 32}, expected {expect_cols}"
    stats = torch.zeros(rows.value, 2, Cout, device=dev)
    p.stats = stats.data_ptr()
    hip.call("ryolo_conv_gemm", p, hip.stream())
    torch.cuda.synchronize()
    x = xfull[:, :Cin].float().view(B, H, W, Cin).permute(0, 3, 1, 2)
    wk = w.float().view(Cout, 3, 3, Cin)
    if mirrored:
        wk = wk.flip(1, 2)
    ref = torch.nn.functional.conv2d(x, wk.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    got = yfull[:, :Cout].float()
    if epi == S.EPI_AFFINE_ACT:
        u = ref * co[2] + co[3]
        ref = {3: u * torch.sigmoid(u), 2: torch.where(u > 0, u, 0.1 * u), 1: u * torch.tanh(torch.nn.functional.softplus(u)), 0: u}[act]
    if epi == S.EPI_ACCUM:
        ref = ref.to(torch.bfloat16).float() + y0[:, :Cout].float()
    err = (got - ref).abs()
    tol = 2.0 ** -7 * ref.abs() + 2e-2
    assert bool((err <= tol).all()), f"max err {float(err.max())} (rel {float((err / (ref.abs() + 1e-3)).max())})"
    assert float((got - ref).norm() / ref.norm()) < 4e-3
    if ld_extra:
        assert torch.equal(yfull[:, Cout:], y0[:, Cout:]), "wrote outside its channel slice"
    if epi == S.EPI_STATS:
        s1, s2 = stats[:, 0].sum(0), stats[:, 1].sum(0)
        assert torch.allclose(s1, got.sum(0), rtol=1e-4, atol=1e-2)
        assert torch.allclose(s2, (got * got).sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (2, 50, 50, 64, 128),      # 2-D tiles 10 x 25
    (3, 25, 25, 96, 128),      # flat runs, tiles span image boundaries, partial last tile
    (1, 20, 30, 32, 64),       # BN = 64 variant, non-square
    (2, 13, 13, 64, 192),      # odd size, Cout not a multiple of the N tile
    (1, 100, 100, 32, 64),     # 2-D tiles on a wide map
    (2, 7, 9, 160, 72),        # tiny map: every pixel is border; Cout multiple of 8 only
])
def test_patch_kernel_raw(B, H, W, Cin, Cout):
    _run(B, H, W, Cin, Cout)


@pytest.mark.parametrize("epi", [1, 2, 4])
@pytest.mark.parametrize("shape", [(2, 50, 50, 64, 128), (3, 25, 25, 64, 64)])
def test_patch_kernel_epilogues(epi, shape):
    _run(*shape, epi=epi, ld_extra=40, seed=epi)


@pytest.mark.parametrize("act", [0, 1, 2])
def test_patch_kernel_activations(act):
    _run(2, 25, 25, 64, 128, epi=2, act=act)


@pytest.mark.parametrize("epi", [0, 1, 2, 4])
def test_patch_kernel_64_column_tiles_on_wide_layers(epi):
    """r06: on grids of fewer 128-column workgroups than CUs the kernel runs its 256 x 64 tile also for Cout > 64 (twice the workgroups;
    pipe bit 0x2000 forces that tile): several column tiles per pixel tile, the last one ragged, statistics rows shared by the column tiles."""
    _run(2, 50, 50, 64, 128, epi=epi, ld_extra=40, seed=60 + epi, pipe_extra=0x2000, expect_cols=64)
    _run(2, 13, 13, 64, 192, epi=epi, seed=64 + epi, pipe_extra=0x2000, expect_cols=64)
    _run(3, 25, 25, 96, 136, epi=epi, seed=68 + epi, pipe_extra=0x2000, expect_cols=64)        # 3 column tiles, the last 8 columns wide


def test_patch_kernel_takes_small_grids_by_default():
    """r06 routing: without the force bit (0x400) a 3x3 stride-1 layer of the 8-image step's size goes to the halo-patch kernel, on 64-column
    tiles when its 128-column grid is smaller than the chip; r02-r05 kept every grid under 512 workgroups on the generic kernel."""
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    for (B, H, Cin, Cout, cols) in [(8, 25, 256, 256, 64), (8, 50, 128, 128, 64), (8, 100, 128, 128, 128), (64, 50, 128, 128, 128)]:
        p = S.ConvGemmParams()
        x = torch.zeros(16, device="cuda:0")
        p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = x.data_ptr(), B, H, H, Cin, Cin
        p.W, p.Nout, p.wtaps = x.data_ptr(), Cout, 9
        p.OH, p.OW, p.sh, p.sw = H, H, 1, 1
        p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, H, H
        p.nclasses = 1
        tc = p.cls[0]
        tc.ntaps = 9
        for k in range(9):
            tc.dh[k], tc.dw[k], tc.widx[k] = k // 3 - 1, k % 3 - 1, k
        p.epi, p.out, p.ldC, p.zeros, p.pipe = 1, x.data_ptr(), Cout, x.data_ptr(), 0x201
        rows, kern = S.I(), S.I()
        hip.call("ryolo_conv_gemm_plan", p, rows, kern)
        assert kern.value & 0xff == 1 and ((kern.value >> 16) & 15) * 32 == cols, (B, H, Cin, Cout, hex(kern.value))


def test_patch_kernel_mirrored_taps():
    _run(2, 50, 50, 128, 64, mirrored=True)
    _run(2, 26, 26, 64, 128, mirrored=True, epi=4)


# ---- the persistent weight-stationary kernel for 64 -> <= 64 channels (csrc/conv3x3_ws.hip): by default only when every workgroup gets >= 4 tiles
@pytest.mark.parametrize("epi", [0, 1, 2, 4])
def test_persistent_64_channel_kernel_multi_tile(epi):
    """8 x 200 x 200: 1280 tiles of 10 x 25 over 256 persistent workgroups — 5 tiles each, so the double-buffered patch pipeline (request two
    tiles ahead), the per-tile barrier and the register-accumulated statistics all run in their steady state; raw / statistics / folded BatchNorm + activation
    (r06: inference) / accumulate epilogues, output in a channel slice of a wider buffer."""
    _run(8, 200, 200, 64, 64, epi=epi, ld_extra=64, seed=10 + epi, expect_kernel=3)


def test_persistent_64_channel_kernel_mirrored_taps_and_narrow_output():
    _run(8, 200, 200, 64, 64, mirrored=True, epi=4, seed=20, expect_kernel=3)          # the data-gradient tap order
    _run(8, 200, 200, 64, 40, epi=1, seed=21, expect_kernel=3)                          # Cout = 40: zero weight rows, partial column chunks
