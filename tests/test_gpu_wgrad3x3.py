"""The 3x3 stride-1 halo-ring weight-gradient kernel (csrc/conv3x3.hip) through the C ABI against a float64 weight gradient
(tests/wgrad_ref.py) on the same bf16 operands: both variants (128-channel tiles; <= 64 output channels with split K halves), channel
strides wider than the tensors (concat slices), Cout that is not a multiple of 32, odd map sizes, accumulation into an existing
gradient.  Sizes are chosen so the library's own dispatch picks the ring kernel (asserted).  Tolerance 2e-5 relative: the operands are the same bf16 values, so what remains is fp32
accumulation against float64 — measured 2e-7 ... 4e-7 on every case (r05; against torch's fp32 MIOpen gradient, whose solver and rounding vary from box
to box, the tests had to allow 2e-3 and still failed once on a cold box).

Round 5: layers with Cin % 64 == 0 run on the 8-wave form (csrc/conv3x3_wgrad8.hip: 2 x (5 | 4) accumulator blocks per wave, rings of any
length with a mirrored head) by default — the cases below with Cin 64 / 128 / 192 / 256 exercise both of its instantiations (<= 64 output
channels: pixel halves + two slabs; 128-channel tiles: output-channel pairs), ring wrap-around over many laps (long K ranges on narrow maps),
maps as wide as its LDS budget allows, ragged Cout, concat strides; Cin = 32 stays on the 4-wave kernels, and the whole file runs once more
with RYOLO_W3_V8=0 in tests/test_gpu_forced_kernels.py so that those keep their coverage for every shape."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(B, H, W, Cin, Cout, ldx_extra=0, ldy_extra=0, seed=0):
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    S.check_layouts()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(seed)
    ldX, coutp = Cin + ldx_extra, ((Cout + 7) // 8) * 8
    ldY = coutp + ldy_extra
    x = torch.randn(B * H * W, ldX, generator=g).to(torch.bfloat16).to(dev)
    dy = torch.zeros(B * H * W, ldY, dtype=torch.bfloat16)
    dy[:, :Cout] = (torch.randn(B * H * W, Cout, generator=g) * 0.1).to(torch.bfloat16)
    dy[:, coutp:] = 3.0                                               # neighbouring slice of a concat buffer: must be ignored
    dy = dy.to(dev)
    dw0 = torch.randn(Cout, Cin, 9, generator=g).to(dev)
    dw = dw0.clone()
    zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
    p = S.WgradParams()
    p.dY, p.ldY, p.Cout, p.CoutPad = dy.data_ptr(), ldY, Cout, coutp
    p.X, p.NB, p.IH, p.IW, p.Cin, p.ldX = x.data_ptr(), B, H, W, Cin, ldX
    p.OH, p.OW, p.sh, p.sw, p.ntaps = H, W, 1, 1, 9
    for r in range(3):
        for s in range(3):
            p.dh[r * 3 + s], p.dw[r * 3 + s] = r - 1, s - 1
    p.dW, p.zeros = dw.data_ptr(), zeros.data_ptr()
    kern, sk, ws = S.I(), S.I(), S.Z()
    hip.call("ryolo_conv_wgrad_kernel", p, kern)
    assert kern.value == 1, "dispatch did not pick the halo-ring kernel for this shape"
    hip.call("ryolo_conv_wgrad_plan", p, sk, ws)
    work = torch.empty(ws.value, dtype=torch.uint8, device=dev)
    p.partial = work.data_ptr()
    hip.call("ryolo_conv_wgrad", p, hip.stream())
    torch.cuda.synchronize()
    from tests.wgrad_ref import wgrad_fp64
    ref = wgrad_fp64(x, dy, B, H, W, Cin, Cout, 3, 3, 1, 1, 1).float()            # float64 products on the same bf16 operands (tests/wgrad_ref.py)
    got = dw - dw0                                                    # the kernel ACCUMULATES into the gradient
    err = float((got - ref).norm() / ref.norm())
    assert err < 2e-5, f"relative error {err:.3e} (max abs {float((got - ref).abs().max()):.3e}, reference norm {float(ref.norm()):.3e})"
    assert bool(torch.isfinite(dw).all())


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (16, 64, 64, 64, 128),        # 128-channel tiles
    (12, 50, 50, 128, 256),       # two output-channel tiles, four input chunks
    (32, 33, 47, 64, 72),         # odd map, Cout not a multiple of 32 (rows of the last quarter masked)
    (48, 40, 40, 64, 64),         # <= 64 output channels: wave pairs split the K step
    (40, 60, 44, 32, 40),         # <= 64 variant, single input chunk, Cout = 40
])
def test_ring_wgrad(B, H, W, Cin, Cout):
    _run(B, H, W, Cin, Cout)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (64, 25, 25, 256, 256),       # the 25 x 25 layers of the step: 320-row rings, many laps per K range
    (16, 50, 50, 128, 128),       # one output tile, two 64-channel chunks
    (8, 100, 100, 128, 256),      # 448-row rings
    (8, 100, 100, 64, 64),        # <= 64 output channels: pixel halves, two slabs per range
    (4, 200, 200, 64, 64),        # 640-row rings
    (1, 400, 400, 64, 64),        # 1024-row rings: 148 KiB of LDS, the widest map of the 800 x 800 step
    (24, 31, 45, 192, 200),       # odd map, three chunks, ragged second output tile (72 of 128 channels)
    (48, 37, 29, 64, 40),         # <= 64 form with a ragged second quarter
    (128, 17, 23, 64, 128),       # tiny maps: a K range crosses many images, most ring rows are padding
])
def test_ring_wgrad_8wave_form(B, H, W, Cin, Cout):
    _run(B, H, W, Cin, Cout, seed=B + W)


def test_ring_wgrad_concat_slices():
    _run(16, 64, 64, 64, 128, ldx_extra=96, ldy_extra=64, seed=4)
    _run(48, 40, 40, 64, 48, ldx_extra=32, ldy_extra=16, seed=5)
