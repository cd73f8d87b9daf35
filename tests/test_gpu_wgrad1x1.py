"""The 8-wave pointwise weight-gradient kernel (csrc/wgrad1x1_8w.hip: 256 x 256 tiles, two 64-KiB LDS-DMA stages, source-side bank swizzle)
through the C ABI against a float64 weight gradient on the same bf16 operands (model/utils.py:6-32 `Conv` with k = 1: autograd of
nn.Conv2d w.r.t. its weight).  Ragged tiles on both channel axes (Cout = 396 = the merged head width of the bench network, Cin = 320), channel
strides wider than the tensors (concat slices), K ranges that end inside a 64-pixel step, accumulation into an existing gradient; dispatch is
asserted (kernel 3), narrower layers stay on the 4-wave kernels by default (the idle-wave form of the 8-wave kernel is tested behind its knob).  Tolerance 2e-5 relative against the float64 product of
the same bf16 operands (tests/wgrad_ref.py; measured 2e-7).  The 4-wave kernels keep their coverage for these shapes through RYOLO_WGRAD_8W=0 in tests/test_gpu_forced_kernels.py."""
import pytest

from tests.test_gpu_wgrad_taps import _run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (20, 50, 50, 256, 256),       # one tile
    (4, 100, 100, 512, 256),      # two input-channel tiles
    (16, 25, 25, 1024, 396),      # ragged second output tile (140 of 256 channels), four input tiles
    (30, 47, 33, 320, 200),        # ragged on both axes, odd map: the last K range ends inside a step
    (8, 80, 80, 256, 512),
])
def test_pointwise_wgrad_8wave(B, H, W, Cin, Cout):
    _run(B, H, W, Cin, Cout, k=(1, 1), stride=1, seed=B + Cin, expect=3)


def test_pointwise_wgrad_8wave_concat_slices():
    _run(20, 50, 50, 256, 256, k=(1, 1), stride=1, ldx_extra=96, ldy_extra=64, seed=4, expect=3)
    _run(8, 64, 64, 512, 200, k=(1, 1), stride=1, ldx_extra=32, ldy_extra=16, seed=5, expect=3)


def test_narrow_pointwise_layers_stay_on_the_4wave_kernels():
    _run(8, 50, 50, 128, 256, k=(1, 1), stride=1, seed=6, expect=0)
    _run(8, 50, 50, 256, 128, k=(1, 1), stride=1, seed=7, expect=0)


def test_layers_narrower_than_the_tile_idle_some_waves():
    """RYOLO_WGRAD_8W_MINC=128 (off by default: -1 % on the step): Cout <= 128 (the wm = 1 waves have no block), Cin = 128 (the wn >= 2 waves), both —
    those waves only move data.  The knob is read once per process, hence the child."""
    import os
    import subprocess
    import sys
    code = ("from tests.test_gpu_wgrad_taps import _run\n"
            "_run(24, 50, 50, 128, 256, k=(1, 1), stride=1, seed=6, expect=3)\n"
            "_run(24, 50, 50, 256, 128, k=(1, 1), stride=1, seed=7, expect=3)\n"
            "_run(24, 50, 50, 128, 128, k=(1, 1), stride=1, seed=8, expect=3)\n"
            "_run(24, 50, 50, 192, 72, k=(1, 1), stride=1, seed=9, expect=3)\n"          # ragged first tile: 72 of 256 output channels
            "_run(24, 50, 50, 192, 64, k=(1, 1), stride=1, seed=10, expect=0)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, RYOLO_WGRAD_8W_MINC="128"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
