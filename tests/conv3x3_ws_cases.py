"""Cases for csrc/conv3x3_ws.hip with RYOLO_P3_WS64=2 (the persistent kernel on EVERY eligible launch, also where a workgroup gets one tile or
none).  Run through tests/test_gpu_conv3x3_ws.py in a child process: the switch is read once per process by the library."""
import pytest

from tests.test_gpu_conv3x3 import _run

pytestmark = pytest.mark.gpu

SHAPES = [
    (2, 50, 50, 64, 64),       # tiles 10 x 25: 20 tiles on 24 workgroups (some idle: they must still write zero statistics rows)
    (3, 100, 100, 64, 64),     # 120 tiles, one per workgroup
    (1, 48, 48, 64, 64),       # tiles 16 x 16 (patch 18 x 18), 9 tiles
    (2, 25, 50, 64, 40),       # tiles 5 x 50 (patch 7 x 52 = 46 pieces), Cout = 40
    (1, 16, 16, 64, 64),       # one tile = the whole image: every halo row is padding
    (9, 50, 50, 64, 8),        # 36 tiles on 40 workgroups, 8 output channels
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2, 4])
def test_forced_persistent_kernel(shape, epi):
    _run(*shape, epi=epi, ld_extra=24 if epi else 0, seed=epi + 3 * len(shape), expect_kernel=3)


def test_forced_persistent_kernel_mirrored():
    _run(2, 50, 50, 64, 64, mirrored=True, expect_kernel=3)
    _run(3, 100, 100, 64, 64, mirrored=True, epi=4, ld_extra=8, expect_kernel=3)


@pytest.mark.parametrize("act", [0, 1, 2])
def test_forced_persistent_kernel_inference_activations(act):
    """EPI_AFFINE_ACT (r06: the eval tape's 64 -> 64 layers): folded BatchNorm + every activation, narrow output (zero coefficients past Nout)."""
    _run(2, 50, 50, 64, 64, epi=2, act=act, ld_extra=8, seed=40 + act, expect_kernel=3)
    _run(2, 25, 50, 64, 40, epi=2, act=act, seed=50 + act, expect_kernel=3)


def test_not_eligible_shapes_stay_on_the_patch_kernel():
    _run(2, 50, 50, 128, 64, expect_kernel=1)           # Cin = 128: the weights do not fit the register file
    _run(2, 20, 30, 64, 64, expect_kernel=1)            # no tile of >= 224 pixels divides 20 x 30
