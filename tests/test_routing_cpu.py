"""CPU: which kernel family `ryolo_conv_gemm` will launch for a layer (host-side geometry only: `ryolo_conv_gemm_plan` never touches its pointers).
Guards the dispatch rules of DESIGN.md §3.4 / §4 against silent drift: the r06 small-grid routing of the 3x3 stride-1 layers (halo-patch kernel on
every grid of >= 4 tiles, 64-column tiles when the 128-column grid is smaller than the chip), the persistent 64-channel kernel's grid gate, the 256-wide
pointwise GEMM's size gates, and that the inference epilogue (EPI_AFFINE_ACT, r06) takes the same kernels as the training epilogues."""
import pytest


def _plan(B, H, Cin, Cout, k=3, s=1, epi=1, pipe=0x201):
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    p = S.ConvGemmParams()
    pad = (k - 1) // 2
    OH = (H + 2 * pad - k) // s + 1
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = 0x1000, B, H, H, Cin, Cin
    p.W, p.Nout, p.wtaps = 0x2000, Cout, k * k
    p.OH, p.OW, p.sh, p.sw = OH, OH, s, s
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, OH, OH
    p.nclasses = 1
    tc = p.cls[0]
    tc.ntaps = k * k
    for r in range(k):
        for c in range(k):
            tc.dh[r * k + c], tc.dw[r * k + c], tc.widx[r * k + c] = r - pad, c - pad, r * k + c
    p.epi, p.out, p.ldC, p.zeros, p.pipe = epi, 0x3000, Cout, 0x4000, pipe
    p.scale, p.shift = 0x5000, 0x6000
    rows, kern = S.I(), S.I()
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    return kern.value & 0xff, ((kern.value >> 16) & 15) * 32, rows.value


GENERIC, PATCH, WS1X1, WS64, GEMM256, S2C32 = 0, 1, 2, 3, 4, 5


@pytest.mark.parametrize("epi", [0, 1, 2, 4])
def test_3x3_stride1_routing(epi):
    # the 8-image step's small maps: halo-patch kernel, 64-column tiles while the 128-column grid has fewer workgroups than the chip has CUs
    assert _plan(8, 25, 256, 256, epi=epi)[:2] == (PATCH, 64)           # 20 pixel tiles x 2
    assert _plan(8, 25, 1024, 512, epi=epi)[:2] == (PATCH, 64)
    assert _plan(8, 50, 128, 128, epi=epi)[:2] == (PATCH, 64)           # 79 tiles
    assert _plan(8, 100, 128, 128, epi=epi)[:2] == (PATCH, 128)         # 320 tiles: the wide tile from here on
    assert _plan(1, 25, 256, 256, epi=epi)[:2] == (PATCH, 64)           # batch 1: 3 pixel tiles x 4 column tiles
    # the batch-64 step
    assert _plan(64, 25, 256, 256, epi=epi)[:2] == (PATCH, 128)         # 314 workgroups: generic kernel until r06
    assert _plan(64, 50, 128, 128, epi=epi)[:2] == (PATCH, 128)
    # 64 -> 64 channels: persistent kernel when every workgroup gets >= 4 tiles, else the patch kernel's 64-column tile
    assert _plan(64, 400, 64, 64, epi=epi)[0] == WS64
    assert _plan(8, 200, 64, 64, epi=epi)[0] == WS64                    # 1 280 tiles on 256 workgroups
    assert _plan(8, 100, 64, 64, epi=epi)[:2] == (PATCH, 64)            # 320 tiles
    # statistics rows = what the EPI_STATS epilogue writes: pixel tiles (patch kernel), workgroups (persistent kernel)
    assert _plan(8, 25, 256, 256, epi=epi)[2] == 20 and _plan(64, 400, 64, 64, epi=epi)[2] == 256


def test_stride2_and_pointwise_routing():
    assert _plan(64, 800, 32, 64, k=3, s=2)[0] == S2C32                  # the streaming kernel of the second layer (r06)
    assert _plan(64, 400, 64, 128, k=3, s=2)[0] == GENERIC
    assert _plan(8, 50, 256, 256, k=3, s=2)[0] == GENERIC                 # small stride-2 grids stay generic (measured: docs/HISTORY.md I)
    for epi in (0, 1, 2, 4):
        assert _plan(64, 100, 512, 512, k=1, epi=epi)[:2] == (GEMM256, 256)
        assert _plan(64, 25, 2048, 512, k=1, epi=epi)[0] == GENERIC      # 314 tiles < 600
        assert _plan(64, 100, 512, 128, k=1, epi=epi)[0] == GENERIC      # 128 output columns: the 256 x 128 tile only when forced
        assert _plan(64, 200, 256, 256, k=1, epi=epi)[0] == WS1X1        # K <= 256: weight-stationary persistent kernel
        assert _plan(8, 25, 1024, 1024, k=1, epi=epi)[0] == GENERIC      # small pointwise grids: generic (with its deep ring)


def test_force_bits():
    assert _plan(64, 25, 256, 256, pipe=0x201 | 0x2000)[:2] == (PATCH, 64)     # 0x2000: 64-column tile forced
    assert _plan(8, 25, 256, 256, pipe=0x201 | 0x400)[:2] == (PATCH, 128)      # 0x400 (tests): this kernel, 128-column tile
    assert _plan(8, 25, 256, 256, pipe=0x001)[0] == GENERIC                   # without 0x200 the 3x3 kernels are not considered
