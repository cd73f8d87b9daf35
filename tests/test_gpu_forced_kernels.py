"""The per-block and whole-network parity tests once more with the 3x3 halo-patch kernel (forward / data gradient) and the
halo-ring weight-gradient kernel FORCED onto the small grids those tests use (by default the dispatch keeps small problems on
the generic kernels, so the network-level tests would never reach the new ones).  The switches are read once per process
(RYOLO_GEMM_PIPE by the runtime, RYOLO_W3_FORCE by the library), hence the subprocess."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_and_network_parity_with_3x3_kernels_gemm256_and_deep_rings_forced():
    """THREE disjoint kernel selections in one child (r05: each child run of the per-block / whole-network / per-node suites costs the GPU suite
    ~25 s; until r04 these were three runs): RYOLO_GEMM_PIPE=0x601 + RYOLO_W3_FORCE=1 — every 3x3 stride-1 layer on the halo-patch kernel and the
    ring weight gradients (8-wave form where Cin % 64 == 0) on the small grids of those tests; RYOLO_GEMM_256=2 — every pointwise layer with
    Cin >= 512 on the 256-wide kernel; RYOLO_GEMM_DEEP=6 — the 6-stage ring on every remaining launch of the generic 128 x 128 tile (tapped /
    strided layers, short-K pointwise layers).  No launch is claimed by two of them."""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()                       # the child process cannot reuse this process's cached blocks
    env = dict(os.environ, RYOLO_GEMM_PIPE="0x601", RYOLO_W3_FORCE="1", RYOLO_GEMM_256="2", RYOLO_GEMM_DEEP="6")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_blocks.py", "tests/test_gpu_model.py", "tests/test_gpu_teacher_forced.py", "-q", "-m", "gpu",
                        "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_block_and_network_parity_with_wide_64_column_tile_and_persistent_3x3_forced():
    """RYOLO_GEMM_N64=2: the 256 x 64 tile of the generic kernel (layers with 33..64 output columns, by default only when the grid has
    >= 1536 such tiles) on the small grids of the block / network parity tests, batch-statistics epilogue included; in the same child (r05)
    RYOLO_P3_WS64=2: the persistent weight-stationary 3x3 kernel on every 64 -> <= 64 channel 3x3 stride-1 layer (those never reach the generic
    kernel's tiles: disjoint launches).  Its direct C-ABI cases: tests/test_gpu_conv3x3_ws.py."""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ, RYOLO_GEMM_N64="2", RYOLO_P3_WS64="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_blocks.py", "tests/test_gpu_model.py", "tests/test_gpu_teacher_forced.py", "-q", "-m", "gpu",
                        "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_block_and_network_parity_with_persistent_1x1_kernel_forced():
    """RYOLO_GEMM_WS=2: the weight-stationary persistent 1x1 kernel (gemm1x1.hip; by default only grids with >= 6 tiles per wave) on the
    small, ragged grids of the block / network / per-node parity tests: raw, statistics, accumulate and (eval plans) affine epilogues."""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ, RYOLO_GEMM_WS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_blocks.py", "tests/test_gpu_model.py", "tests/test_gpu_teacher_forced.py",
                        "tests/test_gpu_parity_e2e.py", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_block_and_network_parity_with_deep_ring_4_forced():
    """RYOLO_GEMM_DEEP=4: the 4-stage instantiation of the generic 128 x 128 tile (by default chosen for grids of <= 512 tiles) on every eligible
    launch of the block / network parity tests (the 6-stage one runs in the merged child above, per-node suite included)."""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ, RYOLO_GEMM_DEEP="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_blocks.py", "tests/test_gpu_model.py",
                        "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_block_and_network_parity_with_the_alternatives_of_the_r04_defaults():
    """The instantiations the end-of-r04 defaults moved AWAY from stay covered: 64-channel stages of the generic tapped GEMM (RYOLO_GEMM_PIPE
    without 0x100), 32-pixel steps of the pointwise LDS-DMA weight gradient (RYOLO_WGRAD_P1=3), the register-staged tapped weight gradient
    (RYOLO_WGRAD_TAPS_DMA=0), and (r05) the 4-wave ring weight-gradient kernels that the 8-wave form replaced for Cin % 64 == 0
    (RYOLO_W3_V8=0; their direct C-ABI cases run here too), the 4-wave pointwise kernels behind the 8-wave 256 x 256 form (RYOLO_WGRAD_8W=0) and the
    r01-r04 form of the detection heads (RYOLO_HEAD_FUSED=0: row-major fp32 intermediate + ryolo_head_finish_fwd_obj, ImplicitA as a pass, the ImplicitM
    gradient from the pre-activations, the sparse head backward WITH its compact pre-activation column) — independent kernel selections, selected together in ONE child process (each
    run of the per-block, whole-network and per-node parity tests costs the GPU suite ~25 s)."""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ, RYOLO_GEMM_PIPE="0x201", RYOLO_WGRAD_P1="3", RYOLO_WGRAD_TAPS_DMA="0", RYOLO_W3_V8="0", RYOLO_WGRAD_8W="0", RYOLO_HEAD_FUSED="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_blocks.py", "tests/test_gpu_model.py", "tests/test_gpu_teacher_forced.py",
                        "tests/test_gpu_wgrad3x3.py", "-q", "-m", "gpu",
                        "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
