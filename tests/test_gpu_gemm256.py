"""The 256-wide pointwise GEMM for long reductions (csrc/gemm256.hip): in-process at the sizes its default dispatch takes (Cin >= 512, >= 600
tiles), and FORCED (RYOLO_GEMM_256=2, read once per process: child process) onto small / ragged problems; forced onto every eligible 1x1 layer of
the block / network / per-node parity tests in tests/test_gpu_forced_kernels.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("epi", [0, 1, 2, 4])
def test_default_dispatch_takes_the_long_k_layers(epi):
    from tests.gemm256_cases import _run
    _run(64 * 50 * 50, 512, 512, epi=epi, ld_extra=64 if epi else 0, seed=epi)          # 512 -> 512 @50^2, batch 64: 1250 tiles
    _run(64 * 50 * 50, 1024, 128, epi=epi, seed=3 + epi, expect=0)                        # <= 128 columns: generic kernel by default (256 x 128 tiles only forced)
    _run(64 * 50 * 50, 1024, 256, epi=epi, seed=9 + epi)                                  # 625 tiles of 256 x 256
    _run(64 * 25 * 25, 2048, 512, epi=epi, seed=6 + epi, expect=0)         # 314 tiles: stays on the generic kernel


def _child(files, timeout):
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ, RYOLO_GEMM_256="2", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "pytest"] + files + ["-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_direct_cases_forced():
    _child(["tests/gemm256_cases.py"], 900)


# (the block / network / per-node suites with RYOLO_GEMM_256=2: tests/test_gpu_forced_kernels.py, merged child of r05)
