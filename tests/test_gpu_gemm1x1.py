"""The weight-stationary persistent 1x1 kernel (csrc/gemm1x1.hip, off by default) through the C ABI: the cases of tests/gemm1x1_cases.py
(ragged pixel counts and output widths, channel slices, every epilogue, K = 64 ... 256, bit repeatability) in a child process with
RYOLO_GEMM_WS=2 — the knob is read once per process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_persistent_1x1_kernel_through_the_c_abi():
    env = dict(os.environ, RYOLO_GEMM_WS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "gemm1x1_cases.py"), "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
