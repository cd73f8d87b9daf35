"""The persistent weight-stationary 3x3 kernel (csrc/conv3x3_ws.hip) FORCED onto small problems (RYOLO_P3_WS64=2: by default it only takes
launches that give every workgroup >= 4 tiles; tests/test_gpu_conv3x3.py covers that regime in-process): direct C-ABI cases against
torch's conv2d (tests/conv3x3_ws_cases.py) and the block / network / per-node parity tests with the kernel on every 64 -> <= 64 channel
3x3 layer.  The switch is read once per process by the library, hence the child processes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(files, timeout):
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ, RYOLO_P3_WS64="2", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "pytest"] + files + ["-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_direct_cases_with_the_persistent_kernel_forced():
    _child(["tests/conv3x3_ws_cases.py"], 900)


# (the block / network / per-node suites with RYOLO_P3_WS64=2: tests/test_gpu_forced_kernels.py, merged child of r05)
