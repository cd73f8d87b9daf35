import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The oracle side of the tests is torch-CPU.  On the 256-thread GPU hosts torch's default (one OpenMP thread per hardware thread) is
# pathological for the many small operators a test issues and for the convolution backward of the oracle networks: the first full run
# of this suite spent 18 of its 35 minutes in two 8-second tests.  A bounded pool is faster there and changes nothing here.
os.environ.setdefault("OMP_NUM_THREADS", str(min(32, os.cpu_count() or 1)))
os.environ.setdefault("MKL_NUM_THREADS", os.environ["OMP_NUM_THREADS"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))


def pytest_collection_modifyitems(config, items):
    """`-m gpu` (or no marker filter) on a box without a HIP device: the gpu-marked tests SKIP with the reason instead of failing in
    whatever allocates first (pinned staging, a device tensor).  On the GPU box nothing is skipped by this."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False here)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
