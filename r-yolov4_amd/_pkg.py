"""ryolov4_amd — MI355X-native (gfx950) hot path of yingkunwu/R-YOLOv4 behind the reference's own Python surface.

Layout:
  csrc/        hand-written HIP kernels + the C-ABI (include/ryolo.h)  -> csrc/libryolo_hip.so
  hip.py       ctypes binding of the C-ABI (raw device pointers + hipStream_t; fails loudly when the .so is missing)
  model/yolo.py, lib/loss.py, lib/general.py   the reference's three call surfaces (same names/arguments/errors)
  engine/      the static-graph executor that drives the conv stack (forward + backward tapes, hipGraph capture)
  parallel.py  one-process-per-GPU data parallel (RCCL all-reduce over xGMI)
"""
import os as _os
import sys as _sys

# Takes effect when this import precedes the first HIP call of the process (see bench.py): distinct hardware queues for the main,
# weight-gradient and collective streams.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__all__ = ["install_dropin"]


def install_dropin():
    """Make `from model.yolo import Yolo`, `from lib.loss import ComputeCSLLoss, ComputeKFIoULoss` and
    `from lib.general import post_process` (the imports of the reference's train.py:13-16, test.py:9-13,
    detect.py) resolve to this package."""
    import importlib
    for alias, real in (("model", "ryolov4_amd.model"), ("model.yolo", "ryolov4_amd.model.yolo"),
                        ("lib", "ryolov4_amd.lib"), ("lib.loss", "ryolov4_amd.lib.loss"),
                        ("lib.general", "ryolov4_amd.lib.general")):
        _sys.modules[alias] = importlib.import_module(real)
