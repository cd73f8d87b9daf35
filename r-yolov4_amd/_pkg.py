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


def _detectron2_modules():
    """In-memory stand-ins for the four detectron2 module names the reference imports (lib/general.py:4 `from detectron2.layers.nms
    import nms_rotated`, lib/loss.py:5 and test.py:7 `from detectron2.layers.rotated_boxes import pairwise_iou_rotated`), carrying
    this package's HIP ops.  detectron2 itself is not a dependency of the build (BASELINE north star: the CUDA dependency is replaced)."""
    import types
    from .lib import general
    d2 = types.ModuleType("detectron2")
    layers = types.ModuleType("detectron2.layers")
    nms = types.ModuleType("detectron2.layers.nms")
    rb = types.ModuleType("detectron2.layers.rotated_boxes")
    d2.__path__, layers.__path__ = [], []                          # packages: `import detectron2.layers.nms` resolves through sys.modules
    nms.nms_rotated = general.nms_rotated
    rb.pairwise_iou_rotated = general.pairwise_iou_rotated
    layers.nms_rotated = general.nms_rotated                       # detectron2.layers re-exports both names
    layers.pairwise_iou_rotated = general.pairwise_iou_rotated
    layers.nms, layers.rotated_boxes, d2.layers = nms, rb, layers
    d2.__doc__ = "ryolov4_amd stand-in: rotated NMS / IoU on MI355X HIP kernels (csrc/nms.hip)"
    return {"detectron2": d2, "detectron2.layers": layers, "detectron2.layers.nms": nms, "detectron2.layers.rotated_boxes": rb}


_ops_lib = None


def _register_torch_ops():
    """torch.ops.detectron2.{nms_rotated, box_iou_rotated} with detectron2's schemas (fourth surface), dispatched to the HIP
    kernels for device tensors.  Skipped when a real detectron2 already owns the namespace."""
    global _ops_lib
    if _ops_lib is not None:
        return
    import torch
    from .lib import general
    try:
        lib = torch.library.Library("detectron2", "DEF")
        lib.define("nms_rotated(Tensor boxes, Tensor scores, float iou_threshold) -> Tensor")
        lib.define("box_iou_rotated(Tensor boxes1, Tensor boxes2) -> Tensor")
    except RuntimeError:
        return                                                      # already defined by an installed detectron2: leave it alone
    lib.impl("nms_rotated", lambda boxes, scores, iou_threshold: general.nms_rotated(boxes, scores, iou_threshold), "CUDA")
    lib.impl("box_iou_rotated", lambda a, b: general.pairwise_iou_rotated(a, b), "CUDA")
    _ops_lib = lib


def install_dropin(detectron2=True):
    """Make `from model.yolo import Yolo`, `from lib.loss import ComputeCSLLoss, ComputeKFIoULoss`,
    `from lib.general import post_process` (the imports of the reference's train.py:13-16, test.py:9-13, detect.py) and — unless
    detectron2=False — `from detectron2.layers.nms import nms_rotated` / `from detectron2.layers.rotated_boxes import
    pairwise_iou_rotated` (lib/general.py:4, lib/loss.py:5, test.py:7) resolve to this package, and register
    torch.ops.detectron2.{nms_rotated, box_iou_rotated}."""
    import importlib
    for alias, real in (("model", "ryolov4_amd.model"), ("model.yolo", "ryolov4_amd.model.yolo"),
                        ("lib", "ryolov4_amd.lib"), ("lib.loss", "ryolov4_amd.lib.loss"),
                        ("lib.general", "ryolov4_amd.lib.general"), ("lib.load", "ryolov4_amd.lib.load")):
        _sys.modules[alias] = importlib.import_module(real)
    if detectron2:
        _sys.modules.update(_detectron2_modules())
        _register_torch_ops()
