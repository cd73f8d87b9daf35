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


# The reference's hot-path modules and what they resolve to after install_dropin().  ONLY these names are aliased: the caller's
# own `lib` / `model` / `datasets` packages stay importable, so `from lib.logger import Logger, logger` (train.py:15, test.py:12,
# detect.py:14), `from lib.plot import plot_boxes` (detect.py:10) and `lib.augmentations` keep resolving to the caller's files.
HOT_PATH_ALIASES = {"model.yolo": "ryolov4_amd.model.yolo", "lib.loss": "ryolov4_amd.lib.loss", "lib.general": "ryolov4_amd.lib.general",
                    "lib.load": "ryolov4_amd.lib.load"}
# install_dropin(datasets=True): detect.py:12 `from datasets.base_dataset import ImageDataset` (and anything subclassing BaseDataset)
# gets the device-side loader classes as well.  Off by default: the caller's datasets package works unchanged beside the HIP path.
DATASET_ALIASES = {"datasets.base_dataset": "ryolov4_amd.datasets.base_dataset", "datasets.DOTA_dataset": "ryolov4_amd.datasets.DOTA_dataset",
                   "datasets.UCASAOD_dataset": "ryolov4_amd.datasets.UCASAOD_dataset"}


def _is_ours(mod):
    return getattr(mod, "__name__", "").startswith("ryolov4_amd")


class _EmptyPackageFinder:
    """LAST entry of sys.meta_path: when no finder knows a parent package of an aliased module (a caller without its own `lib/` or
    `model/` directory — e.g. a script next to this repo), an empty package carrying the aliases is synthesised so that
    `from lib.loss import ...` still resolves.  A caller that HAS the package never gets here: its own files win."""

    def __init__(self):
        self.children = {}                                          # parent name -> {attribute: module}

    def find_spec(self, name, path=None, target=None):
        if name not in self.children:
            return None
        import importlib.machinery
        spec = importlib.machinery.ModuleSpec(name, self, is_package=True)
        spec.submodule_search_locations = []
        return spec

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        module.__doc__ = "ryolov4_amd: empty stand-in package (the caller has none of this name); only the aliased hot-path submodules exist"
        for attr, mod in self.children[module.__name__].items():
            setattr(module, attr, mod)


_fallback = _EmptyPackageFinder()


def _bind(alias, mod):
    """sys.modules[alias] = mod, and — when the caller's parent package is importable now — also the attribute on it, as the import
    system would have set it (`import lib.loss; lib.loss.X`).  The parent itself is never replaced."""
    import importlib
    import importlib.util
    _sys.modules[alias] = mod
    parent, _, attr = alias.rpartition(".")
    if not parent:
        return
    _fallback.children.setdefault(parent, {})[attr] = mod
    pkg = _sys.modules.get(parent)
    if pkg is not None and _is_ours(pkg):                           # left behind by the round-3 install_dropin: drop the shadow
        del _sys.modules[parent]
        pkg = None
    if pkg is None:
        try:
            found = importlib.util.find_spec(parent) is not None
        except (ImportError, ValueError):
            found = False
        if found:
            try:
                pkg = importlib.import_module(parent)              # the caller's own package (its __init__ runs, as on its first import)
            except Exception:
                pkg = None                                         # broken caller package: its own import will say so later
    if pkg is not None:
        setattr(pkg, attr, mod)


def install_dropin(detectron2=None, datasets=False):
    """Make `from model.yolo import Yolo`, `from lib.loss import ComputeCSLLoss, ComputeKFIoULoss`, `from lib.general import
    post_process`, `from lib.load import load_data` (train.py:13-16, test.py:9-13, detect.py:10-14) resolve to this package —
    as `sys.modules` entries for exactly those four submodules (HOT_PATH_ALIASES) plus attributes on the CALLER's `lib` / `model`
    packages, which stay the caller's: `lib.logger`, `lib.plot`, `lib.augmentations`, `model.utils` ... import as before.

    detectron2: None (default) = install the stand-ins for `detectron2.layers.nms.nms_rotated` / `detectron2.layers.rotated_boxes.
    pairwise_iou_rotated` (lib/general.py:4, lib/loss.py:5, test.py:7) only when no real detectron2 is importable, otherwise patch
    those two functions on the real modules; True = always install the stand-ins; False = leave detectron2 alone.  Either way
    torch.ops.detectron2.{nms_rotated, box_iou_rotated} are registered unless a real detectron2 already owns the namespace.

    datasets=True additionally aliases `datasets.base_dataset` / `DOTA_dataset` / `UCASAOD_dataset` (DATASET_ALIASES) to the
    device-side loader classes (ImageDataset for detect.py:12,43)."""
    import importlib
    import importlib.util
    if detectron2 is not False:
        real_d2 = False
        if detectron2 is None:
            m = _sys.modules.get("detectron2")
            if m is not None:
                real_d2 = not (getattr(m, "__doc__", "") or "").startswith("ryolov4_amd stand-in")
            else:
                try:
                    real_d2 = importlib.util.find_spec("detectron2") is not None
                except (ImportError, ValueError):
                    real_d2 = False
        if real_d2:
            from .lib import general
            try:
                nms_mod = importlib.import_module("detectron2.layers.nms")
                rb_mod = importlib.import_module("detectron2.layers.rotated_boxes")
                layers = importlib.import_module("detectron2.layers")
                nms_mod.nms_rotated = layers.nms_rotated = general.nms_rotated
                rb_mod.pairwise_iou_rotated = layers.pairwise_iou_rotated = general.pairwise_iou_rotated
            except Exception:                                       # an installation that cannot be imported: stand-ins after all
                _sys.modules.update(_detectron2_modules())
        else:
            _sys.modules.update(_detectron2_modules())
        _register_torch_ops()
    if _fallback not in _sys.meta_path:
        _sys.meta_path.append(_fallback)
    aliases = dict(HOT_PATH_ALIASES)
    if datasets:
        aliases.update(DATASET_ALIASES)
    for alias, real in aliases.items():
        _bind(alias, importlib.import_module(real))
