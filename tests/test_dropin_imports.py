"""The drop-in boundary as the reference's drivers see it.  A throw-away CALLER tree is written into
tmp_path — its own lib/__init__.py, lib/logger.py, lib/plot.py, lib/augmentations.py, model/__init__.py, model/utils.py,
datasets/base_dataset.py and a test.py, each a few lines written HERE (not the reference's files), plus hot-path files that raise when
imported — a fresh interpreter chdirs into it, calls ryolov4_amd.install_dropin() and executes EVERY import line of the reference's
drivers:

    train.py:13-17     from model.yolo import Yolo / from lib.load import load_data / from lib.logger import Logger, logger /
                       from lib.loss import ComputeCSLLoss, ComputeKFIoULoss / from test import test
    test.py:7-13       from detectron2.layers.rotated_boxes import pairwise_iou_rotated / from lib.general import post_process /
                       from lib.load import load_data / from lib.loss import ... / from lib.logger import logger / from model.yolo import Yolo
    detect.py:10-14    from lib.plot import plot_boxes / from lib.general import post_process /
                       from datasets.base_dataset import ImageDataset / from model.yolo import Yolo / from lib.logger import logger
    lib/plot.py:6      from lib.general import xywh2xyxy, xywha2xyxyxyxy        (inside the caller's own lib.plot)
    lib/loss.py:5-7    from detectron2.layers.rotated_boxes import pairwise_iou_rotated / from lib.general import norm_angle, xywhr2xywhrsigma
    lib/general.py:4   from detectron2.layers.nms import nms_rotated

Asserted: the hot-path names are this package's (HIP) objects, everything else is the caller's own, the caller's `lib` / `model`
packages are still the caller's (never replaced), and the caller's shadowed hot-path files were never executed.  Runs without a GPU
(imports only); tests/test_gpu_dropin.py runs the loops."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CALLER = {
    "lib/__init__.py": "",
    "lib/logger.py": "class Logger:\n    pass\nlogger = 'the caller logger'\n",
    "lib/plot.py": "from lib.general import xywh2xyxy, xywha2xyxyxyxy\ndef plot_boxes(*a):\n    return 'the caller plot_boxes', xywh2xyxy.__module__, xywha2xyxyxyxy.__module__\n",
    "lib/augmentations.py": "def hsv(*a):\n    return 'the caller hsv'\n",
    "lib/general.py": "raise ImportError('the caller lib/general.py must not be executed after install_dropin()')\n",
    "lib/loss.py": "raise ImportError('the caller lib/loss.py must not be executed after install_dropin()')\n",
    "lib/load.py": "raise ImportError('the caller lib/load.py must not be executed after install_dropin()')\n",
    "model/__init__.py": "",
    "model/utils.py": "WHO = 'the caller model.utils'\n",
    "model/yolo.py": "raise ImportError('the caller model/yolo.py must not be executed after install_dropin()')\n",
    # (the reference's datasets/ is a namespace package; this image has HuggingFace `datasets` installed, a regular package that would win
    # over a namespace portion, so the throw-away tree gets an __init__.py)
    "datasets/__init__.py": "",
    "datasets/base_dataset.py": "from lib.augmentations import hsv\nclass ImageDataset:\n    who = 'the caller ImageDataset'\n",
    "test.py": "from detectron2.layers.rotated_boxes import pairwise_iou_rotated\nfrom lib.general import post_process\n"
               "from lib.load import load_data\nfrom lib.loss import ComputeCSLLoss, ComputeKFIoULoss\nfrom lib.logger import logger\n"
               "from model.yolo import Yolo\ndef test():\n    return 'the caller test', post_process.__module__\n",
}

SCRIPT = r'''
import json, sys
sys.path.insert(0, "")                                         # a script's own directory, as `python train.py` has it
import ryolov4_amd
ryolov4_amd.install_dropin(DATASETS)
out = {}
# ---- train.py:13-17
from model.yolo import Yolo
from lib.load import load_data
from lib.logger import Logger, logger
from lib.loss import ComputeCSLLoss, ComputeKFIoULoss
from test import test
out["train"] = [Yolo.__module__, load_data.__module__, Logger.__module__, logger, ComputeCSLLoss.__module__, ComputeKFIoULoss.__module__, list(test())]
# ---- test.py:7-13
from detectron2.layers.rotated_boxes import pairwise_iou_rotated
from lib.general import post_process
from lib.load import load_data
from lib.loss import ComputeCSLLoss, ComputeKFIoULoss
from lib.logger import logger
from model.yolo import Yolo
out["test"] = [pairwise_iou_rotated.__module__, post_process.__module__]
# ---- detect.py:10-14
from lib.plot import plot_boxes
from lib.general import post_process
from datasets.base_dataset import ImageDataset
from model.yolo import Yolo
from lib.logger import logger
out["detect"] = [list(plot_boxes()), ImageDataset.__module__, getattr(ImageDataset, "who", None)]
# ---- inside the hot-path modules: lib/loss.py:5-7, lib/general.py:4
from lib.general import norm_angle, xywhr2xywhrsigma
from detectron2.layers.nms import nms_rotated
out["inner"] = [norm_angle.__module__, xywhr2xywhrsigma.__module__, nms_rotated.__module__]
import lib, model, lib.loss, model.yolo
import lib.augmentations, model.utils
out["packages"] = [lib.__file__, model.__file__, lib.loss.__name__, lib.general.__name__, lib.load.__name__, model.yolo.__name__,
                   lib.augmentations.hsv(), model.utils.WHO, sys.modules["lib"] is lib, sys.modules["model"] is model]
import torch
out["ops"] = [hasattr(torch.ops.detectron2, "nms_rotated"), hasattr(torch.ops.detectron2, "box_iou_rotated")]
print("RESULT " + json.dumps(out))
'''


def _run(tmp_path, datasets):
    for rel, text in CALLER.items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT.replace("DATASETS", "datasets=True" if datasets else "")], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])


def test_every_driver_import_line_runs_from_a_caller_tree(tmp_path):
    d = _run(tmp_path, datasets=False)
    ours = "ryolov4_amd."
    assert d["train"] == [ours + "model.yolo", ours + "lib.load", "lib.logger", "the caller logger", ours + "lib.loss", ours + "lib.loss",
                          ["the caller test", ours + "lib.general"]], d["train"]
    assert d["test"] == [ours + "lib.general", ours + "lib.general"], d["test"]
    # the caller's own lib.plot imported the HIP lib.general's helpers; the caller's datasets package is untouched by default
    assert d["detect"] == [["the caller plot_boxes", ours + "lib.general", ours + "lib.general"], "datasets.base_dataset", "the caller ImageDataset"], d["detect"]
    assert d["inner"] == [ours + "lib.general"] * 3, d["inner"]
    lib_file, model_file, *rest = d["packages"]
    assert os.path.samefile(lib_file, str(tmp_path / "lib" / "__init__.py")) and os.path.samefile(model_file, str(tmp_path / "model" / "__init__.py"))
    assert rest == [ours + "lib.loss", ours + "lib.general", ours + "lib.load", ours + "model.yolo", "the caller hsv", "the caller model.utils",
                    True, True], rest
    assert d["ops"] == [True, True]


def test_datasets_alias_is_opt_in(tmp_path):
    d = _run(tmp_path, datasets=True)
    assert d["detect"][1:] == ["ryolov4_amd.datasets.base_dataset", None], d["detect"]         # detect.py:12 gets the device-side ImageDataset
    assert d["detect"][0][0] == "the caller plot_boxes"


def test_real_detectron2_is_patched_not_shadowed(tmp_path):
    """with a real detectron2 importable, install_dropin() must not replace the package (its other submodules stay
    importable) — only the two functions are patched.  A fake installation stands in for it."""
    d2 = tmp_path / "site" / "detectron2"
    (d2 / "layers").mkdir(parents=True)
    (d2 / "__init__.py").write_text("__version__ = 'fake-0.6'\n")
    (d2 / "structures.py").write_text("class RotatedBoxes:\n    pass\n")
    (d2 / "layers" / "__init__.py").write_text("from .nms import nms_rotated\nfrom .rotated_boxes import pairwise_iou_rotated\n")
    (d2 / "layers" / "nms.py").write_text("def nms_rotated(*a):\n    raise RuntimeError('the CUDA op')\n")
    (d2 / "layers" / "rotated_boxes.py").write_text("def pairwise_iou_rotated(*a):\n    raise RuntimeError('the CUDA op')\n")
    code = textwrap.dedent('''
        import json, sys
        import ryolov4_amd
        ryolov4_amd.install_dropin()
        import detectron2
        from detectron2.structures import RotatedBoxes
        from detectron2.layers.nms import nms_rotated
        from detectron2.layers import pairwise_iou_rotated
        print("RESULT " + json.dumps([detectron2.__version__, RotatedBoxes.__module__, nms_rotated.__module__, pairwise_iou_rotated.__module__]))
    ''')
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, str(tmp_path / "site"), os.environ.get("PYTHONPATH", "")]), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])
    assert got == ["fake-0.6", "detectron2.structures", "ryolov4_amd.lib.general", "ryolov4_amd.lib.general"], got
