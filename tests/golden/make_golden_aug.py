"""Fixture G11: mosaic-4 / mosaic-9 assembly, their label handling, mixup and the label side of random_warping, produced by RUNNING THE
REFERENCE (datasets/base_dataset.py:188-330, lib/augmentations.py:24-28,45-74 imported from /root/reference) on synthetic images:
only cv2 I/O is replaced (load_image / load_files return arrays instead of reading files; cv2.getRotationMatrix2D is OpenCV's closed
form, cv2.warpPerspective returns oracle/ref_data.py's restatement — pixels of the warp are therefore NOT pinned, its labels are).
The random draws the reference makes are recorded so that the build's host code can be replayed with the same values.
Run here:  python tests/golden/make_golden_aug.py"""
import math
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from tests.golden import make_golden as MG  # noqa: E402
from oracle import ref_data  # noqa: E402


def main():
    MG._install_stubs()
    cv2 = sys.modules["cv2"]
    cv2.getRotationMatrix2D = lambda angle, center, scale: np.array(
        [[scale * math.cos(angle * math.pi / 180), scale * math.sin(angle * math.pi / 180), 0.0],
         [-scale * math.sin(angle * math.pi / 180), scale * math.cos(angle * math.pi / 180), 0.0]])
    cv2.warpPerspective = lambda img, M, dsize, borderValue: ref_data.warp_perspective_numpy(img, M, dsize, borderValue[0])
    os.chdir(MG.REF)
    sys.path.insert(0, MG.REF)
    import importlib.util
    from lib import augmentations as raug
    spec = importlib.util.spec_from_file_location("ref_base_dataset", os.path.join(MG.REF, "datasets", "base_dataset.py"))   # `datasets` clashes with the HF package
    BD = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(BD)
    BaseDataset = BD.BaseDataset

    rng = np.random.RandomState(0)
    S = 32
    shapes = [(32, 24), (20, 32), (32, 32), (18, 26), (32, 17), (25, 32), (31, 32), (32, 30), (16, 32), (32, 21), (28, 32), (32, 32)]
    images = [rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in shapes]
    polys, labels = [], []
    for h, w in shapes:
        n = rng.randint(0, 6)
        c = rng.rand(n, 2)
        d = rng.rand(n, 4, 2) * 0.2 - 0.1
        polys.append(torch.tensor((c[:, None, :] + d).reshape(n, 8), dtype=torch.float32))     # normalised polygons
        labels.append(torch.tensor(rng.randint(0, 3, size=n), dtype=torch.float32))

    class DS(BaseDataset):
        def __init__(self):
            super().__init__({"mosaic": 1.0}, S, True, False, True)
            self.img_files = [__file__] * len(images)
            self.label_files = [__file__] * len(images)
            self._cur = None

        def load_image(self, index):
            self._cur = index
            img = images[index].copy()
            return img, img.shape[:2], img.shape[:2]

        def load_files(self, label_path):
            return polys[self._cur].clone(), labels[self._cur].clone()

    ds = DS()
    draws = []
    real_uniform, real_choices = random.uniform, random.choices

    def rec_uniform(a, b):
        v = real_uniform(a, b)
        draws.append(("u", v))
        return v

    def rec_choices(pop, k):
        v = real_choices(pop, k=k)
        draws.append(("c", list(v)))
        return v
    random.uniform, random.choices = rec_uniform, rec_choices
    g = {"S": np.array(S), "nimg": np.array(len(images))}
    for i, im in enumerate(images):
        g[f"img{i}"] = im
        g[f"polys{i}"] = polys[i].numpy()
        g[f"labels{i}"] = labels[i].numpy()
    for case in range(4):
        random.seed(100 + case)
        draws.clear()
        img4, lab4 = ds.load_mosaic(case)
        yc, xc = int(draws[0][1]), int(draws[1][1])
        idx = [case] + draws[2][1]
        o4, meta = ref_data.mosaic4_numpy([images[i] for i in idx], S, yc, xc)
        assert np.array_equal(o4, img4), case
        g[f"m4_{case}_idx"] = np.array(idx)
        g[f"m4_{case}_yc_xc"] = np.array([yc, xc])
        g[f"m4_{case}_img"] = img4
        g[f"m4_{case}_labels"] = lab4.numpy()
        draws.clear()
        random.seed(200 + case)
        img9, lab9 = ds.load_mosaic9(case)
        idx9 = [case] + draws[0][1]
        yc9, xc9 = int(draws[1][1]), int(draws[2][1])
        g[f"m9_{case}_idx"] = np.array(idx9)
        g[f"m9_{case}_yc_xc"] = np.array([yc9, xc9])
        g[f"m9_{case}_img"] = img9
        g[f"m9_{case}_labels"] = lab9.numpy()
    # mixup (np.random.beta recorded)
    real_beta = np.random.beta
    for case in range(2):
        box = {}
        np.random.beta = lambda a, b: box.setdefault("r", real_beta(a, b))
        np.random.seed(5 + case)
        a, b = g[f"m4_{case}_img"], g[f"m4_{case + 2}_img"]
        la, lb = torch.from_numpy(g[f"m4_{case}_labels"]), torch.from_numpy(g[f"m4_{case + 2}_labels"])
        mi, ml = raug.mixup(a, la, b, lb)
        assert np.array_equal(mi, ref_data.mixup_numpy(a, b, box["r"]))
        g[f"mix_{case}_r"] = np.array(box["r"])
        g[f"mix_{case}_img"] = mi
        g[f"mix_{case}_labels"] = ml.numpy()
    np.random.beta = real_beta
    # random_warping: labels through the reference's own matrix product; the draws are recorded
    for case in range(3):
        random.seed(300 + case)
        draws.clear()
        img, tg = g[f"m4_{case}_img"].copy(), torch.from_numpy(g[f"m4_{case}_labels"].copy())
        border = (-S // 2, -S // 2)
        out, tg2 = raug.random_warping(img, tg, 10, 0.9, 0.1, border)
        a, s, tx, ty = [d[1] for d in draws[:4]]
        g[f"warp_{case}_draws"] = np.array([a, s, tx, ty])
        g[f"warp_{case}_labels"] = tg2.numpy()
        g[f"warp_{case}_img_unpinned"] = out
    random.uniform, random.choices = real_uniform, real_choices
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g11_aug.npz"), **g)
    print("G11 ok", {k: v.shape for k, v in g.items() if k.startswith("m9_0")})


if __name__ == "__main__":
    main()
