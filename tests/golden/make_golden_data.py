#!/usr/bin/env python3
"""Generate tests/golden/g9_data.npz (SURVEY.md §8(f) N2 slice + N4) by IMPORTING THE REFERENCE in the build container and
assert that oracle/ref_data.py reproduces it on every case.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_data.py

What runs for real: BaseDataset.__getitem__ (datasets/base_dataset.py:94-157, the non-mosaic branch) and collate_fn (:159-166)
on a subclass whose load_image / load_target hand back synthetic arrays (no files, no cv2.imread); lib.general.xyxyxyxy2xywha,
xywha2xyxyxyxy; lib.plot.rescale_boxes; datasets.base_dataset.gaussian_label.
Stubs (absent here): detectron2; cv2 — `copyMakeBorder` (numpy pad; the synthetic images are already square so every border is
0), `getRotationMatrix2D` (OpenCV's documented closed form = oracle.ref_data.get_rotation_matrix_2d: third-party, "parity
unpinned"), and `random_warping` is replaced by the identity with a recorder (it needs cv2.warpPerspective; the warp is upstream of
the slice this fixture pins).  Only data is committed; the reference never travels.
"""
import importlib.util
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import ref_data                      # noqa: E402


def _install_stubs():
    mods = {n: types.ModuleType(n) for n in ("detectron2", "detectron2.layers", "detectron2.layers.rotated_boxes", "detectron2.layers.nms", "cv2")}
    mods["detectron2.layers.rotated_boxes"].pairwise_iou_rotated = lambda *a: None
    mods["detectron2.layers.nms"].nms_rotated = lambda *a: None
    cv2 = mods["cv2"]
    cv2.BORDER_CONSTANT, cv2.INTER_LINEAR, cv2.INTER_AREA = 0, 1, 3
    cv2.getRotationMatrix2D = lambda center, angle, scale: ref_data.get_rotation_matrix_2d(center, angle, scale)

    def copy_make_border(img, top, bottom, left, right, kind, value=None):
        assert top == bottom == left == right == 0, "fixture images are square at the network size"
        return img
    cv2.copyMakeBorder = copy_make_border
    sys.modules.update(mods)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synth_polys(g, n, S, edge=False):
    """n clockwise rectangles (image coordinates, y down) as [n, 8] float32 pixels; some centres outside the image when edge."""
    cx = g.uniform(-0.05 * S if edge else 0.1 * S, 1.05 * S if edge else 0.9 * S, n)
    cy = g.uniform(-0.05 * S if edge else 0.1 * S, 1.05 * S if edge else 0.9 * S, n)
    a = g.uniform(4, 60, n)
    b = g.uniform(4, 60, n)
    th = g.uniform(-np.pi, np.pi, n)
    out = np.zeros((n, 8), np.float32)
    for k, (sx, sy) in enumerate(((-1, -1), (1, -1), (1, 1), (-1, 1))):          # clockwise with y pointing down
        out[:, 2 * k] = cx + sx * a / 2 * np.cos(th) - sy * b / 2 * np.sin(th)
        out[:, 2 * k + 1] = cy + sx * a / 2 * np.sin(th) + sy * b / 2 * np.cos(th)
    return out


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    _install_stubs()
    import lib.general as G                       # noqa: E402  (reference)
    import lib.plot as P                          # noqa: E402
    import lib.augmentations as A                 # noqa: E402
    BD = _load("ref_base_dataset", os.path.join(REF, "datasets", "base_dataset.py"))   # `datasets` clashes with the HF package
    out = {}

    # ---- label geometry: xyxyxyxy2xywha, gaussian_label -------------------------------------------------------------------
    g = np.random.default_rng(0)
    polys = synth_polys(g, 64, 256)
    polys[0] = [10, 10, 30, 10, 30, 20, 10, 20]                                    # SURVEY quirk check: -> [20, 15, 10, 20, 0]
    polys[1] = [10, 10, 20, 10, 20, 30, 10, 30]                                    # tall box
    polys[2] = [0, 0, 16, 0, 16, 16, 0, 16]                                        # square: w >= h branch with theta == 0
    pt = torch.from_numpy(polys)
    ref = G.xyxyxyxy2xywha(pt.clone())
    mine = ref_data.xyxyxyxy2xywha(pt.clone())
    assert torch.equal(ref, mine), (ref - mine).abs().max()
    assert torch.allclose(ref[0], torch.tensor([20., 15., 10., 20., 0.]))
    out["poly_in"], out["poly_xywha"] = polys, ref.numpy()
    angles = np.array([0.0, 0.4, 0.999, 1.0, 89.5, 90.0, 90.5, 135.25, 179.0, 179.999, 180.0], np.float32)   # angle_deg + 90 in [0, 180]
    gl = np.stack([BD.gaussian_label(torch.tensor(a), 180, u=0, sig=6) for a in angles])
    for a, row in zip(angles, gl):
        assert np.array_equal(row, ref_data.gaussian_label(torch.tensor(a), 180, u=0, sig=6))
    out["csl_angle"], out["csl_rows"] = angles, gl.astype(np.float32)

    # ---- __getitem__ tail + collate_fn, for real ------------------------------------------------------------------------------
    class Synth(BD.BaseDataset):
        def __init__(self, S, csl, augment, imgs, tgs, hyp):
            super().__init__(hyp, S, augment, csl, False)
            self.img_files = [f"img{i}" for i in range(len(imgs))]
            self._imgs, self._tgs = imgs, tgs

        def load_image(self, index):
            im = self._imgs[index].copy()
            return im, im.shape[:2], im.shape[:2]

        def load_target(self, index, pad, img_size0, img_size, boarder=None):
            t = self._tgs[index].clone()
            t[:, [2, 4, 6, 8]] += pad[1]
            t[:, [3, 5, 7, 9]] += pad[0]
            return t

    for tag, S, csl, augment, seed in (("a", 64, False, False, 1), ("b", 96, True, True, 2), ("c", 52, False, True, 3), ("d", 64, True, True, 4)):
        g = np.random.default_rng(seed)
        B = 5
        imgs = [g.integers(0, 256, (S, S, 3), dtype=np.uint8) for _ in range(B)]
        tgs = []
        for b in range(B):
            n = 0 if (b == 3 and tag != "a") else int(g.integers(1, 12))
            t = torch.zeros((n, 10))
            t[:, 1] = torch.from_numpy(g.integers(0, 16, n).astype(np.float32))
            t[:, 2:] = torch.from_numpy(synth_polys(g, n, S, edge=True))
            tgs.append(t)
        hyp = {"mosaic": 0.0, "mixup": 0.0, "rotate": 0.0, "scale": 0.0, "translate": 0.0, "fliplr": 0.5, "flipud": 0.5,
               "hsv_h": 0.0, "hsv_s": 0.0, "hsv_v": 0.0}
        calls = []
        BD.random_warping = lambda img, targets, *a, **k: (img, targets)          # identity recorder (needs cv2.warpPerspective)
        BD.horizontal_flip = lambda im, t: (calls.append("lr"), A.horizontal_flip(im, t))[1]
        BD.vertical_flip = lambda im, t: (calls.append("ud"), A.vertical_flip(im, t))[1]
        ds = Synth(S, csl, augment, imgs, tgs, hyp)
        random.seed(seed)
        np.random.seed(seed)
        samples, flags = [], []
        for b in range(B):
            del calls[:]
            samples.append(ds[b])
            flags.append((1 if "lr" in calls else 0) | (2 if "ud" in calls else 0))
        paths, bimgs, btg = ds.collate_fn(samples)
        mine = ref_data.collate([ref_data.finalize_sample(imgs[b], tgs[b], flags[b] & 1, flags[b] & 2, csl) for b in range(B)])
        assert torch.equal(mine[0], bimgs) and torch.equal(mine[1], btg), tag
        out[f"{tag}_imgs_u8"] = np.stack(imgs)
        out[f"{tag}_targets10"] = np.concatenate([np.concatenate([np.full((len(t), 1), b, np.float32), t.numpy()[:, 1:]], 1) for b, t in enumerate(tgs)])
        out[f"{tag}_flags"] = np.array(flags, np.uint8)
        out[f"{tag}_csl"] = np.array(int(csl))
        out[f"{tag}_out_imgs_sum"] = bimgs.double().sum(dim=(2, 3)).numpy()       # [B, 3] checksums (the tensor itself is recomputable: /255)
        out[f"{tag}_out_imgs_sample"] = bimgs[:, :, ::7, ::5].numpy()
        out[f"{tag}_out_targets"] = btg.numpy()
        print(tag, "flags", flags, "targets", tuple(btg.shape))

    # ---- detect path: rescale_boxes + xywha2xyxyxyxy ----------------------------------------------------------------------
    g = np.random.default_rng(7)
    for tag, dim, shape in (("sq", 608, (608, 608)), ("wide", 608, (600, 1000)), ("tall", 416, (1333, 800)), ("odd", 800, (1023, 1024))):
        n = 40
        d = np.zeros((n, 7), np.float32)
        d[:, 0:2] = g.uniform(20, dim - 20, (n, 2))
        d[:, 2] = g.uniform(4, 60, n)
        d[:, 3] = d[:, 2] * g.uniform(1, 5, n)
        d[:, 4] = g.uniform(-np.pi / 2, np.pi / 2, n)
        d[:, 5] = g.uniform(0.1, 1, n)
        d[:, 6] = g.integers(0, 16, n)
        d[0, 4], d[1, 4] = 0.0, -np.pi / 2
        dt = torch.from_numpy(d.copy())
        rb = P.rescale_boxes(dt, dim, shape)
        polys = G.xywha2xyxyxyxy(rb[:, :5])
        mb = ref_data.rescale_boxes(torch.from_numpy(d.copy()), dim, shape)
        assert torch.equal(mb, rb) and torch.equal(ref_data.xywha2xyxyxyxy(mb[:, :5]), polys), tag
        out[f"det_{tag}_in"], out[f"det_{tag}_dim"], out[f"det_{tag}_shape"] = d, np.array(dim), np.array(shape)
        out[f"det_{tag}_boxes"], out[f"det_{tag}_polys"] = rb.numpy(), polys.numpy()
    path = os.path.join(ROOT, "tests", "golden", "g9_data.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
