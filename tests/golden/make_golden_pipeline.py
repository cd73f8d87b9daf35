"""Fixture G13: whole BATCHES out of the reference's loader — the real DOTADataset / UCASAODDataset (label files parsed by the reference's
own load_files from a temporary directory), the real BaseDataset.__getitem__ (mosaic-4 / mosaic-9 / mixup / random_warping / hsv /
letterbox / flips, datasets/base_dataset.py:83-157) and collate_fn, imported from /root/reference and RUN with the global `random` /
`numpy.random` seeded per case.  Only cv2 is replaced: imread serves in-memory arrays, every other cv2 function is answered by
oracle/ref_data.py's numpy restatement of OpenCV (resize INTER_LINEAR / INTER_AREA, cvtColor BGR<->HSV, LUT, warpPerspective,
getRotationMatrix2D, copyMakeBorder) — so the ORDER of operations, the random draws, the placement arithmetic and every label value
are the reference's, while the pixel values of the cv2 stages are "parity unpinned" (OpenCV absent, version un-pinned by the reference).
Stored: the source images, the label file texts, per case the configuration + seeds and the batch (imgs as uint8: the reference's
float is exactly uint8 / 255; targets float32).   Run here:  python tests/golden/make_golden_pipeline.py"""
import importlib
import math
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from tests.golden import make_golden as MG  # noqa: E402
from oracle import ref_data  # noqa: E402

CLASSES = ["plane", "small vehicle", "ship"]
HYP = {"hsv_h": 0.015, "hsv_s": 0.7, "hsv_v": 0.4, "rotate": 45, "translate": 0.1, "scale": 0.5, "flipud": 0.5, "fliplr": 0.5, "mosaic": 1.0, "mixup": 0.5}
CASES = [  # name, dataset kind, img_size, augment, csl, hyp overrides, batch indices, seed
    ("dota_mosaic", "DOTA", 32, True, False, {}, [0, 3, 5, 7, 2, 9], 11),
    ("dota_mosaic_csl", "DOTA", 32, True, True, {"mixup": 1.0}, [1, 4, 6, 8], 12),
    ("ucas_plain_aug", "UCAS_AOD", 32, True, False, {"mosaic": 0.0}, [0, 1, 2, 3, 4], 13),
    ("ucas_eval", "UCAS_AOD", 32, False, True, {}, [5, 6, 7, 8, 9, 10], 14),
    ("dota_eval_up", "DOTA", 48, False, False, {}, [0, 2, 11], 15),
]


def install_cv2(images_by_path):
    cv2 = sys.modules["cv2"]
    cv2.INTER_LINEAR, cv2.INTER_AREA, cv2.BORDER_CONSTANT, cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR = 1, 3, 0, 40, 54
    cv2.imread = lambda path: images_by_path[path].copy()

    def resize(img, dsize, interpolation=1):
        return (ref_data.resize_area_numpy if interpolation == cv2.INTER_AREA else ref_data.resize_linear_numpy)(img, dsize)

    def cvt(img, code, dst=None):
        out = ref_data.bgr2hsv_numpy(img) if code == cv2.COLOR_BGR2HSV else ref_data.hsv2bgr_numpy(img)
        if dst is not None:
            dst[...] = out
            return dst
        return out
    cv2.resize, cv2.cvtColor = resize, cvt
    cv2.split = lambda im: [im[..., k] for k in range(im.shape[-1])]
    cv2.merge = lambda chans: np.stack(chans, -1)
    cv2.LUT = lambda ch, lut: lut[ch]
    cv2.copyMakeBorder = lambda img, t, b, l, r, kind, value: np.pad(img, ((t, b), (l, r), (0, 0)), constant_values=value[0])
    cv2.getRotationMatrix2D = lambda angle, center, scale: np.array(
        [[scale * math.cos(angle * math.pi / 180), scale * math.sin(angle * math.pi / 180), 0.0],
         [-scale * math.sin(angle * math.pi / 180), scale * math.cos(angle * math.pi / 180), 0.0]])
    cv2.warpPerspective = lambda img, M, dsize, borderValue: ref_data.warp_perspective_numpy(img, M, dsize, borderValue[0])


def label_text(kind, rng, h, w):
    n = rng.randint(0, 7)
    lines = []
    for _ in range(n):
        c = rng.rand(2) * [w, h]
        d = (rng.rand(4, 2) - 0.5) * [w, h] * 0.35
        pts = (c[None, :] + d).reshape(-1)
        name = CLASSES[rng.randint(0, len(CLASSES))].replace(" ", "-")
        coords = ["%.2f" % v for v in pts]
        lines.append(("\t".join([name] + coords + ["0"]) if kind == "UCAS_AOD" else " ".join(coords + [name, "0"])) + "\n")
    return "".join(lines)


def main():
    MG._install_stubs()
    rng = np.random.RandomState(3)
    shapes = [(40, 64), (64, 48), (32, 32), (24, 30), (64, 64), (50, 37), (33, 61), (48, 20), (16, 16), (29, 64), (64, 31), (12, 20)]
    images = [rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in shapes]
    g = {"nimg": np.array(len(images)), "classes": np.array(CLASSES), "hyp_keys": np.array(sorted(HYP)), "hyp_vals": np.array([HYP[k] for k in sorted(HYP)], np.float64)}
    tmp = tempfile.mkdtemp()
    by_path, texts = {}, {}
    for kind in ("DOTA", "UCAS_AOD"):
        base = os.path.join(tmp, kind)
        os.makedirs(os.path.join(base, "images"), exist_ok=True)
        os.makedirs(os.path.join(base, "annfiles"), exist_ok=True)
        for i, im in enumerate(images):
            ip = os.path.join(base, "images", "%03d.png" % i) if kind == "DOTA" else os.path.join(base, "%03d.png" % i)
            lp = ip.replace("images", "annfiles").replace(".png", ".txt") if kind == "DOTA" else ip.replace(".png", ".txt")
            open(ip, "wb").close()                               # glob finds it; imread is served from memory
            txt = label_text(kind, rng, *im.shape[:2])
            open(lp, "w").write(txt)
            by_path[ip] = im
            texts[(kind, i)] = txt
            g[f"labels_{kind}_{i}"] = np.array(txt)
    for i, im in enumerate(images):
        g[f"img{i}"] = im
    install_cv2(by_path)
    os.chdir(MG.REF)
    sys.path.insert(0, MG.REF)
    pkg = types.ModuleType("refdatasets")                        # the reference's `datasets` package under another name (HF `datasets` clash)
    pkg.__path__ = [os.path.join(MG.REF, "datasets")]
    sys.modules["refdatasets"] = pkg
    dota = importlib.import_module("refdatasets.DOTA_dataset")
    ucas = importlib.import_module("refdatasets.UCASAOD_dataset")
    for name, kind, size, augment, csl, over, indices, seed in CASES:
        hyp = dict(HYP, **over)
        cls = dota.DOTADataset if kind == "DOTA" else ucas.UCASAODDataset
        ds = cls(os.path.join(tmp, kind), CLASSES, hyp, augment, size, csl)
        assert len(ds) == len(images)
        random.seed(seed)
        np.random.seed(seed)
        batch = [ds[i] for i in indices]                        # the reference's __getitem__, sample after sample (num_workers = 0 order)
        paths, imgs, targets = ds.collate_fn(batch)
        u8 = torch.round(imgs * 255)
        assert torch.equal(u8 / 255, imgs)
        g[f"{name}_cfg"] = np.array([kind, str(size), str(int(augment)), str(int(csl)), str(seed)])
        g[f"{name}_hyp"] = np.array([hyp[k] for k in sorted(HYP)], np.float64)
        g[f"{name}_indices"] = np.array(indices)
        g[f"{name}_imgs_u8"] = u8.to(torch.uint8).numpy()
        g[f"{name}_targets"] = targets.numpy()
        print("G13", name, tuple(imgs.shape), tuple(targets.shape))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g13_pipeline.npz"), **g)


if __name__ == "__main__":
    main()
