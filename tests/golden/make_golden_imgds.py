"""Fixture G14: the reference's detect-path dataset — ImageDataset (datasets/base_dataset.py:59-81) imported from /root/reference and
RUN on a temporary folder of images through torch.utils.data.DataLoader(dataset, batch_size, shuffle=False) exactly as detect.py:43-44
builds it.  Only cv2 is replaced (absent in this image): imread serves in-memory arrays, resize / copyMakeBorder are answered by
oracle/ref_data.py's numpy restatement of OpenCV (make_golden_pipeline.install_cv2) — so file order (sorted glob), the letterbox
arithmetic of pad_to_square (rounding of the new size and of the four borders), BGR -> RGB, / 255 and the batch stacking are the
reference's; the INTER_LINEAR pixel values are "parity unpinned" against OpenCV itself.
Stored: the source images (wide, tall, square, already-at-size, odd sizes, one grey image stored as 3 equal channels — cv2.imread's
default flag returns 3 channels), and per img_size the stacked batches as uint8 (the reference's float is exactly uint8 / 255).
Run here:  python tests/golden/make_golden_imgds.py"""
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from tests.golden import make_golden as MG  # noqa: E402
from tests.golden import make_golden_pipeline as MP  # noqa: E402

SHAPES = [(40, 64), (64, 48), (32, 32), (24, 30), (48, 48), (50, 37), (33, 61), (96, 20), (16, 16), (29, 64), (64, 31)]
SIZES = [32, 48]
BATCH = 4


def main():
    MG._install_stubs()
    rng = np.random.RandomState(14)
    images = [rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in SHAPES]
    images[3][:] = images[3][:, :, :1]                          # a grey photograph as cv2.imread returns it
    tmp = tempfile.mkdtemp()
    by_path = {}
    for i, im in enumerate(images):
        p = os.path.join(tmp, "%03d.png" % i)
        open(p, "wb").close()
        by_path[p] = im
    open(os.path.join(tmp, "notes.txt"), "w").close()            # not an image: the ext filter must skip it
    MP.install_cv2(by_path)
    os.chdir(MG.REF)
    sys.path.insert(0, MG.REF)
    pkg = types.ModuleType("refdatasets")                        # the reference's `datasets` package under another name (HF `datasets` clash)
    pkg.__path__ = [os.path.join(MG.REF, "datasets")]
    sys.modules["refdatasets"] = pkg
    base = importlib.import_module("refdatasets.base_dataset")
    g = {"nimg": np.array(len(images)), "sizes": np.array(SIZES), "batch": np.array(BATCH)}
    for i, im in enumerate(images):
        g[f"img{i}"] = im
    for size in SIZES:
        ds = base.ImageDataset(tmp, img_size=size, ext="png")
        assert len(ds) == len(images)
        loader = torch.utils.data.DataLoader(ds, batch_size=BATCH, shuffle=False)
        order, outs = [], []
        for paths, imgs in loader:
            order += [int(os.path.basename(p)[:3]) for p in paths]
            u8 = torch.round(imgs * 255)
            assert torch.equal(u8 / 255, imgs) and tuple(imgs.shape[1:]) == (3, size, size)
            outs.append(u8.to(torch.uint8).numpy())
        g[f"order_{size}"] = np.array(order)
        g[f"imgs_{size}"] = np.concatenate(outs, 0)
        print("G14", size, g[f"imgs_{size}"].shape, order)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g14_imgds.npz"), **g)


if __name__ == "__main__":
    main()
