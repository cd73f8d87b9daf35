"""SURVEY §8(f) N4 / VERDICT r3 missing #2: the detect path's ImageDataset (datasets/base_dataset.py:59-81, detect.py:12,43-44) as a
device-side batch call, against fixture G14 — batches the imported reference's ImageDataset produced through torch's DataLoader
(tests/golden/make_golden_imgds.py; cv2 answered by the numpy restatement of OpenCV, so the INTER_LINEAR pixels are unpinned against
OpenCV itself while file order, letterbox arithmetic, BGR -> RGB, / 255 and stacking are the reference's).  Bit-exact."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g14_imgds.npz"))
NIMG, BATCH = int(G["nimg"]), int(G["batch"])


def _folder(tmp_path):
    images = {}
    for i in range(NIMG):
        p = os.path.join(str(tmp_path), "%03d.png" % i)
        open(p, "wb").close()
        images[p] = G[f"img{i}"]
    open(os.path.join(str(tmp_path), "notes.txt"), "w").close()
    return images


def test_file_list_and_constructor(tmp_path):
    from ryolov4_amd.datasets.base_dataset import ImageDataset
    images = _folder(tmp_path)
    ds = ImageDataset(str(tmp_path), img_size=32, ext="png", device="cpu", imread=lambda p: images[p])
    assert len(ds) == NIMG and ds.files == sorted(images) and ds.img_size == 32
    assert len(ImageDataset(str(tmp_path), ext="jpg", device="cpu")) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("size", [int(s) for s in G["sizes"]])
def test_image_dataset_batches_match_the_reference(tmp_path, size):
    from ryolov4_amd.datasets.base_dataset import ImageDataset
    images = _folder(tmp_path)
    ds = ImageDataset(str(tmp_path), img_size=size, ext="png", device="cuda:0", imread=lambda p: images[p])
    want = G[f"imgs_{size}"]
    # (1) the batch call
    got, paths = [], []
    for p, imgs in ds.loader(BATCH):
        assert imgs.is_cuda and imgs.dtype == torch.float32 and tuple(imgs.shape[1:]) == (3, size, size)
        u8 = torch.round(imgs * 255).to(torch.uint8).cpu()
        assert torch.equal(u8.float() / 255, imgs.cpu())                   # exactly uint8 / 255 (base_dataset.py:79), IEEE division
        got.append(u8.numpy())
        paths += list(p)
    got = np.concatenate(got, 0)
    assert paths == ds.files and [int(os.path.basename(p)[:3]) for p in paths] == G[f"order_{size}"].tolist()
    diff = np.abs(got.astype(int) - want.astype(int))
    assert np.array_equal(got, want), (int(diff.max()), int((diff > 0).sum()))
    # (2) detect.py:44's own construction: torch DataLoader over the dataset (its fetcher uses __getitems__ = the batch call)
    loader = torch.utils.data.DataLoader(ds, batch_size=BATCH, shuffle=False)
    got2 = torch.cat([imgs for _, imgs in loader], 0)
    assert got2.is_cuda and np.array_equal(torch.round(got2 * 255).to(torch.uint8).cpu().numpy(), want)
    # (3) __getitem__
    p, img = ds[5]
    assert p == ds.files[5] and np.array_equal(torch.round(img * 255).to(torch.uint8).cpu().numpy(), want[5])
