"""CPU: the data-side oracle (oracle/ref_data.py) against fixture g9_data.npz, which tests/golden/make_golden_data.py captured by
running the imported reference (BaseDataset.__getitem__ tail + collate_fn, xyxyxyxy2xywha, xywha2xyxyxyxy, rescale_boxes,
gaussian_label).  Bit-exact: same torch-CPU ops in the same order."""
import os

import numpy as np
import torch

from oracle import ref_data


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "g9_data.npz"))


def split_targets(t10, B):
    t = torch.from_numpy(t10)
    return [torch.cat((torch.zeros((int((t[:, 0] == b).sum()), 1)), t[t[:, 0] == b][:, 1:]), 1) for b in range(B)]


def test_poly_to_xywha_and_csl_rows(golden_dir):
    g = _g(golden_dir)
    got = ref_data.xyxyxyxy2xywha(torch.from_numpy(g["poly_in"]))
    assert np.array_equal(got.numpy(), g["poly_xywha"])
    assert np.allclose(got[0].numpy(), [20, 15, 10, 20, 0])                 # the h = long side quirk
    assert (got[:, 3] >= got[:, 2]).all() and (got[:, 4] >= -np.pi / 2).all() and (got[:, 4] < np.pi / 2).all()
    for a, row in zip(g["csl_angle"], g["csl_rows"]):
        assert np.array_equal(ref_data.gaussian_label(torch.tensor(a), 180, u=0, sig=6).astype(np.float32), row)


def test_finalize_and_collate_match_reference(golden_dir):
    g = _g(golden_dir)
    for tag in "abcd":
        imgs, flags, csl = g[f"{tag}_imgs_u8"], g[f"{tag}_flags"], bool(g[f"{tag}_csl"])
        tgs = split_targets(g[f"{tag}_targets10"], len(imgs))
        bi, bt = ref_data.collate([ref_data.finalize_sample(imgs[b], tgs[b], flags[b] & 1, flags[b] & 2, csl) for b in range(len(imgs))])
        assert np.array_equal(bt.numpy(), g[f"{tag}_out_targets"]), tag
        assert np.array_equal(bi[:, :, ::7, ::5].numpy(), g[f"{tag}_out_imgs_sample"]), tag
        assert np.allclose(bi.double().sum(dim=(2, 3)).numpy(), g[f"{tag}_out_imgs_sum"], rtol=0, atol=1e-9), tag


def test_rescale_and_polys_match_reference(golden_dir):
    g = _g(golden_dir)
    for tag in ("sq", "wide", "tall", "odd"):
        b = ref_data.rescale_boxes(torch.from_numpy(g[f"det_{tag}_in"].copy()), int(g[f"det_{tag}_dim"]), tuple(int(v) for v in g[f"det_{tag}_shape"]))
        assert np.array_equal(b.numpy(), g[f"det_{tag}_boxes"]), tag
        assert np.array_equal(ref_data.xywha2xyxyxyxy(b[:, :5]).numpy(), g[f"det_{tag}_polys"]), tag


def test_product_gaussian_label_equals_the_reference_rows(golden_dir):
    """The host helper exported by ryolov4_amd.datasets.base_dataset (np.roll form) against the rows captured from the reference,
    including negative shifts (angles above 90) and the truncation-toward-zero edge cases."""
    from ryolov4_amd.datasets.base_dataset import gaussian_label
    g = _g(golden_dir)
    for a, row in zip(g["csl_angle"], g["csl_rows"]):
        assert np.array_equal(gaussian_label(torch.tensor(a), 180, u=0, sig=6).astype(np.float32), row), a
        assert np.array_equal(gaussian_label(float(a), 180, u=0, sig=6), ref_data.gaussian_label(float(a), 180, u=0, sig=6))
