"""GPU parity of the data-side kernels (csrc/dataprep.hip, through the C ABI) against the oracle and fixture g9_data.npz
(captured from the imported reference).

Tolerances.  ryolo_to_tensor is byte movement + one IEEE division: BIT-EXACT.  Label geometry is fp32 with atan2f / sqrtf on
the device vs torch-CPU's: rows, order, sample index, class and the CSL bin are exact on the fixtures; x, y, w, h within 1e-6
relative, theta within 2e-6 rad.  Polygons of the detect path: 2e-3 px at coordinates up to 1.3e3 (fp32 rotation about the box
centre: the translation column cancels large terms)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_data
from tests.test_data_oracle import split_targets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "g9_data.npz"))


def test_finalize_batch_vs_reference_fixture(golden_dir):
    from ryolov4_amd.datasets.base_dataset import finalize_batch
    g = _g(golden_dir)
    for tag in "abcd":
        imgs, flags, csl = g[f"{tag}_imgs_u8"], g[f"{tag}_flags"], bool(g[f"{tag}_csl"])
        bi, bt = finalize_batch(torch.from_numpy(imgs).to(DEV), torch.from_numpy(g[f"{tag}_targets10"]).to(DEV), torch.from_numpy(flags), csl)
        exp = g[f"{tag}_out_targets"]
        got = bt.cpu().numpy()
        assert got.shape == exp.shape, (tag, got.shape, exp.shape)
        assert np.array_equal(got[:, :2], exp[:, :2]), tag                              # sample index + class, row order
        np.testing.assert_allclose(got[:, 2:6], exp[:, 2:6], rtol=1e-6, atol=1e-7, err_msg=tag)
        np.testing.assert_allclose(got[:, 6], exp[:, 6], rtol=0, atol=2e-6, err_msg=tag)
        if csl:
            assert np.array_equal(got[:, 7:], exp[:, 7:]), tag                          # gaussian rows: values and bin shift exact
        bi = bi.cpu()
        assert np.array_equal(bi[:, :, ::7, ::5].numpy(), g[f"{tag}_out_imgs_sample"]), tag
        assert np.allclose(bi.double().sum(dim=(2, 3)).numpy(), g[f"{tag}_out_imgs_sum"], rtol=0, atol=1e-9), tag


@pytest.mark.parametrize("B,H,W", [(3, 64, 64), (2, 37, 53), (1, 5, 3), (4, 800, 800)])
def test_to_tensor_bit_exact_all_flip_combinations(B, H, W):
    from ryolov4_amd.datasets.base_dataset import finalize_batch
    g = np.random.default_rng(B * 1000 + W)
    imgs = g.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    flags = np.array([(b + 1) % 4 for b in range(B)], np.uint8)
    bi, bt = finalize_batch(torch.from_numpy(imgs).to(DEV), torch.zeros((0, 10)), torch.from_numpy(flags), False)
    assert bt.shape == (0, 7)
    for b in range(B):
        exp, _ = ref_data.finalize_sample(imgs[b], torch.zeros((0, 10)), flags[b] & 1, flags[b] & 2, False)
        assert torch.equal(bi[b].cpu(), exp), (b, flags[b])


@pytest.mark.parametrize("csl", [False, True])
def test_encode_labels_many_targets_order_and_filter(csl):
    """3000 polygons over 8 images (three passes of the 1024-thread compaction), a third of them with the vertex mean outside."""
    from ryolov4_amd.datasets.base_dataset import finalize_batch
    from tests.golden.make_golden_data import synth_polys
    S, B, n = 128, 8, 3000
    g = np.random.default_rng(5)
    t = np.zeros((n, 10), np.float32)
    t[:, 0] = np.sort(g.integers(0, B, n))
    t[:, 1] = g.integers(0, 16, n)
    t[:, 2:] = synth_polys(g, n, S, edge=True)
    t[::3, 2::2] += S                                                     # pushed out of the image -> filtered
    flags = g.integers(0, 4, B).astype(np.uint8)
    imgs = torch.zeros((B, S, S, 3), dtype=torch.uint8, device=DEV)
    _, bt = finalize_batch(imgs, torch.from_numpy(t).to(DEV), torch.from_numpy(flags), csl)
    tgs = split_targets(t, B)
    _, exp = ref_data.collate([ref_data.finalize_sample(np.zeros((S, S, 3), np.uint8), tgs[b], flags[b] & 1, flags[b] & 2, csl) for b in range(B)])
    got, exp = bt.cpu().numpy(), exp.numpy()
    assert got.shape == exp.shape and got.shape[0] < n
    assert np.array_equal(got[:, :2], exp[:, :2])
    np.testing.assert_allclose(got[:, 2:6], exp[:, 2:6], rtol=1e-6, atol=1e-7)
    dth = np.abs(got[:, 6] - exp[:, 6])
    assert (np.minimum(dth, np.abs(dth - np.pi)) < 2e-6).all()            # theta within 2e-6 (a +-pi/2 edge may wrap)
    if csl:
        same = (got[:, 7:] == exp[:, 7:]).all(axis=1)
        assert same.mean() > 0.995                                        # a 1-ulp theta difference may move a bin boundary: < 0.5 % of rows
        assert np.array_equal(np.sort(got[~same, 7:], axis=1), np.sort(exp[~same, 7:], axis=1))   # ... and then only by a rotation


def test_polygon_converters_and_detect_path(golden_dir):
    from ryolov4_amd.lib.general import xyxyxyxy2xywha, xywha2xyxyxyxy
    from ryolov4_amd.lib.plot import detections_to_polys, rescale_boxes
    g = _g(golden_dir)
    got = xyxyxyxy2xywha(torch.from_numpy(g["poly_in"]).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(got[:, :4], g["poly_xywha"][:, :4], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(got[:, 4], g["poly_xywha"][:, 4], rtol=0, atol=2e-6)
    dets, shapes, dims = [], [], set()
    for tag in ("sq", "wide", "tall", "odd"):
        d = torch.from_numpy(g[f"det_{tag}_in"].copy()).to(DEV)
        dim, shape = int(g[f"det_{tag}_dim"]), tuple(int(v) for v in g[f"det_{tag}_shape"])
        polys0 = xywha2xyxyxyxy(d[:, :5])                                            # before the rescale: vs the oracle on the raw boxes
        np.testing.assert_allclose(polys0.cpu().numpy(), ref_data.xywha2xyxyxyxy(torch.from_numpy(g[f"det_{tag}_in"][:, :5])).numpy(), rtol=0, atol=2e-3)
        r = rescale_boxes(d, dim, shape)
        assert r.data_ptr() == d.data_ptr()                                          # in place, like lib/plot.py:19-30
        np.testing.assert_allclose(d.cpu().numpy(), g[f"det_{tag}_boxes"], rtol=2e-6, atol=1e-4, err_msg=tag)
        np.testing.assert_allclose(xywha2xyxyxyxy(d[:, :5]).cpu().numpy(), g[f"det_{tag}_polys"], rtol=0, atol=2e-3, err_msg=tag)
        if dim == 608:
            dets.append(torch.from_numpy(g[f"det_{tag}_in"].copy()).to(DEV)); shapes.append(shape)
    dets.insert(1, torch.zeros((0, 7), device=DEV)); shapes.insert(1, (100, 100))     # an image without detections
    boxes, polys, counts = detections_to_polys(dets, 608, shapes)
    assert counts == [40, 0, 40]
    exp_b = np.concatenate([g["det_sq_boxes"], g["det_wide_boxes"]]); exp_p = np.concatenate([g["det_sq_polys"], g["det_wide_polys"]])
    np.testing.assert_allclose(boxes.cpu().numpy(), exp_b, rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(polys.cpu().numpy(), exp_p, rtol=0, atol=2e-3)


def test_detect_path_end_to_end_against_the_oracle_chain():
    """detect.py:57-61 + Detect.save_results / plot_boxes up to the drawing: network (eval) -> post_process -> rescale_boxes ->
    xywha2xyxyxyxy, on the device, against the oracle chain fed with the SAME inference tensor (ref_ops.post_process with the C NMS,
    ref_data.rescale_boxes, ref_data.xywha2xyxyxyxy): identical detection sets, polygons within 2e-3 px."""
    from oracle import ref_ops
    from ryolov4_amd.lib.general import post_process
    from ryolov4_amd.lib.plot import detections_to_polys
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, fill_state
    m = Yolo(2, CFG, "kfiou", "yolov7")
    m.load_state_dict(fill_state(m.state_dict()))
    m.to(DEV).eval()
    x = torch.rand(3, 3, 160, 160, generator=torch.Generator().manual_seed(8)).to(DEV)
    with torch.no_grad():
        _, inf = m(x, training=False)
    inf_cpu = inf.cpu().clone()
    dets = post_process(inf, 0.3, 0.4)
    shapes = [(480, 640), (600, 600), (1000, 750)]                       # original image sizes of the three letterboxed inputs
    boxes, polys, counts = detections_to_polys(dets, 160, shapes)
    exp = ref_ops.post_process(inf_cpu, 0.3, 0.4)
    assert counts == [int(e.shape[0]) for e in exp] and sum(counts) > 0
    ofs = 0
    for b, e in enumerate(exp):
        n = e.shape[0]
        np.testing.assert_allclose(dets[b].cpu().numpy(), e.numpy(), rtol=1e-5, atol=1e-5)
        rb = ref_data.rescale_boxes(e.clone(), 160, shapes[b])
        np.testing.assert_allclose(boxes[ofs:ofs + n].cpu().numpy(), rb.numpy(), rtol=2e-6, atol=1e-3)
        if n:
            np.testing.assert_allclose(polys[ofs:ofs + n].cpu().numpy(), ref_data.xywha2xyxyxyxy(rb[:, :5]).numpy(), rtol=0, atol=2e-3)
        ofs += n


@pytest.mark.parametrize("mode", ["kfiou", "csl"])
def test_finalize_batch_feeds_a_training_step(mode):
    """The data-side tail hands train.py:183-198 what the reference's collate_fn does: images [B,3,S,S] fp32 in [0,1] and targets
    [n, 7 | 187] whose first column is the sample index — one training step on them runs and yields finite loss and gradients."""
    from ryolov4_amd.datasets.base_dataset import finalize_batch
    from ryolov4_amd.lib.loss import ComputeCSLLoss, ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, fill_state
    from tests.golden.make_golden_data import synth_polys
    B, S, nc = 4, 128, 2
    g = np.random.default_rng(3)
    imgs = torch.from_numpy(g.integers(0, 256, (B, S, S, 3), dtype=np.uint8)).to(DEV)
    t = np.zeros((B * 6, 10), np.float32)
    t[:, 0] = np.repeat(np.arange(B), 6)
    t[:, 1] = g.integers(0, nc, B * 6)
    t[:, 2:] = synth_polys(g, B * 6, S)
    x, tg = finalize_batch(imgs, torch.from_numpy(t).to(DEV), torch.tensor([0, 1, 2, 3], dtype=torch.uint8), mode == "csl")
    assert x.shape == (B, 3, S, S) and float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    assert tg.shape[1] == (187 if mode == "csl" else 7) and set(tg[:, 0].cpu().tolist()) <= set(range(B))
    assert bool((tg[:, 5] >= tg[:, 4]).all()) and bool((tg[:, 6] >= -np.pi / 2).all()) and bool((tg[:, 6] < np.pi / 2).all())
    m = Yolo(nc, CFG, mode, "yolov7")
    m.load_state_dict(fill_state(m.state_dict()))
    m.to(DEV).train()
    crit = (ComputeCSLLoss if mode == "csl" else ComputeKFIoULoss)(m, HYP)
    loss, items = crit(m(x, training=True), tg)
    loss.backward()
    assert np.isfinite(items["total_loss"]) and items["total_loss"] > 0
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
