"""GPU parity tests (through the C ABI): rotated NMS / pairwise IoU / decode / post_process vs the oracle and the
golden fixtures.  Integer outputs (keep indices) are compared bit-exactly; floats to the tolerance written per test."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import ref_ops
from ryolov4_amd.synth import CFG, synth_nms_boxes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _general():
    from ryolov4_amd.lib import general
    return general


@pytest.mark.parametrize("dist", ["U", "C"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 5000])
@pytest.mark.parametrize("thr", [0.65, 0.2])
def test_nms_keep_set_bit_exact(dist, n, thr):
    g = _general()
    b, s = synth_nms_boxes(n, dist, seed=n)
    perm = np.random.RandomState(0).permutation(n)            # un-sorted input: the op sorts by score itself
    bb, ss = b[perm], s[perm]
    for gt in (True, False):
        got = g.nms_rotated(torch.from_numpy(bb).to(DEV), torch.from_numpy(ss).to(DEV), thr, gt_only=gt).cpu().numpy()
        exp = oracle.nms_rotated(bb, ss, thr, gt)
        assert got.dtype == np.int64 and np.array_equal(got, exp), (dist, n, thr, gt, len(got), len(exp))


def test_nms_10k_metric_size_and_ties_and_edges():
    g = _general()
    b, s = synth_nms_boxes(10000, "C", seed=0)
    got = g.nms_rotated(torch.from_numpy(b).to(DEV), torch.from_numpy(s).to(DEV), 0.65).cpu().numpy()
    assert np.array_equal(got, oracle.nms_rotated(b, s, 0.65))
    # score ties -> ascending index; identical boxes; zero-area boxes; no-prune path (thr == 0)
    b2 = b[:300].copy(); s2 = np.full(300, 0.5, np.float32); b2[10] = b2[3]; b2[20, 2] = 0.0
    for thr, gt in ((0.5, True), (0.0, True), (0.0, False), (1e-7, True)):
        got = g.nms_rotated(torch.from_numpy(b2).to(DEV), torch.from_numpy(s2).to(DEV), thr, gt_only=gt).cpu().numpy()
        assert np.array_equal(got, oracle.nms_rotated(b2, s2, thr, gt)), (thr, gt)
    assert g.nms_rotated(torch.zeros(0, 5, device=DEV), torch.zeros(0, device=DEV), 0.5).shape == (0,)


def test_nms_full_size_properties_50k():
    """size-independent properties at the C5 size (oracle too slow): keep is sorted by score, idempotent, and no two
    kept boxes overlap above the threshold (checked on a sample through the pairwise kernel)."""
    g = _general()
    b, s = synth_nms_boxes(50000, "U", seed=9)
    tb, ts = torch.from_numpy(b).to(DEV), torch.from_numpy(s).to(DEV)
    keep = g.nms_rotated(tb, ts, 0.3)
    ks = ts[keep]
    assert torch.all(ks[:-1] >= ks[1:])
    keep2 = g.nms_rotated(tb[keep], ts[keep], 0.3)
    assert torch.equal(keep2, torch.arange(keep.numel(), device=DEV))                 # idempotent
    sub = tb[keep[:3000]]
    iou = g.pairwise_iou_rotated(sub, sub)
    iou.fill_diagonal_(0)
    assert float(iou.max()) <= 0.3


@pytest.mark.parametrize("dist", ["C", "U"])
@pytest.mark.parametrize("thr", [0.2, 0.65])
def test_nms_50k_keep_set_bit_exact_vs_oracle_fixture(golden_dir, dist, thr):
    """BASELINE config C5's candidate count: 50 000 boxes, both synthetic sets, both thresholds — keep set bit-exact against fixture G12,
    generated once by the C oracle (tests/golden/make_golden_nms50k.py: 5 ... 60 s per case on one core); the clustered set is also
    checked live against the oracle (seconds)."""
    g = _general()
    fix = np.load(os.path.join(golden_dir, "g12_nms50k.npz"))
    b, s = synth_nms_boxes(50000, dist, seed=9)
    got = g.nms_rotated(torch.from_numpy(b).to(DEV), torch.from_numpy(s).to(DEV), thr).cpu().numpy()
    want = fix[f"{dist}_{thr}"].astype(np.int64)
    assert got.shape == want.shape and np.array_equal(got, want), (dist, thr, len(got), len(want))
    if dist == "C":
        assert np.array_equal(got, oracle.nms_rotated(b, s, thr))


def test_pairwise_and_diag_iou():
    g = _general()
    b, _ = synth_nms_boxes(700, "C", seed=4)
    b1, b2 = b[:333], b[200:700]
    got = g.pairwise_iou_rotated(torch.from_numpy(b1).to(DEV), torch.from_numpy(b2).to(DEV)).cpu().numpy()
    exp = oracle.pairwise_iou_rotated(b1, b2)
    np.testing.assert_allclose(got, exp, atol=1e-6, rtol=0)                             # tolerance: 1e-6 absolute
    d = g.diag_iou_rotated(torch.from_numpy(b1).to(DEV), torch.from_numpy(b[100:433]).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(d, np.array([oracle.single_iou(x, y) for x, y in zip(b1, b[100:433])], np.float32), atol=1e-6)
    assert g.pairwise_iou_rotated(torch.zeros(0, 5, device=DEV), torch.from_numpy(b2).to(DEV)).shape == (0, 500)


@pytest.mark.parametrize("mode,nc", [("csl", 2), ("csl", 16), ("kfiou", 2), ("kfiou", 16)])
def test_decode_golden(golden_dir, mode, nc):
    from ryolov4_amd.model import yololayer
    gd = np.load(os.path.join(golden_dir, "g3_decode.npz"))
    tag = f"{mode}_nc{nc}"
    logits = [torch.from_numpy(gd[f"{tag}_logits{k}"].astype(np.float32)).to(DEV) for k in range(3)]
    layer = yololayer.make_layer(mode, nc, ref_ops.make_anchors(CFG, mode), [8, 16, 32])
    outs, infer = layer([l.clone() for l in logits], False)
    exp_outs, _ = ref_ops.decode([l.cpu() for l in logits], ref_ops.make_anchors(CFG, mode), nc, mode)
    for a, b in zip(outs, exp_outs):
        assert torch.equal(a.cpu(), b)                                                   # permute is exact
    # tolerance 1e-3 on boxes (north_star); observed ~1e-6 relative (expf ulp differences)
    np.testing.assert_allclose(infer.cpu().numpy(), gd[f"{tag}_infer"], rtol=2e-5, atol=2e-5)
    train_only = layer([l.clone() for l in logits], True)
    assert isinstance(train_only, list) and len(train_only) == 3


@pytest.mark.parametrize("nc,B,img", [(1, 3, 32), (7, 2, 96), (80, 1, 160), (16, 5, 64)])
def test_decode_kfiou_odd_shapes_vs_oracle(nc, B, img):
    """The kfiou decode runs one lane per ELEMENT of the contiguous [rows][attrs] array (r05) and decomposes rows inside a 2 048-element
    workgroup block by small divisions: grids of 1, 2 and 4 cells per side (several image / anchor wraps inside one block), odd row lengths
    (7, 13, 86 floats), blocks that end inside a row and a ragged last block."""
    from ryolov4_amd.model import yololayer
    anchors = ref_ops.make_anchors(CFG, "kfiou")
    g = torch.Generator().manual_seed(nc * 10 + B)
    logits = [torch.randn(B, len(anchors[k]) * (nc + 6), img // st, img // st, generator=g) * 2 for k, st in enumerate((8, 16, 32))]
    layer = yololayer.make_layer("kfiou", nc, anchors, [8, 16, 32])
    _, infer = layer([l.clone().to(DEV) for l in logits], False)
    _, exp = ref_ops.decode(logits, anchors, nc, "kfiou")
    assert infer.shape == exp.shape
    np.testing.assert_allclose(infer.cpu().numpy(), exp.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("nc,M", [(1, 777), (80, 1000), (200, 300), (16, 257)])
def test_post_process_score_pass_row_blocks(nc, M):
    """post_process's score pass stages R rows per workgroup in LDS (R = min(256, 48 KiB / row)): class counts for which R is 256, 141 and 59,
    row counts that are not multiples of R; same detections as the oracle, and the in-place cls *= obj of lib/general.py:155 kept."""
    g = _general()
    gen = torch.Generator().manual_seed(nc + M)
    pred = torch.rand(2, M, nc + 6, generator=gen)
    pred[..., 0:2] *= 300; pred[..., 2] = pred[..., 2] * 40 + 4; pred[..., 3] = pred[..., 2] * 3; pred[..., 4] = (pred[..., 4] - 0.5) * 3.1
    ref = pred.clone()
    exp = ref_ops.post_process(ref, 0.4, 0.3)
    dev = pred.clone().to(DEV)
    got = g.post_process(dev, 0.4, 0.3)
    torch.testing.assert_close(dev.cpu(), ref, rtol=0, atol=0)                            # both mutate their input the same way
    for a, b in zip(exp, got):
        assert a.shape == b.shape
        np.testing.assert_allclose(b.cpu().numpy(), a.numpy(), atol=1e-5)


def test_post_process_golden(golden_dir):
    g = _general()
    gd = np.load(os.path.join(golden_dir, "g7_postprocess.npz"))
    for case in range(4):
        nc, ct, it = gd[f"c{case}_cfg"]
        pred = torch.from_numpy(gd[f"c{case}_pred"].copy()).to(DEV)
        outs = g.post_process(pred, float(ct), float(it))
        ref_sum = float(gd[f"c{case}_mutated_sum"])
        assert abs(pred.double().sum().item() - ref_sum) < 1e-6 * abs(ref_sum)            # in-place cls *= obj kept
        assert len(outs) == pred.shape[0]
        for b, o in enumerate(outs):
            exp = gd[f"c{case}_out{b}"]
            assert tuple(o.shape) == exp.shape, (case, b, o.shape, exp.shape)
            np.testing.assert_allclose(o.cpu().numpy(), exp, atol=1e-6)                   # same rows, same order


def test_post_process_random_vs_oracle_and_empty():
    g = _general()
    gen = torch.Generator().manual_seed(5)
    pred = torch.rand(3, 20000, 22, generator=gen)
    pred[..., 0:2] *= 500; pred[..., 2] = pred[..., 2] * 40 + 4; pred[..., 3] = pred[..., 2] * 3; pred[..., 4] = (pred[..., 4] - 0.5) * 3.1
    pred[1, :, 5] = 0.0                                                                    # image 1: nothing passes
    exp = ref_ops.post_process(pred.clone(), 0.5, 0.3)
    got = g.post_process(pred.clone().to(DEV), 0.5, 0.3)
    assert got[1].shape == (0, 7)
    for a, b in zip(exp, got):
        assert a.shape == b.shape
        np.testing.assert_allclose(b.cpu().numpy(), a.numpy(), atol=1e-5)


@pytest.mark.parametrize("B,M,K", [(1, 1, 1), (2, 63, 63), (3, 5000, 5000), (2, 39375, 5000), (2, 387072, 5000), (1, 70000, 16384), (2, 9000, 100)])
@pytest.mark.parametrize("ninf_frac", [0.0, 0.7, 1.0])
def test_topk_desc_equals_stable_sort(B, M, K, ninf_frac):
    """csrc/topk.hip (radix select + LDS bitonic sort) against torch's STABLE descending sort: the K best of every row in (key desc,
    index asc) order, bit for bit, with heavy ties (keys quantised to 1/64), -0.0 / +0.0, and -inf rows (never selected: -inf / -1
    padding).  Sizes: the reference cap 5000 (lib/general.py:148) out of 39 375 (C4) and 387 072 (C5) candidates."""
    from ryolov4_amd.lib import general
    g = torch.Generator().manual_seed(B * 131 + M + K)
    key = (torch.randint(-8, 64, (B, M), generator=g).float() / 64.0)
    key[key == 0] = -0.0
    key[torch.rand(B, M, generator=g) < 0.01] = 0.0
    key[torch.rand(B, M, generator=g) < ninf_frac] = float("-inf")
    key = key.cuda()
    skey, order = general.topk_desc(key, K)
    ref_k, ref_o = torch.sort(key, dim=1, descending=True, stable=True)
    ref_k, ref_o = ref_k[:, :K], ref_o[:, :K]
    sel = ref_k > float("-inf")
    assert torch.equal(skey[sel], ref_k[sel]) and torch.equal(order[sel], ref_o[sel])
    assert bool((skey[~sel] == float("-inf")).all()) and bool((order[~sel] == -1).all())


@pytest.mark.parametrize("N", [1, 2, 100, 4097, 10000, 16384, 16385, 50000, 200000])
def test_argsort_desc_equals_stable_sort(N):
    """Full descending argsort (nms_rotated's score sort): LDS sort up to 16 384, global bitonic merge steps beyond."""
    from ryolov4_amd.lib import general
    g = torch.Generator().manual_seed(N)
    sc = (torch.randint(0, max(2, N // 3), (N,), generator=g).float() / 7.0).cuda()          # every value ~3 times: ties matter
    got = general.argsort_desc(sc)
    ref = torch.sort(sc, descending=True, stable=True)[1]
    assert torch.equal(got, ref)
