"""CPU tests: the torch-CPU oracle (oracle/ref_ops.py, oracle/ref_model.py) against the golden vectors captured from
the imported reference (tests/golden/make_golden.py).  This is what 'pins' the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_model, ref_ops
from ryolov4_amd.synth import CFG, HYP, fill_state


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("ver", ["yolov4", "yolov5", "yolov7"])
@pytest.mark.parametrize("mode", ["csl", "kfiou"])
def test_g2_full_network(golden_dir, ver, mode):
    g = _load(golden_dir, "g2_fullnet.npz")
    net = ref_model.Yolo(2, CFG, mode, ver)
    net.load_state_dict(fill_state(net.state_dict()), strict=True)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    for train in (False, True):
        net.train(train)
        with torch.no_grad():
            hm = net.head_maps(x)
        tag = f"{ver}_{mode}_{'train' if train else 'eval'}"
        for k, a in enumerate(hm):
            ref_abs = float(g[f"{tag}_abs{k}"])
            assert abs(a.double().sum().item() - float(g[f"{tag}_sum{k}"])) < 1e-4 * max(1.0, ref_abs)
            samp = a.flatten()[:: max(1, a.numel() // 64)][:64].numpy()
            np.testing.assert_allclose(samp, g[f"{tag}_head{k}_sample"], rtol=1e-4, atol=1e-5)
        if not train:
            _, inf = ref_ops.decode(hm, net.anchors, 2, mode)
            samp = inf.flatten()[:: inf.numel() // 64][:64].numpy()
            np.testing.assert_allclose(samp, g[f"{tag}_infer_sample"], rtol=1e-4, atol=1e-4)


def test_state_dict_entry_counts():
    # v4 648 entries, v5 612, v7 564 (kfiou/csl only change head widths)
    for ver, n in (("yolov4", 648), ("yolov5", 612), ("yolov7", 564)):
        assert len(ref_model.Yolo(2, CFG, "kfiou", ver).state_dict()) == n


@pytest.mark.parametrize("mode,nc", [("csl", 2), ("csl", 16), ("kfiou", 2), ("kfiou", 16)])
def test_g3_decode(golden_dir, mode, nc):
    g = _load(golden_dir, "g3_decode.npz")
    tag = f"{mode}_nc{nc}"
    logits = [torch.from_numpy(g[f"{tag}_logits{k}"].astype(np.float32)) for k in range(3)]
    _, inf = ref_ops.decode(logits, ref_ops.make_anchors(CFG, mode), nc, mode)
    np.testing.assert_allclose(inf.numpy(), g[f"{tag}_infer"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("mode,nc", [("csl", 2), ("csl", 16), ("kfiou", 2), ("kfiou", 16)])
def test_g46_targets_and_loss(golden_dir, mode, nc):
    g = _load(golden_dir, "g46_loss.npz")
    anchors = ref_ops.make_anchors(CFG, mode)
    for case in range(3):
        tag = f"{mode}_nc{nc}_c{case}"
        tg = torch.from_numpy(g[f"{tag}_targets"])
        outs = [torch.from_numpy(g[f"{tag}_out{i}"].astype(np.float32)).requires_grad_() for i in range(3)]
        bt = ref_ops.build_targets([(o.shape[2], o.shape[3]) for o in outs], tg, anchors, mode)
        for i in range(3):
            idx = torch.stack((bt[i]["b"], bt[i]["a"], bt[i]["gj"], bt[i]["gi"], bt[i]["c"]), 1).numpy()
            assert np.array_equal(idx.reshape(-1, 5), g[f"{tag}_idx{i}"].reshape(-1, 5))          # bit-exact indices
            np.testing.assert_allclose(bt[i]["tbox"].numpy().reshape(-1), g[f"{tag}_tbox{i}"].reshape(-1), atol=1e-6)
        loss, items = ref_ops.compute_loss(outs, tg, anchors, nc, mode, HYP)
        names = [str(s) for s in g[f"{tag}_item_names"]]
        for nm, ref in zip(names, g[f"{tag}_items"]):
            assert abs(float(items[nm]) - ref) < 2e-5 * max(1.0, abs(ref)), (tag, nm)
        loss.backward()
        for i in range(3):
            np.testing.assert_allclose(outs[i].grad.numpy(), g[f"{tag}_grad{i}"], rtol=1e-4, atol=1e-7)


def test_g5_box_math(golden_dir):
    g = _load(golden_dir, "g5_boxmath.npz")
    ci = ref_ops.bbox_ciou(torch.from_numpy(g["ciou_p"]), torch.from_numpy(g["ciou_t"]))
    np.testing.assert_allclose(ci.numpy(), g["ciou"], atol=1e-6)
    l, k = ref_ops.kf_loss(torch.from_numpy(g["kf_p"]), torch.from_numpy(g["kf_t"]))
    assert abs(l.item() - float(g["kf_loss"])) < 1e-4 * abs(float(g["kf_loss"]))
    np.testing.assert_allclose(k.numpy(), g["kfiou"], rtol=1e-4, atol=1e-6)
    l1, k1 = ref_ops.kf_loss(torch.from_numpy(g["kf_p"][:1]), torch.from_numpy(g["kf_t"][:1]))
    assert abs(l1.item() - float(g["kf1_loss"])) < 1e-5 and np.allclose(k1.numpy(), g["kfiou1"], atol=1e-6)
    assert np.array_equal(ref_ops.norm_angle(torch.from_numpy(g["ang_in"])).numpy(), g["ang_out"])


def test_g7_post_process(golden_dir):
    g = _load(golden_dir, "g7_postprocess.npz")
    for case in range(4):
        nc, ct, it = g[f"c{case}_cfg"]
        pred = torch.from_numpy(g[f"c{case}_pred"].copy())
        outs = ref_ops.post_process(pred, float(ct), float(it))
        assert abs(pred.double().sum().item() - float(g[f"c{case}_mutated_sum"])) < 1e-6 * abs(float(g[f"c{case}_mutated_sum"]))
        for b, o in enumerate(outs):
            exp = g[f"c{case}_out{b}"]
            assert o.shape == exp.shape
            np.testing.assert_allclose(o.numpy(), exp, atol=1e-6)


@pytest.mark.parametrize("tag", ["csl_nc2", "kfiou_nc2", "kfiou_nc16", "csl_nc16", "csl_nc16_empty"])
def test_g10_focal_loss(golden_dir, tag):
    """Oracle with FocalLoss active + non-unit pos_weights against the fixture the imported reference produced (make_golden_focal.py)."""
    g = _load(golden_dir, "g10_focal.npz")
    mode, nc = tag.split("_")[0], int(tag.split("_")[1][2:])
    hyp = {str(k): float(v) for k, v in zip(g["hyp_keys"], g["hyp_vals"])}
    tg = torch.from_numpy(g[f"{tag}_targets"])
    outs = [torch.from_numpy(g[f"{tag}_out{i}"].astype(np.float32)).requires_grad_() for i in range(3)]
    loss, items = ref_ops.compute_loss(outs, tg, ref_ops.make_anchors(CFG, mode), nc, mode, hyp)
    loss.backward()
    for nm, ref in zip([str(s) for s in g[f"{tag}_item_names"]], g[f"{tag}_items"]):
        assert abs(float(items[nm]) - ref) < 2e-5 * max(1.0, abs(ref)), (tag, nm)
    for i in range(3):
        np.testing.assert_allclose(outs[i].grad.numpy(), g[f"{tag}_grad{i}"], rtol=1e-4, atol=1e-7)


def test_sl1iou_extra_mode_oracle_is_self_consistent():
    """The extra mode's fp64 oracle (no reference code exists for it): value = mean |-log IoU|, identical boxes cost nothing, and the
    autograd gradient has the direction of the smooth-L1 gradient scaled by |-log IoU| / S."""
    p = torch.tensor([[3.0, 4.0, 6.0, 2.0, 0.3], [5.0, 5.0, 4.0, 4.0, -0.2]], dtype=torch.float64, requires_grad=True)
    t = torch.tensor([[3.5, 4.2, 5.0, 2.5, 0.1], [5.0, 5.0, 4.0, 4.0, -0.2]], dtype=torch.float64)
    loss, iou = ref_ops.sl1iou_loss(p, t)
    assert abs(float(iou[1]) - 1.0) < 1e-6 and 0.3 < float(iou[0]) < 0.9
    assert abs(float(loss) - 0.5 * (-np.log(float(iou[0])))) < 1e-9
    loss.backward()
    d = (p.detach() - t)[0]
    S = float((0.5 * d * d).sum())
    assert torch.allclose(p.grad[0], d * (-np.log(float(iou[0]))) / S / 2, rtol=1e-9)
    assert float(p.grad[1].abs().sum()) == 0.0
