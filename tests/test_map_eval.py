"""SURVEY.md §8(f) N1 — mAP evaluation (test.py:16-164).  CPU: the oracle restatement and the package's host-side
ap_per_class against the fixture captured from the imported reference (tests/golden/make_golden_map.py).  GPU: the
single-launch matching kernel through the C ABI — true-positive matrices bit-exact, the in-place degrees side effect,
empty images, and the downstream AP numbers."""
import os

import numpy as np
import pytest
import torch

G8 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_map.npz"))
NCASES = 5


def _case(ci):
    nimg = int(G8[f"c{ci}_n"][0])
    outs = [torch.from_numpy(G8[f"c{ci}_out{b}"].copy()) for b in range(nimg)]
    return outs, torch.from_numpy(G8[f"c{ci}_targets"].copy()), nimg


def _check(ci, stats, outs):
    assert len(stats) == int(G8[f"c{ci}_nstats"])
    for k, st in enumerate(stats):
        assert np.array_equal(np.asarray(st[0]).astype(np.uint8), G8[f"c{ci}_tp{k}"]), (ci, k)
        assert np.array_equal(np.asarray(st[1], dtype=np.float32), G8[f"c{ci}_conf{k}"])
        assert np.array_equal(np.asarray(st[2], dtype=np.float32), G8[f"c{ci}_pcls{k}"])
        assert np.array_equal(np.asarray(st[3], dtype=np.float32), G8[f"c{ci}_tcls{k}"])
    for b, o in enumerate(outs):
        assert np.array_equal(o.cpu().numpy(), G8[f"c{ci}_mut{b}"]), "in-place radians -> degrees side effect (test.py:126)"


@pytest.mark.parametrize("ci", range(NCASES))
def test_oracle_matches_reference_fixture(ci):
    from oracle import ref_ops
    outs, targets, _ = _case(ci)
    stats = ref_ops.get_batch_statistics(outs, targets, torch.from_numpy(G8["iouv"]), 10)
    _check(ci, stats, outs)


@pytest.mark.parametrize("ci", range(NCASES))
def test_ap_per_class_host_code(ci):
    if f"c{ci}_ap" not in G8:
        pytest.skip("no true positives in this case")
    from ryolov4_amd.lib import evaluate
    n = int(G8[f"c{ci}_nstats"])
    cat = [np.concatenate([G8[f"c{ci}_{name}{k}"] for k in range(n)], 0) for name in ("tp", "conf", "pcls", "tcls")]
    cat[0] = cat[0].astype(bool)
    p, r, ap, f1, cls = evaluate.ap_per_class(*cat)
    for got, name in ((p, "p"), (r, "r"), (ap, "ap"), (f1, "f1"), (cls, "cls")):
        assert np.array_equal(got, G8[f"c{ci}_{name}"]), name
    res = evaluate.calculate_eval_stats(cat, 17, host=True)
    assert abs(res[-2] - G8[f"c{ci}_ap"][:, 0].mean()) < 1e-12 and abs(res[-1] - G8[f"c{ci}_ap"].mean(1).mean()) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(NCASES))
def test_hip_matching_bit_exact(ci):
    from ryolov4_amd.lib import evaluate
    outs, targets, _ = _case(ci)
    outs = [o.cuda() for o in outs]
    stats = evaluate.get_batch_statistics(outs, targets.cuda(), torch.from_numpy(G8["iouv"]), 10)
    _check(ci, stats, outs)


@pytest.mark.gpu
def test_hip_matching_random_vs_oracle():
    """Larger random batches (up to 300 predictions / 60 labels per image, 16 classes) against the oracle."""
    from oracle import ref_ops
    from ryolov4_amd.lib import evaluate
    from tests.golden.make_golden_map import make_case
    iouv = torch.linspace(0.5, 0.95, 10)
    for seed in (5, 6, 7):
        outs, targets = make_case(seed, 8, 16, empty_pred=(3,), empty_lab=(5,))
        big = []
        for o in outs:                                        # densify: replicate with jitter so classes collide often
            if len(o):
                g = torch.Generator().manual_seed(seed)
                rep = o.repeat(6, 1)
                rep[:, :2] += torch.randn(rep.shape[0], 2, generator=g) * 1.5
                rep[:, 5] = torch.rand(rep.shape[0], generator=g)
                o = rep[torch.argsort(-rep[:, 5], stable=True)]
            big.append(o)
        ref = ref_ops.get_batch_statistics([o.clone() for o in big], targets.clone(), iouv, 10)
        dev_out = [o.clone().cuda() for o in big]
        got = evaluate.get_batch_statistics(dev_out, targets.cuda(), iouv, 10)
        assert len(ref) == len(got)
        for a, b in zip(ref, got):
            assert np.array_equal(np.asarray(a[0]), np.asarray(b[0]))
            assert a[3] == b[3]


def test_no_predictions_and_no_labels_anywhere():
    """Nothing to launch: images without predictions contribute a row only when they have labels (test.py:112-115)."""
    from ryolov4_amd.lib import evaluate
    outs = [torch.zeros((0, 7)), torch.zeros((0, 7))]
    assert evaluate.get_batch_statistics(outs, torch.zeros((0, 7)), torch.linspace(0.5, 0.95, 10), 10) == []
    tg = torch.tensor([[1.0, 3.0, 10.0, 10.0, 4.0, 8.0, 0.1]])
    st = evaluate.get_batch_statistics(outs, tg, torch.linspace(0.5, 0.95, 10), 10)
    assert len(st) == 1 and st[0][0].shape == (0, 10) and st[0][3] == [3.0]


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(NCASES))
def test_device_ap_equals_host_ap_on_the_fixture(ci):
    """ryolo_ap_per_class (sort, curves, envelope, 101-point integral, 1000-point p / r curves on the device) against the fixture values
    the imported reference produced: bit for bit."""
    if f"c{ci}_ap" not in G8:
        pytest.skip("no true positives in this case")
    from ryolov4_amd.lib import evaluate
    n = int(G8[f"c{ci}_nstats"])
    cat = [np.concatenate([G8[f"c{ci}_{name}{k}"] for k in range(n)], 0) for name in ("tp", "conf", "pcls", "tcls")]
    cat[0] = cat[0].astype(bool)
    p, r, ap, f1, cls = evaluate.ap_per_class_device(*cat, num_classes=17)
    for got, name in ((p, "p"), (r, "r"), (ap, "ap"), (f1, "f1"), (cls, "cls")):
        assert np.array_equal(got, G8[f"c{ci}_{name}"]), (name, np.abs(np.asarray(got, dtype=np.float64) - G8[f"c{ci}_{name}"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("n,nc,seed", [(20000, 16, 0), (3000, 3, 1), (1, 2, 2), (1025, 5, 3)])
def test_device_ap_random_vs_host(n, nc, seed):
    """Larger random statistics (several 1024-element scan chunks per class, classes without predictions or labels, distinct confidences
    so that the sort order is defined): device == host numpy path, exactly."""
    from ryolov4_amd.lib import evaluate
    rng = np.random.RandomState(seed)
    conf = rng.permutation(n).astype(np.float32) / np.float32(n) * np.float32(0.98) + np.float32(0.01)      # distinct
    pcls = rng.randint(0, nc, n).astype(np.float32)
    tp = rng.rand(n, 10) < (conf[:, None] * np.linspace(0.9, 0.2, 10)[None])
    tp = np.logical_and.accumulate(tp, axis=1)                         # a TP at a higher threshold is a TP at the lower ones
    tcls = rng.randint(0, nc, max(n // 2, 1)).astype(np.float32)
    if nc > 2:
        pcls[pcls == 1] = 0                                           # class 1: labels but no predictions
        tcls = tcls[tcls != 2]                                        # class 2: predictions but no labels
    hp, hr, hap, hf1, hcls = evaluate.ap_per_class(tp, conf, pcls, tcls)
    dp, dr, dap, df1, dcls = evaluate.ap_per_class_device(tp, conf, pcls, tcls, num_classes=nc)
    assert np.array_equal(hcls, dcls)
    for a, b, name in ((hp, dp, "p"), (hr, dr, "r"), (hap, dap, "ap"), (hf1, df1, "f1")):
        assert np.array_equal(a, b), (name, np.abs(a - b).max())
