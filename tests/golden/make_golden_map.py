#!/usr/bin/env python3
"""Generate tests/golden/g8_map.npz (mAP evaluation, test.py:16-164) by IMPORTING THE REFERENCE's
test.py in the build container, and assert that oracle/ref_ops.py reproduces it on every case.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_map.py
Stubs (absent here): detectron2 (pairwise_iou_rotated answers with the build's C oracle — that third-party boundary is
"parity unpinned", see oracle/rotated_iou.c), cv2, colorlog-backed lib.logger, lib.load (clashes with the installed
HuggingFace `datasets` package).  Only data is committed; the reference never travels.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference"
sys.path.insert(0, ROOT)
import oracle                                   # noqa: E402
from oracle import ref_ops                      # noqa: E402


def _install_stubs():
    mods = {n: types.ModuleType(n) for n in ("detectron2", "detectron2.layers", "detectron2.layers.rotated_boxes", "detectron2.layers.nms",
                                             "cv2", "lib.load", "lib.logger")}
    mods["detectron2.layers.rotated_boxes"].pairwise_iou_rotated = lambda a, b: torch.from_numpy(
        oracle.pairwise_iou_rotated(a.numpy(), b.numpy()))
    mods["detectron2.layers.nms"].nms_rotated = lambda *a: None
    mods["lib.load"].load_data = None
    mods["lib.logger"].logger = None
    sys.modules.update(mods)


def make_case(seed, nimg, nc, empty_pred=(), empty_lab=()):
    g = np.random.default_rng(seed)
    outs, tg = [], []
    for b in range(nimg):
        nl = 0 if b in empty_lab else int(g.integers(1, 13))
        t = np.zeros((nl, 7), np.float32)
        t[:, 0] = b
        t[:, 1] = g.integers(0, nc, nl)
        t[:, 2:4] = g.uniform(40, 216, (nl, 2))
        t[:, 4] = g.uniform(8, 30, nl)
        t[:, 5] = t[:, 4] * g.uniform(1, 4, nl)
        t[:, 6] = g.uniform(-np.pi / 2, np.pi / 2, nl)
        tg.append(t)
        if b in empty_pred:
            outs.append(torch.zeros((0, 7)))
            continue
        rows = []
        for k in range(nl):                                   # 0-3 jittered detections per label (duplicates -> later ones are FP)
            for _ in range(int(g.integers(0, 4))):
                r = t[k, 2:7].copy()
                r[:2] += g.normal(0, 2.0, 2)
                r[2:4] *= g.uniform(0.85, 1.15, 2)
                r[4] += g.normal(0, 0.06)
                cls = t[k, 1] if g.random() < 0.85 else g.integers(0, nc)      # some with the wrong class
                rows.append(np.concatenate([r, [g.uniform(0.05, 1.0), cls]]))
        for _ in range(int(g.integers(0, 6))):                # clutter, possibly of a class with no label in the image
            rows.append(np.array([g.uniform(0, 256), g.uniform(0, 256), g.uniform(8, 30), g.uniform(20, 90), g.uniform(-1.5, 1.5),
                                  g.uniform(0.05, 1.0), g.integers(0, nc + 1)]))
        p = np.array(rows, np.float32).reshape(-1, 7)
        if len(p) > 3:
            p[1, :5] = p[0, :5]                              # an exact duplicate box: IoU ties between predictions
        p = p[np.argsort(-p[:, 5], kind="stable")]           # post_process order: score descending
        outs.append(torch.from_numpy(p))
    return outs, torch.from_numpy(np.concatenate(tg, 0))


def main():
    _install_stubs()
    os.chdir(REF)
    sys.path.insert(0, REF)
    import test as rtest                                     # the reference's evaluation script (module level is import-safe)
    iouv = torch.linspace(0.5, 0.95, 10)
    fx = {"iouv": iouv.numpy()}
    cases = [(80, 4, 3, (), ()), (81, 5, 4, (2,), (3,)), (82, 3, 2, (), (0, 1, 2)), (83, 6, 16, (0,), ()), (84, 2, 1, (), ())]
    for ci, (seed, nimg, nc, ep, el) in enumerate(cases):
        outs, targets = make_case(seed, nimg, nc, ep, el)
        ref_out = [o.clone() for o in outs]
        ref_stats = rtest.get_batch_statistics(ref_out, targets.clone(), iouv, 10)
        my_out = [o.clone() for o in outs]
        my_stats = ref_ops.get_batch_statistics(my_out, targets.clone(), iouv, 10)
        assert len(ref_stats) == len(my_stats), ci
        for a, b in zip(ref_stats, my_stats):
            assert np.array_equal(np.asarray(a[0]), np.asarray(b[0])), ci
            assert np.array_equal(np.asarray(a[1]), np.asarray(b[1])) and np.array_equal(np.asarray(a[2]), np.asarray(b[2])) and a[3] == b[3], ci
        for a, b in zip(ref_out, my_out):                    # the in-place radians -> degrees side effect
            assert torch.equal(a, b), ci
        fx[f"c{ci}_n"] = np.array([nimg, nc])
        fx[f"c{ci}_targets"] = targets.numpy()
        for b, o in enumerate(outs):
            fx[f"c{ci}_out{b}"] = o.numpy()
            fx[f"c{ci}_mut{b}"] = ref_out[b].numpy()
        fx[f"c{ci}_nstats"] = np.array(len(ref_stats))
        for k, st in enumerate(ref_stats):
            fx[f"c{ci}_tp{k}"] = np.asarray(st[0]).astype(np.uint8)
            fx[f"c{ci}_conf{k}"] = np.asarray(st[1], dtype=np.float32)
            fx[f"c{ci}_pcls{k}"] = np.asarray(st[2], dtype=np.float32)
            fx[f"c{ci}_tcls{k}"] = np.asarray(st[3], dtype=np.float32)
        if len(ref_stats):
            cat = [np.concatenate([np.asarray(s[i]) for s in ref_stats], 0) for i in range(4)]
            if cat[0].any():
                ra = rtest.ap_per_class(*cat)
                ma = ref_ops.ap_per_class(*cat)
                for x, y in zip(ra, ma):
                    assert np.allclose(x, y, rtol=0, atol=0), ci
                for name, x in zip(("p", "r", "ap", "f1", "cls"), ra):
                    fx[f"c{ci}_{name}"] = np.asarray(x)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g8_map.npz"), **fx)
    print("G8 ok", {k: v.shape for k, v in fx.items() if k.endswith("_ap")})


if __name__ == "__main__":
    main()
