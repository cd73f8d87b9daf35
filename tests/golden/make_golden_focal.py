"""Fixture G10: the reference's losses with FocalLoss ACTIVE (hyp['fl_gamma'] = 1.5) and non-unit pos_weights — run by importing
/root/reference (same stub modules as make_golden.py), asserting that the oracle restatement (oracle/ref_ops.py) reproduces
loss items and logit gradients while generating.  Run here (the reference does not travel):  python tests/golden/make_golden_focal.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from tests.golden import make_golden as MG  # noqa: E402
from oracle import ref_ops  # noqa: E402
from ryolov4_amd.synth import CFG, synth_targets  # noqa: E402

HYP_FL = {"fl_gamma": 1.5, "box": 0.05, "obj": 1.0, "obj_pw": 1.3, "cls": 0.5, "cls_pw": 0.8}


def main():
    MG._install_stubs()
    os.chdir(MG.REF)
    sys.path.insert(0, MG.REF)
    from model.yolo import Yolo as RefYolo
    from lib import loss as rloss
    g10 = {"hyp_keys": np.array(sorted(HYP_FL)), "hyp_vals": np.array([HYP_FL[k] for k in sorted(HYP_FL)], np.float64)}
    for tag, mode, nc, (B, S, nt) in (("csl_nc2", "csl", 2, (2, 32, 12)), ("kfiou_nc2", "kfiou", 2, (2, 64, 12)), ("kfiou_nc16", "kfiou", 16, (2, 64, 24)),
                                      ("csl_nc16", "csl", 16, (2, 64, 24)), ("csl_nc16_empty", "csl", 16, (1, 32, 0))):
        ref = RefYolo(nc, CFG, mode, "yolov4")
        L = (rloss.ComputeCSLLoss if mode == "csl" else rloss.ComputeKFIoULoss)(ref, HYP_FL)
        assert isinstance(L.BCEobj, rloss.FocalLoss) and isinstance(L.BCEcls, rloss.FocalLoss)
        na = 3 if mode == "csl" else 18
        attrs = nc + (185 if mode == "csl" else 6)
        tg = synth_targets(B, nt // max(B, 1) if nt else 0, nc, mode == "csl", seed=31, edge_cases=True)
        g = torch.Generator().manual_seed(41)
        outs = [torch.randn(B, na, S // s, S // s, attrs, generator=g).half().float().requires_grad_() for s in (8, 16, 32)]
        loss, items = L(outs, tg)
        loss.backward()
        outs2 = [o.detach().clone().requires_grad_() for o in outs]
        loss2, items2 = ref_ops.compute_loss(outs2, tg, ref.anchors, nc, mode, HYP_FL)
        loss2.backward()
        assert abs(loss.item() - loss2.item()) < 2e-5 * max(1, abs(loss.item())), (mode, loss.item(), loss2.item())
        for k in items:
            assert abs(items[k] - float(items2[k])) < 2e-5 * max(1, abs(items[k])), (mode, k)
        for o, o2 in zip(outs, outs2):
            assert torch.allclose(o.grad, o2.grad, rtol=1e-4, atol=1e-7), (mode, (o.grad - o2.grad).abs().max())
        if nt:      # every matched-target term must be live: a fixture whose targets match nothing pins only the objectness term
            assert all(items[k] > 1e-3 for k in items), (tag, items)
        g10[f"{tag}_targets"] = tg.numpy()
        for i, o in enumerate(outs):
            g10[f"{tag}_out{i}"] = o.detach().numpy().astype(np.float16)
            g10[f"{tag}_grad{i}"] = o.grad.numpy()
        g10[f"{tag}_items"] = np.array([items[k] for k in sorted(items)], np.float64)
        g10[f"{tag}_item_names"] = np.array(sorted(items))
        print("G10 ok", tag, {k: round(v, 5) for k, v in items.items()})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g10_focal.npz"), **g10)


if __name__ == "__main__":
    main()
