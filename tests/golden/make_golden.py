#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (/root/reference) in the build container.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
The reference cannot travel to the GPU box, so only data (inputs + expected outputs) is committed.
`detectron2` and `cv2` are absent here; they are replaced by in-memory stub modules.  The stubbed
`nms_rotated` records its arguments (fixture G7) and answers with the build's C oracle, so the final post_process
outputs in G7 pin everything in lib/general.post_process EXCEPT the third-party NMS itself (parity unpinned).

While generating, this script also asserts that the oracle restatement (oracle/ref_ops.py, oracle/ref_model.py)
reproduces the imported reference on every vector — that is the "pin" of the oracle.
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
from oracle import ref_ops, ref_model           # noqa: E402
from ryolov4_amd.synth import fill_state, synth_targets, CFG, HYP   # noqa: E402

NMS_CALLS = []


def _stub_nms(boxes, scores, thr):
    NMS_CALLS.append((boxes.clone(), scores.clone(), float(thr)))
    return torch.from_numpy(oracle.nms_rotated(boxes.numpy(), scores.numpy(), thr, gt_only=True))


def _install_stubs():
    d2 = types.ModuleType("detectron2")
    layers = types.ModuleType("detectron2.layers")
    rb = types.ModuleType("detectron2.layers.rotated_boxes")
    nms = types.ModuleType("detectron2.layers.nms")
    rb.pairwise_iou_rotated = lambda a, b: torch.from_numpy(oracle.pairwise_iou_rotated(a.numpy(), b.numpy()))
    nms.nms_rotated = _stub_nms
    sys.modules.update({"detectron2": d2, "detectron2.layers": layers, "detectron2.layers.rotated_boxes": rb,
                        "detectron2.layers.nms": nms, "cv2": types.ModuleType("cv2")})


def _np(t):
    return t.detach().cpu().numpy()


def main():
    _install_stubs()
    os.chdir(REF)
    sys.path.insert(0, REF)
    from model.yolo import Yolo as RefYolo
    from lib import loss as rloss, general as rgen
    out = os.path.join(ROOT, "tests", "golden")
    torch.manual_seed(0)
    np.random.seed(0)

    # ---------------------------------------------------------------- G2/G3: full nets @64x64, nc=2, closed-form weights
    g2 = {}
    for ver in ("yolov4", "yolov5", "yolov7"):
        for mode in ("csl", "kfiou"):
            ref = RefYolo(2, CFG, mode, ver)
            sd = fill_state(ref.state_dict())
            ref.load_state_dict(sd, strict=True)
            mine = ref_model.Yolo(2, CFG, mode, ver)
            assert list(mine.state_dict().keys()) == list(sd.keys()), (ver, mode, "state_dict key ABI")
            mine.load_state_dict(sd, strict=True)
            g = torch.Generator().manual_seed(1)
            x = torch.rand(2, 3, 64, 64, generator=g)
            for train in (False, True):      # eval first: it must see the pristine closed-form running statistics
                ref.train(train)
                mine.train(train)
                with torch.no_grad():
                    hm_ref = list(ref.neck(*reversed(ref.backbone(x))))
                    hm_mine = mine.head_maps(x)
                for a, b in zip(hm_ref, hm_mine):
                    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (ver, mode, train, (a - b).abs().max())
                tag = f"{ver}_{mode}_{'train' if train else 'eval'}"
                for k, a in enumerate(hm_ref):
                    g2[f"{tag}_sum{k}"] = np.array(a.double().sum().item())
                    g2[f"{tag}_abs{k}"] = np.array(a.double().abs().sum().item())
                    g2[f"{tag}_head{k}_sample"] = _np(a.flatten()[:: max(1, a.numel() // 64)][:64])
                if not train:
                    with torch.no_grad():
                        o_ref, inf_ref = ref.yolo([h.clone() for h in hm_ref], False)
                        o_mine, inf_mine = ref_ops.decode(hm_mine, mine.anchors, 2, mode)
                    assert torch.allclose(inf_ref, inf_mine, rtol=1e-5, atol=1e-5)
                    g2[f"{tag}_infer_sum"] = np.array(inf_ref.double().sum().item())
                    g2[f"{tag}_infer_sample"] = _np(inf_ref.flatten()[:: inf_ref.numel() // 64][:64])
            print("G2 ok", ver, mode)
    np.savez_compressed(os.path.join(out, "g2_fullnet.npz"), **g2)

    # ---------------------------------------------------------------- G3: YoloLayer decode on random logits
    g3 = {}
    for mode in ("csl", "kfiou"):
        for nc in (2, 16):
            ref = RefYolo(nc, CFG, mode, "yolov4")     # only .yolo / .anchors are used
            na = 3 if mode == "csl" else 18
            attrs = nc + (185 if mode == "csl" else 6)
            gs_list = (4, 2, 5)
            g = torch.Generator().manual_seed(3)
            logits = [(torch.randn(2, na * attrs, gs, gs, generator=g) * 2).half().float() for gs in gs_list]
            if mode == "csl":   # force an argmax tie to pin the first-max contract
                v = logits[0].view(2, na, attrs, 4, 4)
                v[0, 0, 5 + nc + 7, 1, 2] = 9.0
                v[0, 0, 5 + nc + 99, 1, 2] = 9.0
            with torch.no_grad():
                o_ref, inf_ref = ref.yolo([l.clone() for l in logits], False)
                o_mine, inf_mine = ref_ops.decode(logits, ref.anchors, nc, mode)
            assert all(torch.equal(a, b) for a, b in zip(o_ref, o_mine))
            assert torch.allclose(inf_ref, inf_mine, rtol=1e-6, atol=1e-6), (inf_ref - inf_mine).abs().max()
            tag = f"{mode}_nc{nc}"
            for k, l in enumerate(logits):
                g3[f"{tag}_logits{k}"] = _np(l).astype(np.float16)   # exactly representable
            g3[f"{tag}_infer"] = _np(inf_ref)
    np.savez_compressed(os.path.join(out, "g3_decode.npz"), **g3)
    print("G3 ok")

    # ---------------------------------------------------------------- G4/G6: targets, losses, grads
    g46 = {}
    for mode in ("csl", "kfiou"):
        for nc in (2, 16):
            ref = RefYolo(nc, CFG, mode, "yolov4")
            L = (rloss.ComputeCSLLoss if mode == "csl" else rloss.ComputeKFIoULoss)(ref, HYP)
            na = 3 if mode == "csl" else 18
            attrs = nc + (185 if mode == "csl" else 6)
            cases = ((2, 32, 12), (2, 64, 24), (1, 32, 0)) if mode == "csl" else ((2, 64, 12), (2, 96, 40), (2, 64, 0))
            for case, (B, S, nt) in enumerate(cases):
                tg = synth_targets(B, nt // max(B, 1) if nt else 0, nc, mode == "csl", seed=10 + case, edge_cases=True)
                g = torch.Generator().manual_seed(20 + case)
                outs = [torch.randn(B, na, S // s, S // s, attrs, generator=g).half().float().requires_grad_() for s in (8, 16, 32)]
                bt = L.build_targets(outs, tg)
                mine_bt = ref_ops.build_targets([(o.shape[2], o.shape[3]) for o in outs], tg, ref.anchors, mode)
                idx = bt[4] if mode == "csl" else bt[2]
                tbox = bt[1]
                tag = f"{mode}_nc{nc}_c{case}"
                for i in range(3):
                    b, a, gj, gi = idx[i]
                    m = mine_bt[i]
                    assert torch.equal(b, m["b"]) and torch.equal(a, m["a"]) and torch.equal(gj, m["gj"]) \
                        and torch.equal(gi, m["gi"]) and torch.equal(bt[0][i], m["c"]), (tag, i)
                    assert torch.allclose(tbox[i], m["tbox"], atol=1e-6), (tag, i)
                    g46[f"{tag}_idx{i}"] = _np(torch.stack((b, a, gj, gi, bt[0][i]), 1)) if b.numel() else np.zeros((0, 5), np.int64)
                    g46[f"{tag}_tbox{i}"] = _np(tbox[i])
                loss, items = L(outs, tg)
                if loss.requires_grad:
                    loss.backward()
                outs2 = [o.detach().clone().requires_grad_() for o in outs]
                loss2, items2 = ref_ops.compute_loss(outs2, tg, ref.anchors, nc, mode, HYP)
                loss2.backward()
                assert abs(loss.item() - loss2.item()) < 2e-5 * max(1, abs(loss.item())), (tag, loss.item(), loss2.item())
                for k in items:
                    assert abs(items[k] - float(items2[k])) < 2e-5 * max(1, abs(items[k])), (tag, k)
                for o, o2 in zip(outs, outs2):
                    assert torch.allclose(o.grad, o2.grad, rtol=1e-4, atol=1e-7), (tag, (o.grad - o2.grad).abs().max())
                g46[f"{tag}_targets"] = _np(tg)
                for i, o in enumerate(outs):
                    g46[f"{tag}_out{i}"] = _np(o).astype(np.float16)   # exactly representable
                    g46[f"{tag}_grad{i}"] = _np(o.grad)
                g46[f"{tag}_items"] = np.array([items[k] for k in sorted(items)], np.float64)
                g46[f"{tag}_item_names"] = np.array(sorted(items))
            print("G4/G6 ok", mode, nc)
    np.savez_compressed(os.path.join(out, "g46_loss.npz"), **g46)

    # ---------------------------------------------------------------- G5: elementwise box maths
    g5 = {}
    g = torch.Generator().manual_seed(5)
    p = torch.rand(64, 4, generator=g) * 4 + 0.05
    t = torch.rand(64, 4, generator=g) * 4 + 0.05
    p[0] = t[0]
    ci = rloss.bbox_ciou(p, t)
    assert torch.allclose(ci, ref_ops.bbox_ciou(p, t), atol=1e-6)
    g5.update(ciou_p=_np(p), ciou_t=_np(t), ciou=_np(ci))
    pk = torch.cat((p, (torch.rand(64, 1, generator=g) - 0.5) * 3.1), 1)
    tk = torch.cat((t, (torch.rand(64, 1, generator=g) - 0.5) * 3.1), 1)
    pk[1] = tk[1]
    pk[2, 2:4] = 1e-5
    tk[3, 2:4] = 2e4
    kl, kf = rloss.KFLoss()(pk, tk)
    kl2, kf2 = ref_ops.kf_loss(pk, tk)
    assert torch.allclose(kf, kf2, rtol=1e-4, atol=1e-6) and abs(kl.item() - kl2.item()) < 1e-4 * abs(kl.item()), (kl, kl2)
    g5.update(kf_p=_np(pk), kf_t=_np(tk), kf_loss=np.array(kl.item()), kfiou=_np(kf))
    for n1 in (1,):
        kl, kf = rloss.KFLoss()(pk[:n1], tk[:n1])
        g5.update(kf1_loss=np.array(kl.item()), kfiou1=_np(kf))
    ang = torch.tensor([np.pi / 2, -np.pi / 2, 1.6, -1.7, 0.0, 1.57, -1.5707964], dtype=torch.float32)
    na_ = rgen.norm_angle(ang.clone())
    assert torch.equal(na_, ref_ops.norm_angle(ang.clone()))
    g5.update(ang_in=_np(ang), ang_out=_np(na_))
    np.savez_compressed(os.path.join(out, "g5_boxmath.npz"), **g5)
    print("G5 ok")

    # ---------------------------------------------------------------- G7: post_process
    g7 = {}
    for case, (nc, M, ct, it) in enumerate(((2, 3000, 0.25, 0.4), (16, 5600, 0.001, 0.65), (16, 500, 0.7, 0.2), (2, 64, 0.99999, 0.4))):
        g = torch.Generator().manual_seed(70 + case)
        pred = torch.rand(1 if M > 5000 else 2, M, nc + 6, generator=g)
        pred[..., 0:2] *= 256
        pred[..., 2] = pred[..., 2] * 30 + 4
        pred[..., 3] = pred[..., 2] * (1 + 3 * torch.rand(pred.shape[0], M, generator=g))
        pred[..., 4] = (pred[..., 4] - 0.5) * np.pi
        pred[0, 10:20, 5:] = pred[0, 0:10, 5:]          # exact score ties -> pins the stable tie-break
        NMS_CALLS.clear()
        ref_in = pred.clone()
        outs = rgen.post_process(ref_in, conf_thres=ct, iou_thres=it)
        mine_in = pred.clone()
        mine = ref_ops.post_process(mine_in, ct, it)
        # the reference's argsort(descending) is unstable: compare as sets of rows when ties exist, exactly otherwise
        for a, b in zip(outs, mine):
            assert a.shape == b.shape, (case, a.shape, b.shape)
            if a.numel():
                assert torch.allclose(a[a[:, 5].argsort(stable=True)].sort(0)[0], b.sort(0)[0], atol=1e-6), case
        assert torch.equal(ref_in, mine_in)
        g7[f"c{case}_pred"] = _np(pred)
        g7[f"c{case}_cfg"] = np.array([nc, ct, it])
        g7[f"c{case}_mutated_sum"] = np.array(ref_in.double().sum().item())
        for b, o in enumerate(mine):
            g7[f"c{case}_out{b}"] = _np(o)
        g7[f"c{case}_ncalls"] = np.array(len(NMS_CALLS))
    np.savez_compressed(os.path.join(out, "g7_postprocess.npz"), **g7)
    print("G7 ok")


if __name__ == "__main__":
    main()
