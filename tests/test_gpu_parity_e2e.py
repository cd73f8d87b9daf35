"""GPU: end-to-end parity statements the round-1 review asked for ("Close the parity gaps that are testable now").

(a) whole-network TRAIN-MODE (batch-statistics BatchNorm) forward + fused loss + backward against the oracles.  Train-mode BN at
    random init amplifies any rounding difference with depth, so the bound is stated against bf16's own noise floor: the distance of
    the HIP path to the bf16-emulating oracle may not exceed 1.5x the distance between the fp32 oracle and that same emulation
    (+ a small absolute slack), for the loss and for the gradient direction (1 - cos over all parameters).
(b) end-to-end loss and decoded boxes on the BASELINE configurations' shapes (C1 yolov4 kfiou 416 b2, C2 yolov4 608, C3 yolov7 csl
    800, C4 yolov7 kfiou nc=16 800 — the bench network) against the FP32 oracle, eval-mode BatchNorm.  ASSERTED is the north star's own
    line: boxes and every loss item < 1e-3 relative (measured r04: boxes <= 3.1e-4, items <= 7.3e-5).  Scores (obj * cls after two sigmoids
    of bf16-rounded logits: a 2^-9 relative logit error of a logit of magnitude ~4 is ~8e-3 absolute in the logit, times sigmoid' <= 0.25)
    are held to 2e-3 (measured 8.8e-4 on C3, 6.7e-5 on C1 / C2); the raw head maps — bf16 activations end to end, not a north-star
    quantity — to 5e-3 (measured 2.0e-3 ... 2.5e-3 = a few bf16 ulps accumulated over ~100 layers).  Written to
    gpurun_out/r02_parity_e2e.json and printed."""
import json
import os

import pytest
import torch

from oracle import ref_model, ref_ops
from tests.bf16_emu import emulate_bf16
from ryolov4_amd.synth import CFG, HYP, fill_state, synth_targets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _cos(a, b):
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def _report(key, val):
    path = os.path.join(ROOT, "gpurun_out", "r02_parity_e2e.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[key] = val
    json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    print("PARITY", key, json.dumps(val))


@pytest.mark.parametrize("ver,mode", [("yolov7", "kfiou"), ("yolov4", "csl")])
def test_full_network_train_mode_backward_vs_noise_floor(ver, mode):
    from ryolov4_amd.lib.loss import ComputeCSLLoss, ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    nc, B, S = 2, 4, 128
    net = Yolo(nc, CFG, mode, ver)
    sd = fill_state(net.state_dict())
    net.load_state_dict(sd)
    net.to(DEV).train()
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(4))
    tg = synth_targets(B, 8, nc, mode == "csl", seed=6, img_size=S)
    grads, losses = {}, {}
    for tag in ("fp32", "bf16emu"):
        orc = ref_model.Yolo(nc, CFG, mode, ver)
        orc.load_state_dict(sd)
        if tag == "bf16emu":
            emulate_bf16(orc)
        orc.train()
        loss, _ = ref_ops.compute_loss(orc(x, True), tg, orc.anchors, nc, mode, HYP)
        loss.backward()
        grads[tag] = torch.cat([q.grad.flatten() for q in orc.parameters()]).double()
        losses[tag] = float(loss)
    crit = (ComputeCSLLoss if mode == "csl" else ComputeKFIoULoss)(net, HYP)
    loss, items = crit(net(x.to(DEV), training=True), tg.to(DEV))
    loss.backward()
    g = torch.cat([p.grad.flatten().cpu() for p in net.parameters()]).double()
    floor_cos = 1.0 - _cos(grads["fp32"], grads["bf16emu"])
    mine_cos = 1.0 - _cos(g, grads["bf16emu"])
    floor_loss = abs(losses["fp32"] - losses["bf16emu"]) / abs(losses["bf16emu"])
    mine_loss = abs(items["total_loss"] - losses["bf16emu"]) / abs(losses["bf16emu"])
    _report(f"train_mode_backward_{ver}_{mode}", dict(one_minus_cos_hip_vs_bf16emu=mine_cos, one_minus_cos_fp32_vs_bf16emu=floor_cos,
                                                      loss_rel_hip_vs_bf16emu=mine_loss, loss_rel_fp32_vs_bf16emu=floor_loss,
                                                      cos_hip_vs_fp32=_cos(g, grads["fp32"])))
    assert torch.isfinite(g).all()
    assert mine_cos < 1.5 * floor_cos + 5e-3, (mine_cos, floor_cos)
    assert mine_loss < 1.5 * floor_loss + 2e-3, (mine_loss, floor_loss)
    # (no absolute bound on the direction is meaningful here: at this initialisation even the two ORACLES disagree — measured
    # 1 - cos(fp32, bf16-emulation) = 0.87 for yolov7: ~100 train-mode BatchNorm layers amplify a 2^-9 rounding difference by
    # 1.1-1.5x each.  Well-conditioned gradient checks: per block in train mode (test_gpu_blocks.py, 3e-2 per tensor) and the whole
    # network with frozen statistics (test_gpu_model.py, cos > 0.999); the drop-in test follows six SGD steps against the oracle.)


@pytest.mark.parametrize("cfg,ver,mode,nc,S,B", [("C1", "yolov4", "kfiou", 2, 416, 2), ("C2", "yolov4", "kfiou", 2, 608, 1),
                                                 ("C3", "yolov7", "csl", 16, 800, 1), ("C4", "yolov7", "kfiou", 16, 800, 1)])
def test_end_to_end_loss_and_boxes_vs_fp32_oracle(cfg, ver, mode, nc, S, B):
    from ryolov4_amd.lib.loss import ComputeCSLLoss, ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    net = Yolo(nc, CFG, mode, ver)
    sd = fill_state(net.state_dict())
    net.load_state_dict(sd)
    net.to(DEV).eval()
    orc = ref_model.Yolo(nc, CFG, mode, ver)
    orc.load_state_dict(sd)
    orc.eval()
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(8))
    tg = synth_targets(B, 16, nc, mode == "csl", seed=12, img_size=S)
    with torch.no_grad():
        hm_o, inf_o = orc(x, False)
        loss_o, items_o = ref_ops.compute_loss(hm_o, tg, orc.anchors, nc, mode, HYP)
        outs, inf = net(x.to(DEV), training=False)
        crit = (ComputeCSLLoss if mode == "csl" else ComputeKFIoULoss)(net, HYP)
        _, items = crit(outs, tg.to(DEV))
    inf = inf.cpu()
    box_cols = [0, 1, 2, 3] if mode == "csl" else [0, 1, 2, 3, 4]
    e_box = rel(inf[..., box_cols], inf_o[..., box_cols])
    e_score = rel(inf[..., 5:], inf_o[..., 5:])
    e_items = {k: abs(items[k] - float(items_o[k])) / max(abs(float(items_o[k])), 1e-12) for k in items}
    e_maps = [rel(a.cpu(), b) for a, b in zip(outs, hm_o)]
    rep = dict(config=f"{ver} {mode} nc={nc} {S}x{S} batch {B}, eval-mode BatchNorm, bf16 activations vs fp32 oracle",
               boxes_rel_l2=e_box, scores_rel_l2=e_score, loss_items_rel=e_items, head_maps_rel_l2=e_maps,
               north_star_1e3_met=dict(boxes=e_box < 1e-3, losses=max(e_items.values()) < 1e-3))
    _report(f"e2e_{cfg}", rep)
    assert e_box < 1e-3 and max(e_items.values()) < 1e-3, rep          # the north star's tolerance, as stated
    assert e_score < 2e-3 and max(e_maps) < 5e-3, rep                 # (docstring: why these two are not 1e-3)
