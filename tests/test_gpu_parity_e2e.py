"""GPU: end-to-end parity statements the round-1 review asked for ("Close the parity gaps that are testable now").

(a) whole-network TRAIN-MODE (batch-statistics BatchNorm) forward + fused loss + backward against the oracles.  Train-mode BN at
    random init amplifies any rounding difference with depth, so the bound is stated against bf16's own noise floor: the distance of
    the HIP path to the bf16-emulating oracle may not exceed 1.5x the distance between the fp32 oracle and that same emulation
    (+ a small absolute slack), for the loss and for the gradient direction (1 - cos over all parameters).
(b) end-to-end loss and decoded boxes on the BASELINE configurations' shapes (C1 yolov4 kfiou 416 b2, C2 yolov4 608, C3 yolov7 csl
    800, C4 yolov7 kfiou nc=16 800 — the bench network) against the FP32 oracle, eval-mode BatchNorm.  ASSERTED is the north star's own
    line: boxes and every loss item < 1e-3 relative (measured r04: boxes <= 3.1e-4, items <= 7.3e-5).  Scores (obj * cls after two sigmoids
    of bf16-rounded logits: a 2^-9 relative logit error of a logit of magnitude ~4 is ~8e-3 absolute in the logit, times sigmoid' <= 0.25)
    are held to 5e-4 (yolov4; measured 6.7e-5) / 3e-4 (yolov7; measured 4.2e-5 since r05, 8.8e-4 in r04 — the fused head of r05 no longer rounds
    x + ImplicitA to bf16); the raw head maps — bf16 activations end to end, not a north-star quantity — to 5e-3 (yolov4; measured 2.0e-3 ... 2.5e-3
    = a few bf16 ulps accumulated over ~100 layers) / 5e-4 (yolov7; measured 1.1e-4).  Written to gpurun_out/r02_parity_e2e.json and printed.
(c) r06: the fraction of rows whose confidence-threshold decision (conf_thres = 0.001, test.py:270) differs between the two paths, with the head
    biases moved so that the score distribution is centred on the line."""
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
    # scores / raw head maps, per family (r06: tightened to what r05 measured, x4-x5 margin).  yolov4: the head is a plain bf16 conv chain, the maps
    # carry a few bf16 ulps (measured 2.0e-3 ... 2.5e-3; scores 6.7e-5).  yolov7: since r05 the fused head keeps ImplicitA in an fp32 bias
    # (W (x + a) + b = W x + (b + W a), ryolo_head_bias_fold) instead of rounding x + a to a bf16 tensor in front of the GEMM, and applies ImplicitM
    # in the fp32 epilogue: head maps 2.3e-3 (r04) -> 1.1e-4, scores 8.8e-4 (r04, C3) -> 4.2e-5 (profiles/r05_parity_e2e.json).
    tol_score, tol_maps = (3e-4, 5e-4) if ver == "yolov7" else (5e-4, 5e-3)
    assert e_score < tol_score and max(e_maps) < tol_maps, rep


def test_confidence_threshold_flips_are_rare_and_sit_on_the_line():
    """test.py:270 evaluates at conf_thres = 0.001, and profiles/r05_map_parity.json showed 65 661 detections on the HIP path
    against 65 573 on the fp32 oracle path for the SAME weights — candidates whose score crosses the 0.001 line under bf16 activations.  C4's
    shape (yolov7 kfiou nc=16 800^2), weights as in the end-to-end test but with the head biases moved so that the score distribution is CENTRED
    on the threshold (objectness ~ sigmoid(-4.6) = 0.01, class ~ sigmoid(-2.2) = 0.1: obj * cls ~ 0.001 — the worst case for flips; at random
    init every score is ~ 0.25 and nothing can flip).  Asserted: (i) the rows whose side of the line differs between the two paths are < 0.5 % of
    the rows, although ~half of all rows sit within a few percent of the line; (ii) EVERY flipped row's fp32 score is within 0.5 % of the
    line (element-wise; the relative-L2 score error of the previous test is ~4e-5) — a flip is rounding at the threshold, never a wrong score."""
    from ryolov4_amd.model.yolo import Yolo
    ver, mode, nc, S, B = "yolov7", "kfiou", 16, 800, 1
    net = Yolo(nc, CFG, mode, ver)
    sd = fill_state(net.state_dict())
    attrs = nc + 6
    for k in ("neck.conv5.conv.0.bias", "neck.conv6.conv.0.bias", "neck.conv7.conv.0.bias"):
        b = sd[k].clone()
        ch = torch.arange(b.numel()) % attrs
        b[ch == 5] += -4.6                                             # objectness logit
        b[ch >= 6] += -2.2 + 0.02 * (ch[ch >= 6] - 6).float()          # class logits (a slope so that the arg-max class is well defined)
        sd[k] = b
    net.load_state_dict(sd)
    net.to(DEV).eval()
    orc = ref_model.Yolo(nc, CFG, mode, ver)
    orc.load_state_dict(sd)
    orc.eval()
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        _, inf_o = orc(x, False)
        _, inf = net(x.to(DEV), training=False)
    inf = inf.cpu()
    thr = 0.001
    s_o = (inf_o[..., 6:] * inf_o[..., 5:6]).max(-1)[0].flatten().double()          # lib/general.py:155-157
    s_h = (inf[..., 6:] * inf[..., 5:6]).max(-1)[0].flatten().double()
    rows = s_o.numel()
    flip = (s_o > thr) != (s_h > thr)
    nflip = int(flip.sum())
    near = float(((s_o - thr).abs() < 0.05 * thr).double().mean())
    worst = float(((s_o[flip] - thr).abs() / thr).max()) if nflip else 0.0
    rep = dict(rows=rows, pass_fp32=int((s_o > thr).sum()), pass_hip=int((s_h > thr).sum()), flips=nflip, flip_frac=nflip / rows,
               rows_within_5pct_of_line=near, worst_flip_distance_rel=worst, score_rel_l2=rel(s_h, s_o))
    _report("conf_threshold_flips_C4", rep)
    assert 0.02 < rep["pass_fp32"] / rows < 0.98, rep                  # the line really cuts through the distribution
    assert nflip / rows < 5e-3, rep
    assert worst < 5e-3, rep                                           # element-wise distance of a flipped row's fp32 score from the line: < 0.5 % of it
