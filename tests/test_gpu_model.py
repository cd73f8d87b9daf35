"""GPU parity tests of the conv stack and the fused loss (through the C ABI) against the torch-CPU fp32 oracle and the
golden fixtures captured from the reference.

Tolerances.  The loss kernels are fp32: loss items within 1e-4 relative, gradients rtol 2e-3 (north_star: 1e-3 on losses),
target indices bit-exact.  The conv stack stores activations in bf16 (fp32 accumulate, fp32 BN statistics), so head maps
and parameter gradients are compared by relative L2 error against the fp32 oracle: <= 2e-2 forward, <= 6e-2 gradients
(bf16 has 8 mantissa bits; ~100 layers deep)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import ref_model, ref_ops
from tests.bf16_emu import emulate_bf16
from ryolov4_amd.synth import CFG, HYP, fill_state, synth_targets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-12))


# ------------------------------------------------------------------------------------------------ full networks
def _five(b, na):
    B_, _, gs, _ = b.shape
    return b.view(B_, na, -1, gs, gs).permute(0, 1, 3, 4, 2)


@pytest.mark.parametrize("ver", ["yolov4", "yolov5", "yolov7"])
@pytest.mark.parametrize("mode", ["csl", "kfiou"])
def test_full_network_eval_vs_golden_and_oracle(golden_dir, ver, mode):
    """Eval-mode forward (running-statistics BN) of every ver x mode at 64x64 with the closed-form weights:
    vs the fp32 oracle (rel-L2 <= 1e-2; observed 2e-3 = bf16 storage) and vs the samples captured from the imported reference."""
    from ryolov4_amd.model.yolo import Yolo
    g2 = np.load(os.path.join(golden_dir, "g2_fullnet.npz"))
    net = Yolo(2, CFG, mode, ver)
    sd = fill_state(net.state_dict())
    net.load_state_dict(sd, strict=True)
    net.to(DEV).eval()
    orc = ref_model.Yolo(2, CFG, mode, ver)
    orc.load_state_dict(sd, strict=True)
    orc.eval()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    na = 3 if mode == "csl" else 18
    with torch.no_grad():
        outs, inf = net(x.to(DEV), training=False)
        hm_o = orc.head_maps(x)
    tag = f"{ver}_{mode}_eval"
    for k, (a, b) in enumerate(zip(outs, hm_o)):
        assert rel(a.cpu(), _five(b, na)) < 1e-2, (tag, k, rel(a.cpu(), _five(b, na)))
        nchw = a.cpu().permute(0, 1, 4, 2, 3).reshape(b.shape)
        samp = nchw.flatten()[:: max(1, nchw.numel() // 64)][:64].numpy()
        ref = g2[f"{tag}_head{k}_sample"]
        assert np.linalg.norm(samp - ref) / (np.linalg.norm(ref) + 1e-9) < 2e-2, (tag, k)
    _, inf_o = ref_ops.decode(hm_o, orc.anchors, 2, mode)
    inf = inf.cpu()
    assert inf.shape == inf_o.shape
    cols = [0, 1, 2, 3, 5, 6, 7] if mode == "csl" else list(range(8))       # csl theta = argmax over 180 bins: compared separately
    assert rel(inf[..., cols], inf_o[..., cols]) < 1e-2
    if mode == "csl":
        assert float((inf[..., 4] - inf_o[..., 4]).abs().lt(1e-6).float().mean()) > 0.9


@pytest.mark.parametrize("ver,mode", [("yolov7", "kfiou"), ("yolov7", "csl"), ("yolov4", "kfiou"), ("yolov5", "csl")])
def test_full_network_backward_frozen_bn(ver, mode):
    """Whole-network forward + fused loss + backward with BatchNorm frozen to its running statistics (so the comparison is
    well conditioned: train-mode BN at random init amplifies ANY rounding difference exponentially with depth), against the
    bf16-emulating oracle in eval mode.  Checks every dgrad / wgrad / concat-slice accumulation of the real wiring."""
    from ryolov4_amd.lib.loss import ComputeCSLLoss, ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    nc = 16
    net = Yolo(nc, CFG, mode, ver)
    sd = fill_state(net.state_dict())
    net.load_state_dict(sd)
    net.to(DEV).eval()
    net.frozen_bn = True
    orc = ref_model.Yolo(nc, CFG, mode, ver)
    orc.load_state_dict(sd)
    emulate_bf16(orc)
    orc.eval()
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(7))
    tg = synth_targets(2, 8, nc, mode == "csl", seed=3, img_size=96)
    outs_o = orc(x, True)
    loss_o, items_o = ref_ops.compute_loss(outs_o, tg, orc.anchors, nc, mode, HYP)
    loss_o.backward()
    crit = (ComputeCSLLoss if mode == "csl" else ComputeKFIoULoss)(net, HYP)
    outs = net(x.to(DEV), training=True)
    for a, b in zip(outs, outs_o):
        assert rel(a.cpu(), b) < 1e-2, rel(a.cpu(), b)
    loss, items = crit(outs, tg.to(DEV))
    loss.backward()
    assert abs(items["total_loss"] - float(items_o["total_loss"])) < 2e-3 * abs(float(items_o["total_loss"]))
    gp = torch.cat([p.grad.flatten().cpu() for p in net.parameters()]).double()
    go = torch.cat([q.grad.flatten() for q in orc.parameters()]).double()
    cos = float((gp @ go) / (gp.norm() * go.norm()))
    errs = sorted(((rel(p.grad.cpu(), q.grad), n) for (n, p), (_, q) in zip(net.named_parameters(), orc.named_parameters())), reverse=True)
    assert cos > 0.999 and errs[0][0] < 0.15 and errs[len(errs) // 2][0] < 0.03, (cos, errs[:5], errs[len(errs) // 2])


def test_full_network_train_mode_within_bf16_noise_floor():
    """Batch-statistics BN at random init is chaotic, so an absolute tolerance is meaningless; instead the HIP path must sit
    inside the noise floor of bf16 itself: its distance to the bf16-emulating oracle may not exceed 1.5x the distance between
    the fp32 oracle and the bf16-emulating oracle (same weights, same input)."""
    from ryolov4_amd.model.yolo import Yolo
    nc, mode, ver = 2, "kfiou", "yolov7"
    net = Yolo(nc, CFG, mode, ver)
    sd = fill_state(net.state_dict())
    net.load_state_dict(sd)
    net.to(DEV).train()
    o32 = ref_model.Yolo(nc, CFG, mode, ver)
    o32.load_state_dict(sd)
    o32.train()
    o16 = ref_model.Yolo(nc, CFG, mode, ver)
    o16.load_state_dict(sd)
    emulate_bf16(o16)
    o16.train()
    x = torch.rand(2, 3, 160, 160, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        a32, a16 = o32(x, True), o16(x, True)
        ours = net(x.to(DEV), training=True)
    for k in range(3):
        floor = rel(a32[k], a16[k])
        mine = rel(ours[k].cpu(), a16[k])
        assert mine < 1.5 * floor + 1e-2, (k, mine, floor)


# ------------------------------------------------------------------------------------------------ loss kernels
@pytest.mark.parametrize("mode,nc", [("csl", 2), ("csl", 16), ("kfiou", 2), ("kfiou", 16)])
def test_loss_golden(golden_dir, mode, nc):
    from ryolov4_amd.lib import loss as L

    class M:
        pass
    m = M()
    m.anchors, m.nc = ref_ops.make_anchors(CFG, mode), nc
    crit = (L.ComputeCSLLoss if mode == "csl" else L.ComputeKFIoULoss)(m, HYP)
    assert set(crit.loss_items) == ({"reg_loss", "theta_loss", "conf_loss", "cls_loss", "total_loss"} if mode == "csl"
                                    else {"reg_loss", "conf_loss", "cls_loss", "total_loss"})
    g = np.load(os.path.join(golden_dir, "g46_loss.npz"))
    for case in range(3):
        tag = f"{mode}_nc{nc}_c{case}"
        tg = torch.from_numpy(g[f"{tag}_targets"]).to(DEV)
        outs = [torch.from_numpy(g[f"{tag}_out{i}"].astype(np.float32)).to(DEV).requires_grad_() for i in range(3)]
        loss, items = crit(outs, tg)
        assert loss.shape == (1,)
        names = [str(s) for s in g[f"{tag}_item_names"]]
        for nm, ref in zip(names, g[f"{tag}_items"]):
            assert abs(items[nm] - ref) < 1e-4 * max(1.0, abs(ref)), (tag, nm, items[nm], ref)
        if loss.requires_grad:
            loss.backward()
            for i in range(3):
                np.testing.assert_allclose(outs[i].grad.cpu().numpy(), g[f"{tag}_grad{i}"], rtol=2e-3, atol=2e-7, err_msg=f"{tag} grad{i}")
        # bit-exact target assignment (indices) read back from the kernel's match records
        recs = crit.debug_matches()
        for i in range(3):
            exp = g[f"{tag}_idx{i}"].reshape(-1, 5)
            got = recs[i][:, :5]
            assert np.array_equal(got, exp), (tag, i, got.shape, exp.shape)
        # no-grad path (test.py:188-190)
        with torch.no_grad():
            l2, it2 = crit([o.detach() for o in outs], tg)
        assert abs(it2["total_loss"] - items["total_loss"]) < 1e-6 * max(1.0, abs(items["total_loss"]))


def test_out_of_range_image_index_raises_also_without_the_per_call_sync(golden_dir):
    """lib/loss.py:209,385: the reference indexes pi[b, a, gj, gi] with the targets' image column and raises IndexError when it is >= the
    batch.  sync_items=True reads the dropped-row count with the loss items; sync_items=False (the asynchronous training path) copies it
    to pinned memory behind an event and raises at the NEXT call or at flush() — never silently."""
    from ryolov4_amd.lib import loss as L

    class M:
        pass
    m = M()
    m.anchors, m.nc = ref_ops.make_anchors(CFG, "kfiou"), 2
    g = np.load(os.path.join(golden_dir, "g46_loss.npz"))
    tag = "kfiou_nc2_c0"
    outs = [torch.from_numpy(g[f"{tag}_out{i}"].astype(np.float32)).to(DEV) for i in range(3)]
    good = torch.from_numpy(g[f"{tag}_targets"]).to(DEV)
    bad = good.clone()
    bad[0, 0] = outs[0].shape[0] + 3
    crit = L.ComputeKFIoULoss(m, HYP)
    with pytest.raises(IndexError):
        crit(outs, bad)
    crit = L.ComputeKFIoULoss(m, HYP)
    crit(outs, good, sync_items=False)
    crit(outs, bad, sync_items=False)                              # nothing read back yet
    assert float(crit.dropped_targets) == 1.0
    with pytest.raises(IndexError):
        crit(outs, good, sync_items=False)                         # the previous call's count surfaces here
    crit(outs, bad, sync_items=False)
    with pytest.raises(IndexError):
        crit.flush()
    crit.flush()                                                   # consumed: no second raise


@pytest.mark.parametrize("tag", ["csl_nc2", "kfiou_nc2", "kfiou_nc16", "csl_nc16", "csl_nc16_empty"])
def test_focal_loss_golden(golden_dir, tag):
    """hyp['fl_gamma'] > 0: FocalLoss (lib/loss.py:10-33) around the objectness / class / CSL-angle BCE terms, with non-unit pos_weights —
    loss items 1e-4 and logit gradients rtol 2e-3 against fixture G10 (the reference itself ran with fl_gamma = 1.5, obj_pw = 1.3,
    cls_pw = 0.8: tests/golden/make_golden_focal.py); the last case has no targets."""
    from ryolov4_amd.lib import loss as L
    g = np.load(os.path.join(golden_dir, "g10_focal.npz"))
    mode, nc = tag.split("_")[0], int(tag.split("_")[1][2:])
    hyp = {str(k): float(v) for k, v in zip(g["hyp_keys"], g["hyp_vals"])}
    assert hyp["fl_gamma"] == 1.5

    class M:
        pass
    m = M()
    m.anchors, m.nc = ref_ops.make_anchors(CFG, mode), nc
    crit = (L.ComputeCSLLoss if mode == "csl" else L.ComputeKFIoULoss)(m, hyp)
    tg = torch.from_numpy(g[f"{tag}_targets"]).to(DEV)
    outs = [torch.from_numpy(g[f"{tag}_out{i}"].astype(np.float32)).to(DEV).requires_grad_() for i in range(3)]
    loss, items = crit(outs, tg)
    for nm, ref in zip([str(s) for s in g[f"{tag}_item_names"]], g[f"{tag}_items"]):
        assert abs(items[nm] - ref) < 1e-4 * max(1.0, abs(ref)), (tag, nm, items[nm], ref)
    loss.backward()
    for i in range(3):
        np.testing.assert_allclose(outs[i].grad.cpu().numpy(), g[f"{tag}_grad{i}"], rtol=2e-3, atol=2e-7, err_msg=f"{tag} grad{i}")


@pytest.mark.parametrize("nc,B,S,per", [(2, 2, 64, 6), (16, 2, 96, 20), (16, 1, 64, 0)])
def test_sl1iou_extra_mode_vs_fp64_oracle(nc, B, S, per):
    """EXTRA mode `sl1iou` (ComputeSL1IoULoss; SURVEY §8a L7 / BASELINE config C2): the reference ships no code for the smooth-L1-IoU
    loss its Readme names, so there is NO reference oracle — the HIP kernel is held to this build's own fp64 definition
    (oracle/ref_ops.sl1iou_loss: direction from smooth-L1, magnitude |-log SkewIoU| with the detectron2-semantics IoU): loss items 1e-4,
    logit gradients rtol 5e-3 (the IoU inside is fp32 on both sides; the oracle's direction term is fp64)."""
    from ryolov4_amd.lib import loss as L
    from ryolov4_amd.synth import synth_targets

    class M:
        pass
    m = M()
    m.anchors, m.nc = ref_ops.make_anchors(CFG, "kfiou"), nc
    crit = L.ComputeSL1IoULoss(m, HYP)
    assert set(crit.loss_items) == {"reg_loss", "conf_loss", "cls_loss", "total_loss"}
    tg = synth_targets(B, per, nc, False, seed=5, edge_cases=True) if per else torch.zeros((0, 7))
    g = torch.Generator().manual_seed(9)
    outs = [torch.randn(B, 18, S // s, S // s, nc + 6, generator=g).half().float() for s in (8, 16, 32)]
    o_ref = [o.clone().requires_grad_() for o in outs]
    l_ref, it_ref = ref_ops.compute_loss(o_ref, tg, m.anchors, nc, "sl1iou", HYP)
    l_ref.backward()
    o_dev = [o.to(DEV).requires_grad_() for o in outs]
    loss, items = crit(o_dev, tg.to(DEV))
    loss.backward()
    for k in crit.KEYS:
        assert abs(items[k] - float(it_ref[k])) < 1e-4 * max(1.0, abs(float(it_ref[k]))), (k, items[k], float(it_ref[k]))
    if per:
        assert items["reg_loss"] > 0
    for a, b in zip(o_dev, o_ref):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=5e-3, atol=5e-7)


@pytest.mark.parametrize("mode,B,per", [("kfiou", 16, 64), ("csl", 64, 64), ("kfiou", 1, 3)])
def test_loss_target_assignment_many_targets_bit_exact(mode, B, per):
    """Target assignment at bench-like target counts (every one of the 32 workgroups per scale owns a non-empty candidate range):
    match records in the reference's enumeration order, bit-exact against oracle.build_targets (itself pinned to the reference by
    the g46 index vectors), and loss items against the oracle loss (1e-4 relative)."""
    from ryolov4_amd.lib import loss as L

    class M:
        pass
    nc, S = 16, 416
    m = M()
    m.anchors, m.nc = ref_ops.make_anchors(CFG, mode), nc
    crit = (L.ComputeCSLLoss if mode == "csl" else L.ComputeKFIoULoss)(m, HYP)
    tg = synth_targets(B, per, nc, mode == "csl", seed=11, img_size=S, edge_cases=True)
    na = 3 if mode == "csl" else 18
    attrs = (5 + 180 + nc) if mode == "csl" else (6 + nc)
    g = torch.Generator().manual_seed(5)
    outs = [torch.randn(B, na, S // st, S // st, attrs, generator=g) * 0.5 for st in (8, 16, 32)]
    with torch.no_grad():
        _, items = crit([o.to(DEV) for o in outs], tg.to(DEV))
    recs = crit.debug_matches()
    bt = ref_ops.build_targets([(o.shape[2], o.shape[3]) for o in outs], tg, m.anchors, mode)
    for i in range(3):
        exp = torch.stack((bt[i]["b"], bt[i]["a"], bt[i]["gj"], bt[i]["gi"], bt[i]["c"], bt[i]["tidx"]), 1).numpy()
        assert np.array_equal(recs[i][:, :6], exp), (i, recs[i].shape, exp.shape)
    _, items_o = ref_ops.compute_loss(outs, tg, m.anchors, nc, mode, HYP)
    assert abs(items["total_loss"] - float(items_o["total_loss"])) < 1e-4 * abs(float(items_o["total_loss"]))


def test_loss_backward_chain_rule_scalar_is_applied_on_device():
    """loss.backward() (incoming gradient exactly 1: the scaling kernel exits on the device) vs (3 * loss).backward()."""
    from ryolov4_amd.lib import loss as L

    class M:
        pass
    m = M()
    m.anchors, m.nc = ref_ops.make_anchors(CFG, "kfiou"), 2
    crit = L.ComputeKFIoULoss(m, HYP)
    tg = synth_targets(2, 6, 2, False, seed=4, img_size=128).to(DEV)
    g = torch.Generator().manual_seed(9)
    base = [torch.randn(2, 18, 128 // st, 128 // st, 8, generator=g) for st in (8, 16, 32)]
    grads = []
    for k in (1.0, 3.0):
        outs = [b.clone().to(DEV).requires_grad_() for b in base]
        loss, _ = crit(outs, tg)
        (loss * k).backward() if k != 1.0 else loss.backward()
        grads.append([o.grad.clone() for o in outs])
    for a, b in zip(*grads):
        assert a.abs().sum() > 0
        torch.testing.assert_close(b, 3.0 * a, rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_captured_inference_replay_matches_eager():
    """hipGraph capture of forward + decode (BASELINE config C5): replays on new inputs equal the eager path bit for bit."""
    import torch
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, fill_state
    m = Yolo(16, CFG, "kfiou", "yolov7")
    m.load_state_dict(fill_state(m.state_dict()))
    m.cuda().eval()
    run = m.capture_inference(2, 128)
    for seed in (1, 2):
        x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(seed)).cuda()
        with torch.no_grad():
            _, ref = m(x, False)
            ref = ref.clone()
        _, got = run(x)
        assert torch.equal(ref, got)


@pytest.mark.gpu
def test_captured_inference_with_post_process_in_the_graph():
    """BASELINE config C5: forward + decode + post_process (score filter, radix-select top-K, rotated NMS with the on-device greedy
    reduce) captured in ONE hipGraph on worst-case buffers, counts left on the device: replays on new inputs give the eager
    post_process result bit for bit (the eager path is checked against the oracle in test_gpu_postprocess.py)."""
    import torch
    from ryolov4_amd.lib.general import post_process
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, fill_state
    m = Yolo(16, CFG, "kfiou", "yolov7")
    m.load_state_dict(fill_state(m.state_dict()))
    m.cuda().eval()
    run = m.capture_inference(2, 128, post=(0.05, 0.4))
    for seed in (3, 4):
        x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(seed)).cuda()
        with torch.no_grad():
            _, inf = m(x, False)
            want = post_process(inf.clone(), 0.05, 0.4)
        _, _, dets, num = run(x)
        n = num.cpu().tolist()
        assert sum(n) > 0
        for b in range(2):
            assert n[b] == want[b].shape[0] and torch.equal(dets[b, :n[b]], want[b])
            assert float(dets[b, n[b]:].abs().sum()) == 0.0


@pytest.mark.gpu
def test_load_state_dict_after_first_forward_keeps_the_engine_in_sync():
    """Checkpoint loading AFTER the runtime exists (parameters are views of one flat buffer by then): load_state_dict copies in
    place, the next forward must see the new weights (bf16 images are repacked every forward) — same output as a fresh model."""
    from ryolov4_amd.model.yolo import Yolo
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(3)).to(DEV)
    a = Yolo(2, CFG, "kfiou", "yolov7").to(DEV).eval()
    with torch.no_grad():
        a(x, False)                                                # builds the runtime + plan with the random init
    sd = fill_state(a.state_dict())
    a.load_state_dict(sd, strict=True)
    b = Yolo(2, CFG, "kfiou", "yolov7")
    b.load_state_dict(sd, strict=True)
    b.to(DEV).eval()
    with torch.no_grad():
        _, ia = a(x, False)
        _, ib = b(x, False)
    assert torch.equal(ia, ib)
    saved = {k: v.clone() for k, v in a.state_dict().items()}     # save -> load round trip (train.py:88-90)
    for k in sd:
        assert torch.equal(saved[k].cpu(), sd[k].cpu()), k


def test_repconv_inference_fold_matches_unfused_branches():
    """Eval-mode RepConv as ONE re-parameterised 3x3 GEMM (ryolo_repconv_fold) vs the un-fused two-branch plan the reference's
    forward describes (model/utils.py:209-215): same module, same input; bf16 weight rounding is the only difference (rel-L2 < 6e-3),
    and both sit within 1e-2 of the fp32 oracle."""
    import subprocess, sys, json
    code = r'''
import json, torch
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, fill_state
m = Yolo(2, CFG, "kfiou", "yolov7"); m.load_state_dict(fill_state(m.state_dict())); m.cuda().eval()
x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(5)).cuda()
with torch.no_grad():
    outs, inf = m(x, False)
print(json.dumps([o.double().sum().item() for o in outs] + [inf.double().abs().sum().item()]))
torch.save([o.cpu() for o in outs], "/tmp/_rep_%s.pt" % __import__("os").environ.get("RYOLO_FOLD_REPCONV", "1"))
'''
    import gc
    gc.collect()
    torch.cuda.empty_cache()                       # the child processes cannot reuse this process's cached blocks
    for flag in ("1", "0"):
        env = dict(os.environ, RYOLO_FOLD_REPCONV=flag)
        subprocess.run([sys.executable, "-c", code], check=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    a, b = torch.load("/tmp/_rep_1.pt"), torch.load("/tmp/_rep_0.pt")
    for fa, fb in zip(a, b):
        assert rel(fa, fb) < 6e-3, rel(fa, fb)
        assert not torch.equal(fa, fb)                       # the fold really ran (different rounding), not the same plan twice


@pytest.mark.parametrize("mode", ["csl", "kfiou"])
def test_loss_gradient_with_many_duplicate_cells_vs_oracle_autograd(mode):
    """Targets crowded into a few cells: many (image, anchor, cell) triples are matched by 3-10 targets, the case where the reference's
    `pi[b, a, gj, gi]` gather back-propagates through index_put_(accumulate=True).  Loss items 1e-4, gradients rtol 2e-3 against the
    oracle's autograd (fp32 CPU)."""
    from ryolov4_amd.lib import loss as L

    class M:
        pass
    nc, S, B = 4, 64, 2
    m = M()
    m.anchors, m.nc = ref_ops.make_anchors(CFG, mode), nc
    crit = (L.ComputeCSLLoss if mode == "csl" else L.ComputeKFIoULoss)(m, HYP)
    tg = synth_targets(B, 40, nc, mode == "csl", seed=21, img_size=S)
    tg[:, 2:4] = 0.30 + 0.10 * tg[:, 2:4]                       # all centres inside a 6-pixel square -> the same 1-2 cells per scale
    tg[:, 4:6] = tg[:, 4:6].clamp(max=0.5)
    na = 3 if mode == "csl" else 18
    attrs = (5 + 180 + nc) if mode == "csl" else (6 + nc)
    g = torch.Generator().manual_seed(6)
    base = [torch.randn(B, na, S // st, S // st, attrs, generator=g) * 0.5 for st in (8, 16, 32)]
    outs = [b.clone().to(DEV).requires_grad_() for b in base]
    loss, items = crit(outs, tg.to(DEV))
    loss.backward()
    recs = crit.debug_matches()
    cells = np.concatenate([r[:, 6] + 10_000_000 * i for i, r in enumerate(recs)])
    _, counts = np.unique(cells, return_counts=True)
    assert counts.max() >= 3 and (counts >= 3).sum() >= 5        # the duplicate path is really exercised
    outs_o = [b.clone().requires_grad_() for b in base]
    loss_o, items_o = ref_ops.compute_loss(outs_o, tg, m.anchors, nc, mode, HYP)
    loss_o.backward()
    assert abs(items["total_loss"] - float(items_o["total_loss"])) < 1e-4 * abs(float(items_o["total_loss"]))
    for a, b in zip(outs, outs_o):
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=2e-3, atol=2e-7)


@pytest.mark.parametrize("ver,mode,scale", [("yolov7", "kfiou", 1.0), ("yolov7", "csl", 1.0), ("yolov4", "kfiou", 3.0)])
def test_compact_head_gradient_handoff_is_bit_identical_and_used(ver, mode, scale, monkeypatch):
    """r05: the engine's head backward takes the fused loss's gradient in compact form (objectness gradients + owner grid + dense rows of matched
    cells only, lib/loss.py compact_head_grad).  Same parameter gradients bit for bit as the dense hand-over (RYOLO_HEAD_SPARSE=0), also with
    a scaled loss; the tape must show the sparse entry point when the loss's own tensors arrive and the dense one otherwise."""
    from ryolov4_amd.lib import loss as L
    from ryolov4_amd.model.yolo import Yolo
    nc = 16
    net = Yolo(nc, CFG, mode, ver)
    net.load_state_dict(fill_state(net.state_dict()))
    net.to(DEV).train()
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(11)).to(DEV)
    tg = synth_targets(2, 8, nc, mode == "csl", seed=5, img_size=96).to(DEV)
    crit = (L.ComputeCSLLoss if mode == "csl" else L.ComputeKFIoULoss)(net, HYP)
    flat, names, items = [], [], []
    for sparse in (True, False, True):
        monkeypatch.setattr(L, "_HEAD_SPARSE", sparse)
        net.zero_grad(set_to_none=False)
        for p in net.parameters():
            if p.grad is not None:
                p.grad.zero_()
        outs = net(x, training=True)
        loss, _ = crit(outs, tg)
        (loss * scale).backward() if scale != 1.0 else loss.backward()
        flat.append(torch.cat([p.grad.flatten() for p in net.parameters()]).clone())
        items.append((float(loss), list(crit._used_head_obj)))
        names.append(sorted({n for _f, _a, n in _last_plan(net).bwd if n.startswith("ryolo_head_finish_bwd")}))
    assert names[0] == ["ryolo_head_finish_bwd_sparse"] and names[1] == ["ryolo_head_finish_bwd"], names
    # r05 default: the head GEMM writes the final layout itself (no finish pass in the forward tape); ImplicitM heads finish their parameter
    # gradients behind the weight gradient
    fwd_names = [n for _f, _a, n in _last_plan(net).fwd]
    bwd_names = [n for _f, _a, n in _last_plan(net).bwd]
    if os.environ.get("RYOLO_HEAD_FUSED", "1") != "0":
        assert not any(n.startswith("ryolo_head_finish_fwd") for n in fwd_names)
        assert "ryolo_chan_add" not in fwd_names and "ryolo_colsum_bf16" not in bwd_names      # ImplicitA folded into the bias / finished from s
        assert bwd_names.count("ryolo_head_wgrad_finish") == (3 if ver == "yolov7" else 0)
    else:       # (tests/test_gpu_forced_kernels.py re-runs this file on the r01-r04 form: row-major intermediate + finish pass with compact copies)
        assert fwd_names.count("ryolo_head_finish_fwd_obj") == 3 and "ryolo_head_wgrad_finish" not in bwd_names
    assert flat[0].abs().sum() > 0 and torch.isfinite(flat[0]).all()
    assert torch.equal(flat[0], flat[1])
    assert torch.equal(flat[0], flat[2])
    # the objectness pass of the loss read the engine's compact logits (same values: same loss, bit for bit) / the maps themselves
    assert items[0][1] == [True] * 3 and items[1][1] == [False] * 3 and items[0][0] == items[1][0] == items[2][0], items


def test_compact_head_gradient_is_dropped_when_the_criterion_ran_again_or_the_gradient_was_touched(monkeypatch):
    """The owner grids live in the criterion's workspace: a second call of the criterion before backward invalidates them, and a gradient that
    reaches the engine as another tensor (here: outputs used twice, autograd sums two maps) is not the loss's map.  Both fall back to the
    dense hand-over and give the right gradients."""
    from ryolov4_amd.lib import loss as L
    from ryolov4_amd.model.yolo import Yolo
    nc, mode = 2, "kfiou"
    net = Yolo(nc, CFG, mode, "yolov7")
    net.load_state_dict(fill_state(net.state_dict()))
    net.to(DEV).train()
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(12)).to(DEV)
    tg = synth_targets(2, 8, nc, False, seed=6, img_size=96).to(DEV)
    crit = L.ComputeKFIoULoss(net, HYP)

    def grads():
        return torch.cat([p.grad.flatten() for p in net.parameters()]).clone()

    def zero():
        for p in net.parameters():
            if p.grad is not None:
                p.grad.zero_()
    monkeypatch.setattr(L, "_HEAD_SPARSE", False)
    outs = net(x, training=True)
    loss, _ = crit(outs, tg)
    loss.backward()
    want = grads()
    monkeypatch.setattr(L, "_HEAD_SPARSE", True)
    # (1) criterion called again (other targets) between forward and backward of the first loss
    zero()
    outs = net(x, training=True)
    loss, _ = crit(outs, tg)
    with torch.no_grad():
        crit([o.detach() for o in outs], tg[:3])
    loss.backward()
    assert {n for _f, _a, n in _last_plan(net).bwd if n.startswith("ryolo_head_finish_bwd")} == {"ryolo_head_finish_bwd"}
    assert torch.equal(grads(), want)
    assert crit._used_head_obj == [True] * 3
    # (1b) an output edited in place through torch is no longer what the engine's compact copy describes
    zero()
    outs = net(x, training=True)
    with torch.no_grad():
        outs[1].mul_(1.0)
    loss, _ = crit(outs, tg)
    assert crit._used_head_obj == [True, False, True]
    loss.backward()
    assert torch.equal(grads(), want)
    # (2) two losses on the same outputs: the engine receives the SUM of two maps, a tensor the loss does not know
    zero()
    outs = net(x, training=True)
    l1, _ = crit(outs, tg)
    l2, _ = L.ComputeKFIoULoss(net, HYP)(outs, tg)
    (0.5 * l1 + 0.5 * l2).backward()
    assert {n for _f, _a, n in _last_plan(net).bwd if n.startswith("ryolo_head_finish_bwd")} == {"ryolo_head_finish_bwd"}
    torch.testing.assert_close(grads(), want, rtol=1e-5, atol=1e-7)


def _last_plan(net):
    graphs = [g for k, g in net.runtime()._graphs.items() if k[3]]         # the (one) training plan
    assert len(graphs) == 1
    return graphs[0]


@pytest.mark.gpu
def test_whole_training_step_replays_from_one_graph():
    """r06: forward + fused loss + two-stream backward + fused SGD step captured into ONE hipGraph (tools/bench_graph_step.py times it: 14.5 vs 14.3 ms
    eager at 8 images — a capability, not a speed-up).  The loss's dropped-row read-back (a pinned copy behind an event the next call waits on) kept
    the step un-capturable in r05; under capture the count stays on the device.  Two models from the same state: three eager steps on one, one
    captured step replayed three times on the other — the replays read the static batch, so parameters and loss end equal bit for bit."""
    import torch
    from ryolov4_amd.lib.loss import ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, fill_state, synth_targets
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(1)).cuda()
    tg = synth_targets(2, 6, 2, False, seed=2, img_size=96).cuda()
    finals, losses = [], []
    for captured in (False, True):
        m = Yolo(2, CFG, "kfiou", "yolov7")
        m.load_state_dict(fill_state(m.state_dict()))
        m.cuda().train()
        crit = ComputeKFIoULoss(m, HYP)
        rt = m.runtime()

        def step():
            loss, _ = crit(m(x, training=True), tg, sync_items=False)
            loss.backward()
            rt.sgd_step(0.01, 0.937, zero_grad=True)
            return loss

        if not captured:
            for _ in range(3):
                last = step()
            torch.cuda.synchronize()
        else:
            state = {k: v.clone() for k, v in m.state_dict().items()}
            flat, mom = rt.flat.clone(), None
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()                                            # warm-up outside the capture: plan build, lazy kernel attributes, allocator pools
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                last = step()
            torch.cuda.synchronize()
            # back to the initial state (parameters, momentum, BatchNorm running statistics), then three replays = three steps
            rt.flat.copy_(flat)
            rt.momentum_buf.zero_()
            m.load_state_dict(state)
            rt.gflat.zero_()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
        finals.append(rt.flat.clone())
        losses.append(float(last.detach()))
    assert torch.equal(finals[0], finals[1]), float((finals[0] - finals[1]).abs().max())
    assert losses[0] == losses[1]
