"""GPU: training TRAJECTORIES of the HIP path against the fp32 torch-CPU oracle: the reference's own loop —
train.py:28-33 initialisation, torch.optim.SGD(momentum 0.937, nesterov) or torch.optim.Adam (train.py:153-156) driving the
model's Parameters, loss.backward() / optimizer.step() / optimizer.zero_grad() (train.py:195-202) — for 40 (SGD) / 12 (Adam) steps on
one fixed batch, batch-statistics BatchNorm, same initial weights on both sides.

What can and cannot be asserted: at random init the per-step gradient DIRECTION of a 100-layer train-mode network is not reproducible
between any two roundings (tests/test_gpu_teacher_forced.py explains and checks every node separately); the LOSS CURVE is — it is
dominated by the well-conditioned last layers.  Asserted: every step's total loss within BAND of the oracle's, both curves fall by at
least a third, and the update of the detection-head parameters (the last, well-conditioned layers) has cos > 0.9 with the oracle's.
Curves are written to gpurun_out/r03_trajectory.json."""
import json
import os

import pytest
import torch

from oracle import ref_model, ref_ops
from ryolov4_amd.synth import CFG, HYP, synth_targets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAND = {"sgd": 0.05, "adam": 0.08}


def weights_init_normal(m):                                      # train.py:28-33
    if isinstance(m, torch.nn.Conv2d):
        torch.nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif isinstance(m, torch.nn.BatchNorm2d):
        torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
        torch.nn.init.constant_(m.bias.data, 0.0)


def _report(key, val):
    path = os.path.join(ROOT, "gpurun_out", "r03_trajectory.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[key] = val
    json.dump(doc, open(path, "w"), indent=1, sort_keys=True)


@pytest.mark.parametrize("opt_name,steps", [("sgd", 40), ("adam", 12)])
def test_loss_trajectory_follows_the_oracle(opt_name, steps):
    from ryolov4_amd.lib.loss import ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    nc, B, S, mode, ver = 2, 4, 128, "kfiou", "yolov7"
    torch.manual_seed(42)                                        # train.py:20-25
    orc = ref_model.Yolo(nc, CFG, mode, ver)
    orc.apply(weights_init_normal)
    sd0 = {k: v.clone() for k, v in orc.state_dict().items()}
    net = Yolo(nc, CFG, mode, ver)
    net.load_state_dict(sd0)
    net.to(DEV).train()
    orc.train()
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(7))
    tg = synth_targets(B, 8, nc, False, seed=9, img_size=S)

    def make_opt(params):
        if opt_name == "sgd":
            return torch.optim.SGD(params, lr=0.01, momentum=0.937, nesterov=True)
        return torch.optim.Adam(params, lr=1e-3)

    crit = ComputeKFIoULoss(net, HYP)
    opt = make_opt(net.parameters())
    xd, tgd = x.to(DEV), tg.to(DEV)
    dev_curve = []
    for _ in range(steps):
        loss, items = crit(net(xd, training=True), tgd)
        loss.backward()
        opt.step()
        opt.zero_grad()
        dev_curve.append(float(items["total_loss"]))
    oopt = make_opt(orc.parameters())
    cpu_curve = []
    for _ in range(steps):
        loss, items = ref_ops.compute_loss(orc(x, True), tg, orc.anchors, nc, mode, HYP)
        loss.backward()
        oopt.step()
        oopt.zero_grad()
        cpu_curve.append(float(items["total_loss"]))
    dev_sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    orc_sd = orc.state_dict()
    # update of the detection-head convolutions (conv5/6/7 of the yolov7 neck: bias + weight), the last layers of the network
    head_keys = [k for k in sd0 if k.startswith(("neck.conv5.", "neck.conv6.", "neck.conv7.")) and sd0[k].dtype.is_floating_point]
    assert head_keys
    du = torch.cat([(dev_sd[k] - sd0[k]).flatten() for k in head_keys]).double()
    ou = torch.cat([(orc_sd[k] - sd0[k]).flatten() for k in head_keys]).double()
    cos_head = float(du @ ou / (du.norm() * ou.norm() + 1e-300))
    allk = [k for k in sd0 if sd0[k].dtype.is_floating_point and "running" not in k]
    da = torch.cat([(dev_sd[k] - sd0[k]).flatten() for k in allk]).double()
    oa = torch.cat([(orc_sd[k] - sd0[k]).flatten() for k in allk]).double()
    rep = dict(dev=dev_curve, oracle=cpu_curve, max_rel_dev=max(abs(a - b) / abs(b) for a, b in zip(dev_curve, cpu_curve)),
               cos_update_heads=cos_head, cos_update_all=float(da @ oa / (da.norm() * oa.norm() + 1e-300)),
               update_norm_ratio_all=float(da.norm() / oa.norm()), band=BAND[opt_name],
               config=f"{ver} {mode} nc={nc} {S}x{S} batch {B}, {opt_name}, {steps} steps, train.py:28-33 init")
    _report(f"{opt_name}_{steps}", rep)
    print("TRAJ", opt_name, json.dumps({k: v for k, v in rep.items() if k not in ("dev", "oracle")}), [round(v, 4) for v in dev_curve[::4]],
          [round(v, 4) for v in cpu_curve[::4]])
    assert all(v == v and abs(v) < 1e6 for v in dev_curve)
    assert dev_curve[-1] < 0.67 * dev_curve[0] and cpu_curve[-1] < 0.67 * cpu_curve[0], (dev_curve[0], dev_curve[-1], cpu_curve[-1])
    assert rep["max_rel_dev"] < BAND[opt_name], rep["max_rel_dev"]
    assert cos_head > 0.9, cos_head
    assert 0.5 < rep["update_norm_ratio_all"] < 2.0
