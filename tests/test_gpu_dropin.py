"""GPU: the drop-in boundary itself (VERDICT r1 "install_dropin() has no test").  A fresh interpreter calls
ryolov4_amd.install_dropin() and then runs the reference's OWN import lines and loops, restated from train.py / test.py:

    from model.yolo import Yolo                      train.py:14   test.py:10
    from lib.loss import ComputeCSLLoss, ComputeKFIoULoss     train.py:16   test.py:12
    from lib.general import post_process             test.py:11

* train.py:150-158,186-202: nominal batch 64 accumulation, warm-up interpolation of `accumulate` and the learning rate,
  torch.optim.SGD(momentum 0.937, nesterov) + LambdaLR, `loss.backward()` / `optimizer.step()` / `optimizer.zero_grad()`;
* test.py:188-207: eval forward, loss under no_grad, post_process, `targets[:, 2:6] *= img_size`, batch statistics, AP.
The same loops run on the torch-CPU oracle with the same weights and inputs; losses must agree within the bf16 noise of a
random-init train-mode network (3 %), the evaluation loss within 1 %."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, math, sys
import numpy as np
import torch
import ryolov4_amd
ryolov4_amd.install_dropin()
from model.yolo import Yolo                                   # the reference's import lines
from lib.loss import ComputeCSLLoss, ComputeKFIoULoss
from lib.general import post_process
from ryolov4_amd.lib.evaluate import get_batch_statistics, calculate_eval_stats
from ryolov4_amd.synth import CFG, fill_state, synth_batch
from oracle import ref_model, ref_ops                          # the checker

hyp_cfg = {"fl_gamma": 0.0, "box": 0.05, "obj": 1.0, "obj_pw": 1.0, "cls": 0.5, "cls_pw": 1.0, "warmup_prop": 0.1, "lrf": 0.01}
mode, ver, nc, img_size, batch_size, epochs, lr0 = "kfiou", "yolov7", 2, 96, 2, 2, 0.01
device = torch.device("cuda:0")
model = Yolo(nc, CFG, mode, ver)
sd = fill_state(model.state_dict())
model.load_state_dict(sd)
model = model.to(device)
orc = ref_model.Yolo(nc, CFG, mode, ver)
orc.load_state_dict(sd)
compute_loss = ComputeKFIoULoss(model, hyp_cfg)
assert list(compute_loss.loss_items) == ["reg_loss", "conf_loss", "cls_loss", "total_loss"]     # train.py:178 reads the keys before the first call
batches = [synth_batch(batch_size, img_size, nc, False, seed=100 + i, per_image=5) for i in range(3)]

def one_cycle(y1=0.0, y2=1.0, steps=100):                     # lib/scheduler.py as used at train.py:160
    return lambda x: ((1 - math.cos(x * math.pi / steps)) / 2) * (y2 - y1) + y1

def train(model, loss_fn, to_dev):
    model.train()
    nbs = 64
    accumulate = max(round(nbs / batch_size), 1)
    optimizer = torch.optim.SGD(model.parameters(), lr=lr0, momentum=0.937, nesterov=True)
    num_iters_per_epoch = len(batches)
    nw = max(int((epochs * num_iters_per_epoch) * hyp_cfg["warmup_prop"]), 4)          # (the reference's floor of 1000 shortened to 4)
    lf = one_cycle(1, hyp_cfg["lrf"], epochs)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lf)
    initial_lr = optimizer.param_groups[0]["initial_lr"]
    log = []
    for epoch in range(epochs):
        for batch, (imgs, targets) in enumerate(batches):
            global_step = num_iters_per_epoch * epoch + batch + 1
            imgs, targets = to_dev(imgs), to_dev(targets)
            if global_step <= nw:
                xi = [0, nw]
                accumulate = max(1, np.interp(global_step, xi, [1, nbs / batch_size]).round())
                optimizer.param_groups[0]["lr"] = np.interp(global_step, xi, [0.0, initial_lr * lf(epoch)])
            outputs = model(imgs, training=True)
            loss, loss_items = loss_fn(outputs, targets)
            loss.backward()
            if global_step % accumulate == 0:
                optimizer.step()
                optimizer.zero_grad()
            log.append(float(loss_items["total_loss"]))
        scheduler.step()
    return log

dev_log = train(model, compute_loss, lambda t: t.to(device))
cpu_log = train(orc, lambda o, t: ref_ops.compute_loss(o, t, orc.anchors, nc, mode, hyp_cfg), lambda t: t)

def evaluate(model, loss_fn, pp, stats_fn, to_dev):
    model.eval()
    iouv = torch.linspace(0.5, 0.95, 10)
    niou = iouv.numel()
    stats, seen, total = [], 0, {}
    for imgs, targets in batches:
        imgs, targets = to_dev(imgs), to_dev(targets.clone())
        seen += len(imgs)
        with torch.no_grad():
            outputs, infer_outputs = model(imgs, training=False)
            _, loss_items = loss_fn(outputs, targets)
            infer_outputs = pp(infer_outputs, conf_thres=0.05, iou_thres=0.65)
            for item in loss_items:
                total[item] = total.get(item, 0.0) + float(loss_items[item])
        targets[:, 2:6] *= img_size
        stats += stats_fn(infer_outputs, targets, iouv, niou)
    ndet = sum(len(s[1]) for s in stats)
    res = None
    if stats:
        cat = [np.concatenate(x, 0) for x in list(zip(*stats))]
        res = calculate_eval_stats(cat, nc)
    return total, seen, ndet, (float(res[-1]) if res is not None else None)

dev_eval = evaluate(model, compute_loss, post_process, get_batch_statistics, lambda t: t.to(device))
cpu_eval = evaluate(orc, lambda o, t: ref_ops.compute_loss(o, t, orc.anchors, nc, mode, hyp_cfg), ref_ops.post_process,
                    ref_ops.get_batch_statistics, lambda t: t)
print("RESULT " + json.dumps(dict(dev_log=dev_log, cpu_log=cpu_log, dev_eval=[dev_eval[0], dev_eval[1], dev_eval[2], dev_eval[3]],
                                  cpu_eval=[cpu_eval[0], cpu_eval[1], cpu_eval[2], cpu_eval[3]],
                                  grads_are_flat_views=all(p.grad is None or p.grad.data_ptr() == model.runtime().grad_ptr(p) for p in model.parameters()))))
'''


def test_reference_train_and_test_loops_through_install_dropin():
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    d = json.loads(line[len("RESULT "):])
    assert len(d["dev_log"]) == 6 and all(x == x and abs(x) < 1e6 for x in d["dev_log"])
    for a, b in zip(d["dev_log"], d["cpu_log"]):
        assert abs(a - b) < 3e-2 * abs(b), (d["dev_log"], d["cpu_log"])
    dev_items, cpu_items = d["dev_eval"][0], d["cpu_eval"][0]
    assert set(dev_items) == set(cpu_items) == {"reg_loss", "conf_loss", "cls_loss", "total_loss"}
    for k in dev_items:
        assert abs(dev_items[k] - cpu_items[k]) < 1e-2 * max(abs(cpu_items[k]), 1e-6), (k, dev_items[k], cpu_items[k])
    assert d["dev_eval"][1] == d["cpu_eval"][1] == 6
    assert d["grads_are_flat_views"]
