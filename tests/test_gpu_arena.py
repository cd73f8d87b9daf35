"""Liveness-placed buffers (engine/arena.py) against one-tensor-per-buffer plans: every kernel is deterministic, so the two must agree
BIT FOR BIT on head maps, losses and every parameter gradient — any difference is a buffer handed out while still in use."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(ver, mode, reuse, train, steps, B=4, S=256, lag=None):
    import bench
    from ryolov4_amd.lib.loss import ComputeCSLLoss, ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, synth_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = Yolo(16, CFG, mode, ver)
    m.apply(bench.weights_init_normal)
    m.to(dev)
    rt = m.runtime(dev)
    rt.buffer_reuse = reuse
    if lag is not None:
        rt.wgrad_lag = lag
    imgs, tg = synth_batch(B, S, 16, mode == "csl", seed=3)
    imgs, tg = imgs.to(dev), tg.to(dev)
    out = []
    if not train:
        m.eval()
        with torch.no_grad():
            for _ in range(steps):
                heads, inf = m(imgs, training=False)
                out.append([h.clone() for h in heads] + [inf.clone()])
        return out, rt
    crit = (ComputeCSLLoss if mode == "csl" else ComputeKFIoULoss)(m, HYP)
    for _ in range(steps):
        heads = m(imgs, training=True)
        keep = [h.detach().clone() for h in heads]
        loss, _ = crit(heads, tg)
        loss.backward()
        out.append(keep + [loss.detach().clone(), rt.gflat.clone()])
        rt.sgd_step(0.01)
    return out, rt


@pytest.mark.parametrize("ver,mode", [("yolov7", "kfiou"), ("yolov4", "csl"), ("yolov5", "kfiou")])
def test_training_steps_bitwise_equal_with_and_without_reuse(ver, mode):
    a, rta = _run(ver, mode, True, True, 3)
    b, _ = _run(ver, mode, False, True, 3)
    g = rta.graph(4, 256, 256, True)
    assert g.layout is not None and g.layout.total < 0.62 * g.layout.sum_bytes
    for step, (x, y) in enumerate(zip(a, b)):
        for i, (p, q) in enumerate(zip(x, y)):
            assert torch.equal(p, q), (ver, step, i, float((p.float() - q.float()).abs().max()))
    assert bool(torch.isfinite(a[-1][-1]).all())


def test_lag_one_is_still_exact():
    """The tightest lag (the main stream waits for every weight gradient before the next one is enqueued) packs hardest."""
    a, rta = _run("yolov7", "kfiou", True, True, 2, lag=1)
    b, _ = _run("yolov7", "kfiou", False, True, 2)
    for x, y in zip(a, b):
        for p, q in zip(x, y):
            assert torch.equal(p, q)


@pytest.mark.parametrize("ver", ["yolov7", "yolov4"])
def test_inference_bitwise_equal_with_and_without_reuse(ver):
    a, rta = _run(ver, "kfiou", True, False, 2)
    b, _ = _run(ver, "kfiou", False, False, 2)
    g = rta.graph(4, 256, 256, False)
    assert g.layout.total < 0.35 * g.layout.sum_bytes
    for x, y in zip(a, b):
        for p, q in zip(x, y):
            assert torch.equal(p, q)


def test_reuse_survives_dirty_arena_and_serial_replay():
    """Fill the arena with NaN between steps (nothing may rely on stale contents) and replay with all streams serialized."""
    a, rt = _run("yolov7", "kfiou", True, True, 1)
    import bench
    from ryolov4_amd.lib.loss import ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, synth_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = Yolo(16, CFG, "kfiou", "yolov7")
    m.apply(bench.weights_init_normal)
    m.to(dev)
    rt = m.runtime(dev)
    imgs, tg = synth_batch(4, 256, 16, False, seed=3)
    imgs, tg = imgs.to(dev), tg.to(dev)
    crit = ComputeKFIoULoss(m, HYP)
    g = rt.graph(4, 256, 256, True)
    g.arena.view(torch.bfloat16).fill_(float("nan"))
    g.serial = True
    heads = m(imgs, training=True)
    keep = [h.detach().clone() for h in heads]
    loss, _ = crit(heads, tg)
    loss.backward()
    for p, q in zip(keep + [loss.detach(), rt.gflat], a[0]):
        assert torch.equal(p, q)
