"""smoke(): one tiny forward + loss + backward + SGD step of the flagship configuration (yolov7 kfiou) on cuda:0, with the
eval-mode forward checked against the torch-CPU oracle (imported by __graft_entry__.smoke only)."""
import torch


def run(dev):
    from oracle import ref_model
    from .lib.loss import ComputeKFIoULoss
    from .model.yolo import Yolo
    from .synth import CFG, HYP, fill_state, synth_batch
    nc = 16
    net = Yolo(nc, CFG, "kfiou", "yolov7")
    sd = fill_state(net.state_dict())
    net.load_state_dict(sd)
    net.to(dev)
    orc = ref_model.Yolo(nc, CFG, "kfiou", "yolov7")
    orc.load_state_dict(sd)
    imgs, tg = synth_batch(2, 64, nc, False, seed=1, per_image=4)
    # eval forward vs oracle (bf16 storage: rel-L2 <= 1e-2)
    net.eval(); orc.eval()
    with torch.no_grad():
        outs, infer = net(imgs.to(dev), training=False)
        ref = orc.head_maps(imgs)
    for a, b in zip(outs, ref):
        B_, _, gs, _ = b.shape
        b5 = b.view(B_, 18, -1, gs, gs).permute(0, 1, 3, 4, 2)
        err = float((a.cpu() - b5).norm() / b5.norm())
        assert err < 1e-2, f"eval forward differs from oracle: {err}"
    assert infer.shape == (2, 18 * (64 + 16 + 4), nc + 6)
    # one training step
    net.train()
    crit = ComputeKFIoULoss(net, HYP)
    outs = net(imgs.to(dev), training=True)
    loss, items = crit(outs, tg.to(dev))
    loss.backward()
    rt = net.runtime()
    g = rt.gflat
    assert torch.isfinite(loss).all() and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    before = rt.flat.clone()
    rt.sgd_step(0.01)
    assert not torch.equal(before, rt.flat) and float(rt.gflat.abs().sum()) == 0.0
    torch.cuda.synchronize()
