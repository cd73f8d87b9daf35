"""BatchNorm-backward sums taken in the epilogue of the launch that completes an activation gradient (ConvGemmParams.bstat) against the
stand-alone reduce pass.  With BatchNorm frozen to its running statistics the sums feed ONLY dgamma / dbeta (no coupling back into the
data path), so every other gradient must stay bit-identical and the BatchNorm parameter gradients may differ by fp32 summation order
only.  In batch-statistics mode the sums steer every earlier layer; there the per-block oracle tests (test_gpu_blocks.py, run with the
fusion on) and the whole-network noise-floor tests carry the check, and this file bounds fused vs unfused on a short chain."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _grads(ver, mode, fuse, frozen, B=4, S=192):
    import bench
    from ryolov4_amd.lib.loss import ComputeCSLLoss, ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, synth_batch
    torch.manual_seed(0)
    m = Yolo(16, CFG, mode, ver)
    m.apply(bench.weights_init_normal)
    m.to(DEV)
    if frozen:
        m.eval()
        m.frozen_bn = True
    rt = m.runtime(DEV)
    rt.fuse_bn_reduce = fuse
    rt.fuse_bn_max_elems = int(1e12) if fuse else 0              # (the fold is off by default since r04: runtime.py)
    imgs, tg = synth_batch(B, S, 16, mode == "csl", seed=5)
    imgs, tg = imgs.to(DEV), tg.to(DEV)
    crit = (ComputeCSLLoss if mode == "csl" else ComputeKFIoULoss)(m, HYP)
    heads = m(imgs, training=True)
    keep = [h.detach().clone() for h in heads]
    loss, _ = crit(heads, tg)
    loss.backward()
    g = rt.graph(B, S, S, True, frozen=frozen)
    named = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    return keep, float(loss), named, g


@pytest.mark.parametrize("ver,mode", [("yolov7", "kfiou"), ("yolov4", "csl"), ("yolov5", "kfiou")])
def test_frozen_bn_only_the_bn_parameter_gradients_move_and_only_by_summation_order(ver, mode):
    ha, la, ga, gra = _grads(ver, mode, True, True)
    hb, lb, gb, grb = _grads(ver, mode, False, True)
    assert gra.n_bstat >= 60 and grb.n_bstat == 0                    # most Conv blocks are covered
    assert la == lb and all(torch.equal(x, y) for x, y in zip(ha, hb))
    worst = 0.0
    for n in ga:
        a, b = ga[n], gb[n]
        if a.dim() == 1 and (n.endswith(".weight") or n.endswith(".bias")) and not torch.equal(a, b):
            err = float((a - b).abs().max() / (b.abs().max() + 1e-20))   # BatchNorm gamma / beta (1-D parameters)
            worst = max(worst, err)
            assert err < 2e-5, (n, err)
        else:
            assert torch.equal(a, b), n
    assert worst > 0.0                                                # the fused path really ran (different summation order)


def test_batch_statistics_mode_fused_vs_unfused_on_the_first_layers():
    """Train-mode BatchNorm: the sums steer dY of every layer.  Same inputs, fusion on / off: identical forward and loss; the gradients of
    the layers next to the loss (at most a couple of folded BatchNorms between them and the head maps) agree to 2e-2 — fp32 summation
    order seen through one bf16 rounding of dY — and the whole gradient vector stays finite and positively correlated (at random
    initialisation ~100 batch-statistics BatchNorms amplify any rounding difference, see test_gpu_parity_e2e.py)."""
    ha, la, ga, _ = _grads("yolov7", "kfiou", True, False)
    hb, lb, gb, _ = _grads("yolov7", "kfiou", False, False)
    assert la == lb and all(torch.equal(x, y) for x, y in zip(ha, hb))
    # detection-head convolutions and the Conv blocks right under them: at most one fused BatchNorm between them and the loss
    names = [n for n in ga if n.startswith("neck.") or n.startswith("head.")] or list(ga)
    tail = names[-12:]
    for n in tail:
        a, b = ga[n].float(), gb[n].float()
        rel = float((a - b).norm() / (b.norm() + 1e-20))
        assert rel < 2e-2, (n, rel)
    va = torch.cat([ga[n].flatten().float() for n in ga])
    vb = torch.cat([gb[n].flatten().float() for n in gb])
    assert bool(torch.isfinite(va).all())
    cos = float(torch.dot(va, vb) / (va.norm() * vb.norm()))
    assert cos > 0.0, cos
