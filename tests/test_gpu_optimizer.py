"""GPU: the fused optimizer step (ryolo_sgd_nesterov, csrc/elementwise.hip) against torch.optim.SGD(momentum=0.937, nesterov=True)
(train.py:156,201-202), and the reference-style loop — torch.optim.SGD driving the engine's Parameters — against the fused loop."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("gscale,zero", [(1.0, True), (0.125, False)])
def test_fused_sgd_nesterov_matches_torch_optim(gscale, zero):
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S    # noqa: F401
    g = torch.Generator().manual_seed(0)
    n = 4096 + 64
    p0 = torch.randn(n, generator=g)
    p = p0.clone().to(DEV)
    buf = torch.zeros(n, device=DEV)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=0.01, momentum=0.937, nesterov=True)
    for step in range(4):
        grad = torch.randn(n, generator=g)
        gd = grad.clone().to(DEV)
        hip.call("ryolo_sgd_nesterov", p.data_ptr(), gd.data_ptr(), buf.data_ptr(), n, 0.01, 0.937, gscale, 1 if zero else 0, hip.stream())
        ref.grad = grad * gscale
        opt.step()
        torch.testing.assert_close(p.cpu(), ref.detach(), rtol=1e-6, atol=1e-7)
        assert bool((gd == 0).all()) == zero                      # optimizer.zero_grad() fused into the same pass


def test_torch_optim_sgd_on_engine_parameters_equals_the_fused_loop():
    """train.py's own loop (optimizer = torch.optim.SGD(model.parameters(), ...); loss.backward(); optimizer.step();
    optimizer.zero_grad()) on this build's Yolo, against model.runtime().sgd_step: same parameters after three steps (the
    Parameters are views of the flat buffer either way; 1e-6: the fused kernel uses fma contraction differently)."""
    from ryolov4_amd.lib.loss import ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, fill_state, synth_targets
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(1)).to(DEV)
    tg = synth_targets(2, 6, 2, False, seed=2, img_size=96).to(DEV)
    finals = []
    for fused in (True, False):
        m = Yolo(2, CFG, "kfiou", "yolov7")
        m.load_state_dict(fill_state(m.state_dict()))
        m.to(DEV).eval()
        m.frozen_bn = True                                        # well-conditioned (see test_full_network_backward_frozen_bn)
        crit = ComputeKFIoULoss(m, HYP)
        opt = None if fused else torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.937, nesterov=True)
        for step in range(3):
            loss, _ = crit(m(x, training=True), tg)
            loss.backward()
            if fused:
                m.runtime().sgd_step(0.01, 0.937, zero_grad=True)
            else:
                opt.step()
                opt.zero_grad(set_to_none=False)
        finals.append(torch.cat([p.detach().flatten() for p in m.parameters()]).cpu())
    torch.testing.assert_close(finals[0], finals[1], rtol=2e-5, atol=2e-6)


def test_gradient_accumulation_over_two_backwards_is_the_sum():
    """train.py:192-202 accumulates gradients over `accumulate` iterations before optimizer.step(): two backward passes without
    zeroing leave grad(batch 1) + grad(batch 2) in the flat gradient buffer (weight gradients run on their own stream; the second
    backward's `+=` must land after the first's)."""
    from ryolov4_amd.lib.loss import ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, fill_state, synth_targets
    m = Yolo(2, CFG, "kfiou", "yolov7")
    m.load_state_dict(fill_state(m.state_dict()))
    m.to(DEV).eval()
    m.frozen_bn = True
    crit = ComputeKFIoULoss(m, HYP)
    rt = m.runtime()
    batches = [(torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(s)).to(DEV),
                synth_targets(2, 6, 2, False, seed=s, img_size=96).to(DEV)) for s in (1, 2)]
    singles = []
    for x, tg in batches:
        loss, _ = crit(m(x, training=True), tg)
        loss.backward()
        singles.append(rt.gflat.clone())
        rt.gflat.zero_()
    for x, tg in batches:
        loss, _ = crit(m(x, training=True), tg)
        loss.backward()
    torch.testing.assert_close(rt.gflat, singles[0] + singles[1], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("gscale,zero", [(1.0, True), (0.125, False)])
def test_fused_adam_matches_torch_optim(gscale, zero):
    """ryolo_adam (csrc/elementwise.hip) against torch.optim.Adam(lr) with torch's defaults — train.py:153-154, `--optimizer Adam` —
    over 5 steps (bias corrections change every step).  1e-6: torch forms m / denom with one rounding each, the kernel may contract."""
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S    # noqa: F401
    g = torch.Generator().manual_seed(0)
    n = 4096 + 64
    p0 = torch.randn(n, generator=g)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=0.01)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** torch.randint(-4, 1, (n,), generator=g).float())
        gd = grad.clone().to(DEV)
        hip.call("ryolo_adam", p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 0.01, 0.9, 0.999, 1e-8, step, gscale, 1 if zero else 0,
                 hip.stream())
        ref.grad = (grad * gscale) if gscale != 1.0 else grad.clone()
        opt.step()
        torch.testing.assert_close(p.cpu(), ref.detach(), rtol=2e-6, atol=2e-7)
        torch.testing.assert_close(m.cpu(), opt.state[ref]["exp_avg"], rtol=2e-6, atol=1e-9)
        torch.testing.assert_close(v.cpu(), opt.state[ref]["exp_avg_sq"], rtol=2e-6, atol=1e-20)
        assert bool((gd == 0).all()) == zero


def test_adam_rejects_bad_arguments():
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S    # noqa: F401
    t = torch.zeros(64, device=DEV)
    for args in [(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 64, 0.01, 0.9, 0.999, 1e-8, 0, 1.0, 0),        # step < 1
                 (t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 62, 0.01, 0.9, 0.999, 1e-8, 1, 1.0, 0),        # n not a multiple of 4
                 (t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 64, 0.01, 1.0, 0.999, 1e-8, 1, 1.0, 0)]:       # beta1 == 1
        with pytest.raises(RuntimeError):
            hip.call("ryolo_adam", *args, hip.stream())


def test_torch_optim_adam_on_engine_parameters_equals_the_fused_loop():
    """train.py's loop with `--optimizer Adam` (torch.optim.Adam on model.parameters()) against model.runtime().adam_step: same parameters
    after three steps."""
    from ryolov4_amd.lib.loss import ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, fill_state, synth_targets
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(1)).to(DEV)
    tg = synth_targets(2, 6, 2, False, seed=2, img_size=96).to(DEV)
    finals = []
    for fused in (True, False):
        m = Yolo(2, CFG, "kfiou", "yolov7")
        m.load_state_dict(fill_state(m.state_dict()))
        m.to(DEV).eval()
        m.frozen_bn = True
        crit = ComputeKFIoULoss(m, HYP)
        opt = None if fused else torch.optim.Adam(m.parameters(), lr=1e-4)
        for step in range(3):
            loss, _ = crit(m(x, training=True), tg)
            loss.backward()
            if fused:
                m.runtime().adam_step(1e-4, zero_grad=True)
            else:
                opt.step()
                opt.zero_grad(set_to_none=False)
        finals.append(torch.cat([p.detach().flatten() for p in m.parameters()]).cpu())
    # Adam's first steps move every parameter by ~lr whatever the gradient's size: where a gradient element is ~0 the sign of m / sqrt(v) is
    # decided by rounding, so compare through the update size
    d = (finals[0] - finals[1]).abs()
    assert float(d.max()) <= 2.5e-4 and float((d > 2e-6).float().mean()) < 2e-3, (float(d.max()), float((d > 2e-6).float().mean()))
