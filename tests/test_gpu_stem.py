"""Direct first-layer kernels (csrc/stem.hip) through the C ABI against torch's fp32 conv2d on the same image with
bf16-rounded weights: forward (raw / BatchNorm statistics / folded BN + activation) and the weight gradient.
Tolerance: forward rel 2^-8 (bf16 inputs and output rounding); weight gradient 2e-3 relative (bf16 operands, fp32 sums)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(B, H, W, seed=0):
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    S.check_layouts()
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g).cuda()
    w = (torch.randn(32, 3, 3, 3, generator=g) * 0.2)
    wf = torch.zeros(32, 32)
    wf[:, :27] = w.permute(0, 2, 3, 1).reshape(32, 27)             # k = (r*3 + s)*3 + c
    return hip, S, img, w.cuda(), wf.to(torch.bfloat16).cuda()


@pytest.mark.parametrize("B,H,W", [(2, 32, 48), (3, 17, 16), (1, 64, 64)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_stem_forward(B, H, W, epi):
    hip, S, img, w, wf = _setup(B, H, W)
    ld = 40
    y = torch.full((B * H * W, ld), 7.0, dtype=torch.bfloat16, device="cuda")
    rows, wsb = S.I(), S.Z()
    hip.call("ryolo_stem3x3_plan", B, H, W, 32, rows, wsb)
    stats = torch.zeros(rows.value, 2, 32, device="cuda")
    co = torch.rand(4, 32, device="cuda") + 0.5
    p = S.StemParams()
    p.img, p.NB, p.H, p.W = img.data_ptr(), B, H, W
    p.wf, p.Cout, p.epi, p.out, p.ldC = wf.data_ptr(), 32, epi, y.data_ptr(), ld
    p.stats, p.scale, p.shift, p.act = stats.data_ptr(), co.data_ptr() + 2 * 32 * 4, co.data_ptr() + 3 * 32 * 4, 3
    hip.call("ryolo_stem3x3_fwd", p, hip.stream())
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(img.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, 32)
    if epi == 2:
        u = ref * co[2] + co[3]
        ref = u * torch.sigmoid(u)
    got = y[:, :32].float()
    assert bool(((got - ref).abs() <= 2.0 ** -7 * ref.abs() + 1e-2).all()), float((got - ref).abs().max())
    assert bool((y[:, 32:] == 7.0).all()), "wrote outside its channel slice"
    if epi == 1:
        assert torch.allclose(stats[:, 0].sum(0), got.sum(0), rtol=1e-4, atol=1e-2)
        assert torch.allclose(stats[:, 1].sum(0), (got * got).sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("B,H,W", [(2, 32, 48), (3, 17, 16), (2, 64, 128)])
def test_stem_wgrad(B, H, W):
    hip, S, img, w, wf = _setup(B, H, W, seed=3)
    g = torch.Generator().manual_seed(9)
    ld = 48
    dy = (torch.randn(B * H * W, ld, generator=g) * 0.1).to(torch.bfloat16).cuda()
    rows, wsb = S.I(), S.Z()
    hip.call("ryolo_stem3x3_plan", B, H, W, 32, rows, wsb)
    ws = torch.empty(wsb.value // 4, device="cuda")
    scratch = torch.full((32, 32), 3.0, device="cuda")
    q = S.StemWgradParams()
    q.img, q.NB, q.H, q.W = img.data_ptr(), B, H, W
    q.dY, q.ldY, q.Cout, q.scratch, q.workspace = dy.data_ptr(), ld, 32, scratch.data_ptr(), ws.data_ptr()
    hip.call("ryolo_stem3x3_wgrad", q, hip.stream())
    torch.cuda.synchronize()
    w0 = torch.zeros(32, 3, 3, 3, device="cuda", requires_grad=True)
    out = torch.nn.functional.conv2d(img.to(torch.bfloat16).float(), w0, padding=1)
    out.backward(dy[:, :32].float().view(B, H, W, 32).permute(0, 3, 1, 2))
    ref = w0.grad.permute(0, 2, 3, 1).reshape(32, 27)
    assert float((scratch[:, :27] - ref).norm() / ref.norm()) < 2e-3
    assert float(scratch[:, 27:].abs().max()) == 0.0


@pytest.mark.parametrize("ver", ["yolov7", "yolov4"])
def test_stem_wgrad_with_fused_bn_backward_matches_the_unfused_plan(ver):
    """BatchNorm + activation backward applied INSIDE the stem weight-gradient kernel (StemWgradParams.y set; SiLU for yolov7, Mish
    for yolov4) vs the plan that materialises the raw gradient with ryolo_bn_act_bwd's apply pass: same algebra, same bf16 rounding
    point -> the first conv's weight gradient and its BatchNorm gradients agree to 2e-3 relative (an FMA-contraction difference may
    flip a bf16 ulp here and there), everything downstream is untouched."""
    from ryolov4_amd.lib.loss import ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, fill_state, synth_targets
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(11)).cuda()
    tg = synth_targets(2, 6, 2, False, seed=2, img_size=96).cuda()
    grads = []
    for fuse in (True, False):
        m = Yolo(2, CFG, "kfiou", ver)
        m.load_state_dict(fill_state(m.state_dict()))
        m.cuda().train()
        m.runtime().fuse_stem_bn = fuse
        crit = ComputeKFIoULoss(m, HYP)
        loss, _ = crit(m(x, training=True), tg)
        loss.backward()
        names = [n for _, _, n in m.runtime().graph(2, 96, 96, True).bwd]
        assert "ryolo_stem3x3_wgrad" in names
        ps = list(m.parameters())
        grads.append([ps[0].grad.clone(), ps[1].grad.clone(), ps[2].grad.clone(), ps[3].grad.clone()])
    for a, b in zip(*grads):
        assert a.abs().sum() > 0
        err = float((a - b).norm() / b.norm())
        assert err < 2e-3, err
