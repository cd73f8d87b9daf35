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


@pytest.mark.parametrize("B,H,W", [(2, 32, 64), (3, 17, 32), (1, 96, 96)])
@pytest.mark.parametrize("act,frozen", [(3, 0), (1, 0), (3, 1)])
def test_stem_fused_backward_vs_autograd(B, H, W, act, frozen):
    """ryolo_stem3x3_bwd: the layer's whole backward in one pass over dz with the conv output RECOMPUTED from the image, against torch
    autograd through conv2d (bf16-rounded operands, output rounded to bf16 with a straight-through gradient) -> BatchNorm with the
    batch statistics of that output (frozen: fixed affine map) -> SiLU / Mish.  Weight gradient 1e-2 relative (g and the patches are
    bf16 MFMA operands), BatchNorm gradients 2e-3."""
    hip, S, img, w, wf = _setup(B, H, W, seed=5)
    g = torch.Generator().manual_seed(21)
    ld = 40
    dz = (torch.randn(B * H * W, ld, generator=g) * 0.1).to(torch.bfloat16).cuda()
    gamma = (torch.rand(32, generator=g) + 0.5).cuda().requires_grad_()
    beta = (torch.randn(32, generator=g) * 0.1).cuda().requires_grad_()
    w0 = w.to(torch.bfloat16).float().requires_grad_()
    y = torch.nn.functional.conv2d(img.to(torch.bfloat16).float(), w0, padding=1)
    yr = y + (y.to(torch.bfloat16).float() - y).detach()
    if frozen:
        mean, var = torch.full((32,), 0.05, device="cuda"), torch.full((32,), 0.8, device="cuda")
    else:
        mean, var = yr.mean((0, 2, 3)), yr.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    u = (yr - mean[None, :, None, None]) * (invstd * gamma)[None, :, None, None] + beta[None, :, None, None]
    z = u * torch.sigmoid(u) if act == 3 else torch.nn.functional.mish(u)
    (z * dz[:, :32].float().view(B, H, W, 32).permute(0, 3, 1, 2)).sum().backward()
    co = torch.stack([mean.detach(), invstd.detach(), (gamma * invstd).detach(), (beta - mean * gamma * invstd).detach()]).contiguous()
    wsb = S.Z()
    hip.call("ryolo_stem3x3_bwd_plan", B, H, W, 32, wsb)
    ws = torch.empty(wsb.value // 4, device="cuda")
    dW = torch.full((32, 3, 3, 3), 0.5, device="cuda")
    dg, db = torch.full((32,), 0.25, device="cuda"), torch.full((32,), -0.25, device="cuda")
    q = S.StemBwdParams()
    q.img, q.NB, q.H, q.W = img.data_ptr(), B, H, W
    q.dz, q.lddz, q.wf, q.co, q.act, q.frozen = dz.data_ptr(), ld, wf.data_ptr(), co.data_ptr(), act, frozen
    q.workspace, q.dW, q.dgamma, q.dbeta = ws.data_ptr(), dW.data_ptr(), dg.data_ptr(), db.data_ptr()
    hip.call("ryolo_stem3x3_bwd", q, hip.stream())
    torch.cuda.synchronize()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(dW - 0.5, w0.grad) < 1e-2, rel(dW - 0.5, w0.grad)          # accumulated onto what was there
    assert rel(dg - 0.25, gamma.grad) < 2e-3 and rel(db + 0.25, beta.grad) < 2e-3, (rel(dg - 0.25, gamma.grad), rel(db + 0.25, beta.grad))


def test_stem_stats_only_and_rounded_affine_forward_match_the_two_pass_path():
    """Training forward without the raw output: the statistics-only pass gives the statistics of the bf16-rounded conv output, and
    epilogue 5 (BatchNorm + activation on the rounded accumulator) is BIT-identical to storing it and running ryolo_bn_act_fwd."""
    hip, S, img, w, wf = _setup(2, 32, 64, seed=7)
    B, H, W = 2, 32, 64
    rows = S.I()
    hip.call("ryolo_stem3x3_plan", B, H, W, 32, rows, None)
    y = torch.empty(B * H * W, 32, dtype=torch.bfloat16, device="cuda")
    st = [torch.zeros(rows.value, 2, 32, device="cuda") for _ in range(2)]
    co = torch.rand(4, 32, device="cuda") + 0.5

    def run(epi, out, stats):
        p = S.StemParams()
        p.img, p.NB, p.H, p.W = img.data_ptr(), B, H, W
        p.wf, p.Cout, p.epi, p.out, p.ldC = wf.data_ptr(), 32, epi, out.data_ptr() if out is not None else None, 32
        p.stats = stats.data_ptr() if stats is not None else None
        p.scale, p.shift, p.act = co.data_ptr() + 2 * 32 * 4, co.data_ptr() + 3 * 32 * 4, 3
        hip.call("ryolo_stem3x3_fwd", p, hip.stream())
    run(1, y, st[0])
    run(1, None, st[1])
    assert torch.equal(st[0], st[1])
    z5 = torch.empty_like(y)
    run(5, z5, None)
    z2 = torch.empty_like(y)
    q = S.BnActParams()
    q.y1, q.ld1, q.co1, q.z, q.ldz, q.M, q.C, q.act = y.data_ptr(), 32, co.data_ptr(), z2.data_ptr(), 32, B * H * W, 32, 3
    hip.call("ryolo_bn_act_fwd", q, hip.stream())
    torch.cuda.synchronize()
    assert torch.equal(z5.view(torch.int16), z2.view(torch.int16))


@pytest.mark.parametrize("ver", ["yolov7", "yolov4"])
def test_stem_backward_variants_agree_in_the_plan(ver):
    """Three plans of the first layer's backward: (a) recompute [default] — one fused pass, raw output never stored; (b) BatchNorm +
    activation backward applied inside the weight-gradient kernel (raw output stored); (c) the generic apply pass + plain weight
    gradient.  Same algebra, different rounding points -> the first conv's weight gradient and its BatchNorm gradients agree to 1e-2 /
    2e-3 relative; the forward (hence every other gradient) is bit-identical."""
    from ryolov4_amd.lib.loss import ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, fill_state, synth_targets
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(11)).cuda()
    tg = synth_targets(2, 6, 2, False, seed=2, img_size=96).cuda()
    grads, losses = [], []
    for recompute, fuse in ((True, True), (False, True), (False, False)):
        m = Yolo(2, CFG, "kfiou", ver)
        m.load_state_dict(fill_state(m.state_dict()))
        m.cuda().train()
        m.runtime().stem_recompute, m.runtime().fuse_stem_bn = recompute, fuse
        crit = ComputeKFIoULoss(m, HYP)
        loss, _ = crit(m(x, training=True), tg)
        loss.backward()
        names = [n for _, _, n in m.runtime().graph(2, 96, 96, True).bwd]
        assert ("ryolo_stem3x3_bwd" in names) == recompute and ("ryolo_stem3x3_wgrad" in names) == (not recompute)
        ps = list(m.parameters())
        grads.append([p.grad.clone() for p in ps])
        losses.append(float(loss))
    assert losses[0] == losses[1] == losses[2]
    for k in (0, 1):
        for j, (a, b) in enumerate(zip(grads[k], grads[2])):
            if j >= 3:
                assert torch.equal(a, b), j                              # everything behind the first layer: untouched
            else:
                assert a.abs().sum() > 0
                err = float((a - b).norm() / b.norm())
                assert err < (1e-2 if j == 0 else 2e-3), (k, j, err)
