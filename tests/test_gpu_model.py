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
from ryolov4_amd.synth import CFG, HYP, fill_state, synth_targets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-12))


# ------------------------------------------------------------------------------------------------ block-level net
def _tiny(blocks_mod, heads):
    """A small network exercising every block type; `blocks_mod` is either the product's blocks or the oracle's."""
    B = blocks_mod

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.c0 = B.Conv(3, 32, 3, 1, "swish")
            self.c1 = B.Conv(32, 64, 3, 2, "mish")
            self.b1 = B.Bottleneck(64, 64, True, e=1.0, act="mish")
            self.c2 = B.Conv(64, 64, 1, 1, "leaky")
            self.e1 = B.ELAN1(64, 128)
            self.mc = B.MaxConv(128)
            self.csp = B.CSP(128, 128, 2)
            self.spp = B.SPPCSPC(128, 64)
            self.c3 = B.Conv(64, 128, 1, 1, "swish")
            self.e2 = B.ELAN2(256, 64)
            self.rep = B.RepConv(64, 128)
            self.ia = B.ImplicitA(128)
            self.h1 = B.Conv(128, heads, 1, 1, "linear", bn=False, bias=True)
            self.im = B.ImplicitM(heads)
            self.spp4 = B.SPP(128, 64)
            self.sppf = B.SPPF(64, 64)
            self.c5 = B.C5(64, 64)
            self.c33 = B.C3(64, 64, 2, shortcut=False)
            self.h2 = B.Conv(64, heads, 1, 1, "linear", bn=False, bias=True)
    return Tiny


NA, ATTRS = 3, 8


def _product_tiny():
    from ryolov4_amd.model import blocks as B
    T = _tiny(B, NA * ATTRS)

    class P(T):
        _grad_hook = None

        def _emit(self, g):
            x = self.c0.emit(g, None, stem=True)
            x = self.c2.emit(g, self.b1.emit(g, self.c1.emit(g, x)))
            e1 = self.e1.emit(g, x)                                   # [B,128,H/2]
            m = self.csp.emit(g, self.mc.emit(g, e1))                 # [B,128,H/4]
            s = self.spp.emit(g, m)                                   # [B,64,H/4]
            cat = g.new(e1.N, e1.H, e1.W, 256)
            g.upsample(self.c3.emit(g, s), out=cat.slice(128, 128))
            # e1 is also concatenated: route through a 1x1-free copy by planning a second ELAN input slice
            g.copy_slice(e1, cat.slice(0, 128))
            y = self.e2.emit(g, cat)
            g.head(self.h1.conv[0], self.rep.emit(g, y), NA, ATTRS, implicit_a=self.ia.implicit, implicit_m=self.im.implicit)
            t = self.c33.emit(g, self.c5.emit(g, self.sppf.emit(g, self.spp4.emit(g, m))))
            g.head(self.h2.conv[0], t, NA, ATTRS)
    return P()


def _oracle_tiny():
    T = _tiny(ref_model, NA * ATTRS)

    class O(T):
        def forward(self, x):
            x = self.c2(self.b1(self.c1(self.c0(x))))
            e1 = self.e1(x)
            m = self.csp(self.mc(e1))
            s = self.spp(m)
            cat = torch.cat((e1, nn.functional.interpolate(self.c3(s), scale_factor=2)), 1)
            o1 = self.im(self.h1(self.ia(self.rep(self.e2(cat)))))
            o2 = self.h2(self.c33(self.c5(self.sppf(self.spp4(m)))))
            return [o1, o2]
    return O()


def _to_5d(t):
    b, c, h, w = t.shape
    return t.view(b, NA, ATTRS, h, w).permute(0, 1, 3, 4, 2).contiguous()


def test_blocks_forward_backward_vs_oracle():
    from ryolov4_amd.engine.runtime import NetFunction, Runtime
    torch.manual_seed(0)
    orc = _oracle_tiny()
    sd = fill_state(orc.state_dict())
    orc.load_state_dict(sd)
    prod = _product_tiny()
    assert list(prod.state_dict().keys()) == list(sd.keys())
    prod.load_state_dict(sd)
    prod.to(DEV).train()
    orc.train()
    x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(3))
    gw = [torch.randn(2, NA, 16, 16, ATTRS, generator=torch.Generator().manual_seed(4)),
          torch.randn(2, NA, 8, 8, ATTRS, generator=torch.Generator().manual_seed(5))]
    # oracle
    outs_o = [_to_5d(o) for o in orc(x)]
    sum((o * g).sum() for o, g in zip(outs_o, gw)).backward()
    # product
    rt = Runtime(prod, torch.device(DEV))
    g = rt.graph(2, 32, 32, True)
    flag = torch.zeros(1, requires_grad=True)
    outs_p = NetFunction.apply(x.to(DEV), flag, rt, g)
    for a, b in zip(outs_p, outs_o):
        assert rel(a.cpu(), b) < 2e-2, rel(a.cpu(), b)
    sum((o * gg.to(DEV)).sum() for o, gg in zip(outs_p, gw)).backward()
    worst = {}
    for (n, p), (_, q) in zip(prod.named_parameters(), orc.named_parameters()):
        assert p.grad is not None, n
        worst[n] = rel(p.grad.cpu(), q.grad)
    bad = {k: v for k, v in worst.items() if v > 6e-2}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
    # BatchNorm running statistics and num_batches_tracked follow nn.BatchNorm2d
    for (n, b), (_, q) in zip(prod.named_buffers(), orc.named_buffers()):
        if n.endswith("num_batches_tracked"):
            assert int(b) == int(q) == 1
        else:
            assert rel(b.cpu(), q) < 1e-2, n


# ------------------------------------------------------------------------------------------------ full networks
@pytest.mark.parametrize("ver", ["yolov4", "yolov5", "yolov7"])
@pytest.mark.parametrize("mode", ["csl", "kfiou"])
def test_full_network_vs_golden_and_oracle(golden_dir, ver, mode):
    from ryolov4_amd.model.yolo import Yolo
    g2 = np.load(os.path.join(golden_dir, "g2_fullnet.npz"))
    net = Yolo(2, CFG, mode, ver)
    sd = fill_state(net.state_dict())
    net.load_state_dict(sd, strict=True)
    net.to(DEV)
    orc = ref_model.Yolo(2, CFG, mode, ver)
    orc.load_state_dict(sd, strict=True)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    na = 3 if mode == "csl" else 18
    for train in (True, False):
        net.train(train)
        orc.train(train)
        with torch.no_grad():
            res = net(x.to(DEV), training=train)
            hm_o = orc.head_maps(x)
        outs = res if train else res[0]
        tag = f"{ver}_{mode}_{'train' if train else 'eval'}"
        for k, (a, b) in enumerate(zip(outs, hm_o)):
            B_, _, gs, _ = b.shape
            b5 = b.view(B_, na, -1, gs, gs).permute(0, 1, 3, 4, 2)
            assert rel(a.cpu(), b5) < 3e-2, (tag, k, rel(a.cpu(), b5))
            # golden samples captured from the imported reference: same elements of the NCHW map
            nchw = a.cpu().permute(0, 1, 4, 2, 3).reshape(b.shape)
            samp = nchw.flatten()[:: max(1, nchw.numel() // 64)][:64].numpy()
            ref = g2[f"{tag}_head{k}_sample"]
            assert np.linalg.norm(samp - ref) / (np.linalg.norm(ref) + 1e-9) < 5e-2, (tag, k)
        if not train:
            _, inf_o = ref_ops.decode(hm_o, orc.anchors, 2, mode)
            inf = res[1].cpu()
            assert inf.shape == inf_o.shape
            if mode == "kfiou":
                assert rel(inf, inf_o) < 3e-2
            else:       # csl theta is an argmax over 180 bins: compare everything but the angle column
                keep = [0, 1, 2, 3, 5, 6, 7]
                assert rel(inf[..., keep], inf_o[..., keep]) < 3e-2


def test_training_step_gradients_full_v7_kfiou():
    """One full forward + loss + backward of the flagship configuration (yolov7 kfiou) at 64x64 against the oracle."""
    from ryolov4_amd.lib.loss import ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    nc = 16
    net = Yolo(nc, CFG, "kfiou", "yolov7")
    sd = fill_state(net.state_dict())
    net.load_state_dict(sd)
    net.to(DEV).train()
    orc = ref_model.Yolo(nc, CFG, "kfiou", "yolov7")
    orc.load_state_dict(sd)
    orc.train()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(7))
    tg = synth_targets(2, 6, nc, False, seed=3, img_size=64)
    outs_o = orc(x, True)
    loss_o, items_o = ref_ops.compute_loss(outs_o, tg, orc.anchors, nc, "kfiou", HYP)
    loss_o.backward()
    crit = ComputeKFIoULoss(net, HYP)
    outs = net(x.to(DEV), training=True)
    loss, items = crit(outs, tg.to(DEV))
    loss.backward()
    assert abs(items["total_loss"] - float(items_o["total_loss"])) < 3e-2 * abs(float(items_o["total_loss"]))
    errs = {n: rel(p.grad.cpu(), q.grad) for (n, p), (_, q) in zip(net.named_parameters(), orc.named_parameters())}
    # global direction of the whole gradient (what SGD consumes)
    gp = torch.cat([p.grad.flatten().cpu() for p in net.parameters()]).double()
    go = torch.cat([q.grad.flatten() for q in orc.parameters()]).double()
    cos = float((gp @ go) / (gp.norm() * go.norm()))
    assert cos > 0.995, (cos, sorted(errs.items(), key=lambda kv: -kv[1])[:5])


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
