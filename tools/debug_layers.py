"""Layer-by-layer comparison of the HIP plan against the torch-CPU oracle on the block-level test network."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from tests.test_gpu_model import _oracle_tiny, _product_tiny, rel, NA, ATTRS, _to_5d
from ryolov4_amd.synth import fill_state
from ryolov4_amd.engine.runtime import NetFunction, Runtime
DEV = "cuda:0"
from tests.bf16_emu import emulate_bf16
orc = _oracle_tiny(); sd = fill_state(orc.state_dict()); orc.load_state_dict(sd); emulate_bf16(orc)
prod = _product_tiny(); prod.load_state_dict(sd); prod.to(DEV).train(); orc.train()
x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(3))
cap = {}
def hook(name):
    def f(m, i, o): cap[name] = (i[0].detach(), o.detach())
    return f
for n, m in orc.named_modules():
    if isinstance(m, (nn.Conv2d,)): m.register_forward_hook(hook(n))
    if type(m).__name__ == "Conv": m.register_forward_hook(hook(n + "#block"))
outs_o = orc(x)
rt = Runtime(prod, torch.device(DEV)); g = rt.graph(2, 64, 64, True)
outs_p = NetFunction.apply(x.to(DEV), torch.zeros(1, requires_grad=True), rt, g)
torch.cuda.synchronize()
names = {id(m): n for n, m in prod.named_modules()}
for cid, (y, z, xin) in g.debug.items():
    n = names[cid]
    yo = cap[n][1]; xo = cap[n][0]
    blk = n.rsplit(".conv.", 1)[0] + "#block"
    zo = cap.get(blk, (None, None))[1]
    ex = rel(xin.to_nchw().cpu(), xo) if xin is not None else -1
    ey = rel(y.to_nchw().cpu(), yo)
    ez = rel(z.to_nchw().cpu(), zo) if zo is not None and zo.shape == z.to_nchw().shape else -1
    print(f"{n:40s} x {ex:8.4f}  y {ey:8.4f}  z {ez:8.4f}  shape {tuple(yo.shape)}")
for a, b in zip(outs_p, outs_o):
    print("head", rel(a.cpu(), _to_5d(b)))
gw = [torch.randn(2, NA, 32, 32, ATTRS, generator=torch.Generator().manual_seed(4)), torch.randn(2, NA, 16, 16, ATTRS, generator=torch.Generator().manual_seed(5))]
sum((_to_5d(o) * g_).sum() for o, g_ in zip(outs_o, gw)).backward()
sum((o * g_.to(DEV)).sum() for o, g_ in zip(outs_p, gw)).backward()
for (n, p), (_, q) in zip(prod.named_parameters(), orc.named_parameters()):
    print(f"grad {n:40s} {rel(p.grad.cpu(), q.grad):8.4f}  |g| {float(q.grad.norm()):.3e}")
