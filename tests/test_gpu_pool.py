"""Stride-1 max pools (SPP / SPPF / SPPCSPC windows 5, 9, 13): the separable row + column kernels against the direct k*k kernels
of the same library on inputs FULL of ties (values quantised to a few levels): outputs and argmax indices bit-identical (the
first-maximum rule in (dy, dx) scanning order is what routes the gradient, as in torch.nn.MaxPool2d), input gradients equal up to
the fp32 summation order before the bf16 rounding."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pool(x, k, dz, separable):
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    hip.lib()
    S.check_layouts()
    B, H, W, Cc = x.shape
    M = B * H * W
    z = torch.empty(M, Cc, dtype=torch.bfloat16, device=x.device)
    idx = torch.empty(M, Cc, dtype=torch.uint8, device=x.device)
    dx = torch.zeros(M, Cc, dtype=torch.bfloat16, device=x.device)
    p = S.PoolParams()
    p.x, p.ldx, p.z, p.ldz = x.data_ptr(), Cc, z.data_ptr(), Cc
    p.NB, p.H, p.W, p.C, p.k, p.stride, p.pad, p.OH, p.OW = B, H, W, Cc, k, 1, k // 2, H, W
    p.idx = idx.data_ptr()
    keep = []
    if separable:
        keep = [torch.empty(M, Cc, dtype=torch.bfloat16, device=x.device), torch.empty(M, Cc, dtype=torch.uint8, device=x.device),
                torch.empty(M, Cc, dtype=torch.float32, device=x.device)]
        p.rowmax, p.rowidx, p.growws = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
    hip.call("ryolo_maxpool_fwd", p, hip.stream())
    p.dz, p.lddz, p.dx, p.lddx, p.accum = dz.data_ptr(), Cc, dx.data_ptr(), Cc, 0
    hip.call("ryolo_maxpool_bwd", p, hip.stream())
    torch.cuda.synchronize()
    return z, idx, dx


@pytest.mark.parametrize("k", [5, 9, 13])
@pytest.mark.parametrize("shape", [(2, 25, 25, 64), (3, 13, 19, 40), (1, 7, 5, 8)])
def test_separable_pool_equals_direct(k, shape):
    B, H, W, Cc = shape
    g = torch.Generator().manual_seed(k)
    x = (torch.randint(0, 6, (B, H, W, Cc), generator=g).float() * 0.25 - 0.5).to(torch.bfloat16).cuda()      # 6 levels: ties everywhere
    dz = torch.randn(B * H * W, Cc, generator=g).to(torch.bfloat16).cuda()
    z0, i0, d0 = _pool(x, k, dz, False)
    z1, i1, d1 = _pool(x, k, dz, True)
    assert torch.equal(z0, z1)
    assert torch.equal(i0, i1), "argmax (first maximum in scanning order) differs"
    ref = torch.nn.functional.max_pool2d(x.float().permute(0, 3, 1, 2), k, 1, k // 2).permute(0, 2, 3, 1).reshape(-1, Cc)
    assert torch.equal(z1.float(), ref)
    assert torch.allclose(d0.float(), d1.float(), rtol=2 ** -7, atol=1e-2)
