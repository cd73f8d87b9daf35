"""GPU: the detection-head tail kernels (ryolo_head_finish_fwd / _bwd, csrc/elementwise.hip) through the C ABI against plain torch.

forward  pre [M][ldp] fp32 (conv + bias) [* ImplicitM] -> out [B, na, gs, gs, attrs]          (model/yololayer.py:25,76 fused)
backward dout -> dpre bf16 [M][ldd] (= dout * mul, bf16-rounded), dbias += column sums of the ROUNDED dpre, dmul += sum dout * pre
Shapes cover both layouts' alignment cases: attrs = 22 (kfiou nc=16), 8 (kfiou nc=2), 201 (csl nc=16: odd row length, 16-cell tiles),
gs odd (625 cells: anchor runs start at 4-byte-only aligned addresses; last tile ragged) and tiny grids.
Forward is a permutation (+ one multiply): bit-exact.  Backward: dpre bit-exact; sums to 1e-5 relative (fp32 partials, double fold)."""
import pytest
import torch

from ryolov4_amd import hip
from ryolov4_amd.engine import structs as S     # noqa: F401  (registers the entry points)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,gs,na,attrs,with_mul", [(2, 25, 18, 22, True), (3, 13, 18, 8, False), (2, 10, 3, 201, True), (1, 2, 18, 22, True),
                                                    (2, 50, 18, 22, False), (1, 100, 3, 187, True)])
def test_head_finish_forward_and_backward(B, gs, na, attrs, with_mul):
    g = torch.Generator().manual_seed(B * 1000 + gs)
    C = na * attrs
    ldp = (C + 31) // 32 * 32
    M = B * gs * gs
    pre = torch.randn(M, ldp, generator=g).to(DEV)
    mul = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV) if with_mul else None
    out = torch.empty(B, na, gs, gs, attrs, device=DEV)
    hip.call("ryolo_head_finish_fwd", pre.data_ptr(), ldp, mul.data_ptr() if with_mul else None, B, gs, na, attrs, out.data_ptr(), hip.stream())
    v = pre[:, :C] * mul if with_mul else pre[:, :C]
    exp = v.view(B, gs, gs, na, attrs).permute(0, 3, 1, 2, 4).contiguous()
    assert torch.equal(out, exp)

    dout = torch.randn(B, na, gs, gs, attrs, generator=g).to(DEV)
    dpre = torch.zeros(M, ldp, dtype=torch.bfloat16, device=DEV)
    nblk = B * ((gs * gs + 127) // 128)
    scratch = torch.zeros((nblk + 64) * 2 * C, device=DEV)
    dbias = torch.full((C,), 0.5, device=DEV)                       # accumulated into, not overwritten
    dmul = torch.full((C,), -0.25, device=DEV) if with_mul else None
    hip.call("ryolo_head_finish_bwd", dout.data_ptr(), pre.data_ptr(), ldp, mul.data_ptr() if with_mul else None, B, gs, na, attrs,
             dpre.data_ptr(), ldp, dbias.data_ptr(), dmul.data_ptr() if with_mul else None, scratch.data_ptr(), hip.stream())
    d2 = dout.permute(0, 2, 3, 1, 4).reshape(M, C)
    e_dpre = (d2 * mul if with_mul else d2).to(torch.bfloat16)
    assert torch.equal(dpre[:, :C], e_dpre)
    assert not dpre[:, C:].float().abs().any()                      # pad columns stay zero
    torch.testing.assert_close(dbias, 0.5 + e_dpre.double().sum(0).float(), rtol=1e-5, atol=1e-4)
    if with_mul:
        torch.testing.assert_close(dmul, -0.25 + (d2.double() * pre[:, :C].double()).sum(0).float(), rtol=1e-5, atol=1e-3)
