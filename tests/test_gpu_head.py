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


@pytest.mark.parametrize("with_preobj", [True, False])
@pytest.mark.parametrize("B,gs,na,attrs,och,with_mul", [(2, 25, 18, 22, 5, True), (3, 13, 18, 8, 5, False), (2, 10, 3, 201, 4, True),
                                                        (1, 100, 3, 187, 4, True), (2, 50, 18, 22, 5, False), (2, 13, 18, 7, 5, True),
                                                        (1, 13, 36, 22, 5, True), (2, 3, 18, 22, 5, True)])
def test_head_finish_backward_sparse_form_is_bit_identical(B, gs, na, attrs, och, with_mul, with_preobj):
    """ryolo_head_finish_bwd_sparse reads the compact objectness gradients + the dense rows of matched cells only; on a gradient map of that
    shape (what the fused loss produces) every output must equal the dense entry point's bit for bit.  The unmatched rows of the map the
    sparse call is given are POISONED (NaN) outside the objectness element's compact copy: reading any of them would show.  With preobj (the
    compact objectness column of pre, from ryolo_head_finish_fwd_obj — checked here against a slice of pre) the dedicated kernel runs
    (head_bwd_sparse_kernel; attrs = 7: two objectness columns in one 8-channel chunk); without it, or for na * attrs > 512, the dense kernel's
    sparse tile fill."""
    g = torch.Generator().manual_seed(B * 977 + gs)
    C = na * attrs
    ldp = (C + 31) // 32 * 32
    M = B * gs * gs
    pre = torch.randn(M, ldp, generator=g).to(DEV)
    mul = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV) if with_mul else None
    cells = B * na * gs * gs
    owner = torch.full((cells,), -1, dtype=torch.int32)
    hit = torch.randperm(cells, generator=g)[:max(3, cells // 97)]
    owner[hit] = torch.arange(hit.numel(), dtype=torch.int32)       # (any value >= 0 marks a matched cell; 0 included)
    objgrad = torch.randn(cells, generator=g)
    dense = torch.zeros(cells, attrs)
    dense[:, och] = objgrad
    dense[hit] = torch.randn(hit.numel(), attrs, generator=g)
    objgrad[hit] = float("nan")                                      # matched cells: the dense row is the source, the compact entry is not used...
    poisoned = torch.full((cells, attrs), float("nan"))
    poisoned[hit] = dense[hit]
    # ...except that the kernel copies objgrad first and overwrites every OTHER element from the dense row: keep the objectness element right
    objgrad[hit] = dense[hit, och]
    dense, poisoned, objgrad, owner = dense.to(DEV), poisoned.to(DEV), objgrad.to(DEV), owner.to(DEV)
    nblk = B * ((gs * gs + 127) // 128)
    preobj = None
    if with_preobj:
        out = torch.empty(B, na, gs, gs, attrs, device=DEV)
        preobj = torch.empty(B, na, gs, gs, device=DEV)
        xobj = torch.empty(B, na, gs, gs, device=DEV)
        hip.call("ryolo_head_finish_fwd_obj", pre.data_ptr(), ldp, mul.data_ptr() if with_mul else None, B, gs, na, attrs, out.data_ptr(), och,
                 preobj.data_ptr(), xobj.data_ptr(), hip.stream())
        v = pre[:, :C] * mul if with_mul else pre[:, :C]
        assert torch.equal(out, v.view(B, gs, gs, na, attrs).permute(0, 3, 1, 2, 4).contiguous())
        assert torch.equal(xobj, out[..., och])
        assert torch.equal(preobj, pre[:, :C].view(B, gs, gs, na, attrs)[..., och].permute(0, 3, 1, 2).contiguous())
    res = []
    for sparse in (False, True):
        dpre = torch.zeros(M, ldp, dtype=torch.bfloat16, device=DEV)
        scratch = torch.zeros((nblk + 64) * 2 * C, device=DEV)
        dbias = torch.full((C,), 0.5, device=DEV)
        dmul = torch.full((C,), -0.25, device=DEV) if with_mul else None
        tail = (pre.data_ptr(), ldp, mul.data_ptr() if with_mul else None, B, gs, na, attrs, dpre.data_ptr(), ldp, dbias.data_ptr(),
                dmul.data_ptr() if with_mul else None, scratch.data_ptr(), hip.stream())
        if sparse:
            hip.call("ryolo_head_finish_bwd_sparse", poisoned.data_ptr(), objgrad.data_ptr(), owner.data_ptr(), och,
                     preobj.data_ptr() if with_preobj else None, *tail)
        else:
            hip.call("ryolo_head_finish_bwd", dense.data_ptr(), *tail)
        res.append((dpre, dbias, dmul))
    assert torch.isfinite(res[1][0].float()).all()
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    if with_mul:
        assert torch.equal(res[0][2], res[1][2])


def test_head_finish_backward_sparse_rejects_bad_arguments():
    t = torch.zeros(64, device=DEV)
    with pytest.raises(RuntimeError):
        hip.call("ryolo_head_finish_bwd_sparse", t.data_ptr(), None, None, 5, None, t.data_ptr(), 32, None, 1, 1, 1, 8, t.data_ptr(), 32, t.data_ptr(), None,
                 t.data_ptr(), hip.stream())
    with pytest.raises(RuntimeError):
        hip.call("ryolo_head_finish_fwd_obj", t.data_ptr(), 32, None, 1, 1, 1, 8, t.data_ptr(), 8, t.data_ptr(), None, hip.stream())   # och outside the row
