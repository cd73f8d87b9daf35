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


def _head_gemm(x, w, bias, mul, B, gs, na, attrs, och, fused, xobj=None):
    """The detection head's 1x1 GEMM through ryolo_conv_gemm: row-major fp32 [M][ldC] + ryolo_head_finish_fwd (the r01-r04 form), or the
    final layout from the GEMM's own epilogue (ConvGemmParams.head_attrs)."""
    M, Cin, C = B * gs * gs, x.shape[1], na * attrs
    ldp = (C + 31) // 32 * 32
    zeros = torch.zeros(256, dtype=torch.uint8, device=DEV)
    p = S.ConvGemmParams()
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = x.data_ptr(), B, gs, gs, Cin, Cin
    p.W, p.Nout, p.wtaps = w.data_ptr(), C, 1
    p.OH, p.OW, p.sh, p.sw = gs, gs, 1, 1
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, gs, gs
    p.nclasses = 1
    p.cls[0].ntaps = 1
    p.epi, p.bias = S.EPI_F32_BIAS, bias.data_ptr()
    p.zeros, p.pipe = zeros.data_ptr(), 0x301
    out = torch.full((B, na, gs, gs, attrs), float("nan"), device=DEV)
    if fused:
        p.out, p.ldC = out.data_ptr(), attrs
        p.head_attrs, p.head_och = attrs, och
        p.scale = mul.data_ptr() if mul is not None else None
        p.stats = xobj.data_ptr() if xobj is not None else None
        hip.call("ryolo_conv_gemm", p, hip.stream())
    else:
        pre = torch.zeros(M, ldp, device=DEV)
        p.out, p.ldC = pre.data_ptr(), ldp
        hip.call("ryolo_conv_gemm", p, hip.stream())
        hip.call("ryolo_head_finish_fwd", pre.data_ptr(), ldp, mul.data_ptr() if mul is not None else None, B, gs, na, attrs, out.data_ptr(), hip.stream())
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("B,gs,na,attrs,och,Cin,with_mul", [(2, 25, 18, 22, 5, 256, True), (3, 13, 18, 8, 5, 64, False), (2, 10, 3, 201, 4, 128, True),
                                                            (1, 50, 18, 22, 5, 512, True), (2, 7, 18, 7, 5, 32, True), (1, 20, 3, 187, 4, 1024, False)])
def test_head_gemm_writes_the_final_layout_bit_identically(B, gs, na, attrs, och, Cin, with_mul):
    """ConvGemmParams.head_attrs: bias, ImplicitM and the [B, na, gs, gs, attrs] permute in the epilogue of the head's GEMM.  Same accumulators, same
    two fp32 operations per element as the row-major GEMM + ryolo_head_finish_fwd: equal bit for bit; the compact objectness logits equal the
    map's column.  Shapes: ragged last M tile, quads that straddle an anchor boundary (attrs = 22, 7, 201, 187 are not multiples of 4), a last
    N tile that is mostly masked (C = 396, 144, 603, 126, 561)."""
    g = torch.Generator().manual_seed(B * 131 + gs)
    M, C = B * gs * gs, na * attrs
    x = torch.randn(M, Cin, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(C, Cin, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    bias = torch.randn(C, generator=g).to(DEV)
    mul = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV) if with_mul else None
    ref = _head_gemm(x, w, bias, mul, B, gs, na, attrs, och, fused=False)
    xobj = torch.full((B, na, gs, gs), float("nan"), device=DEV)
    got = _head_gemm(x, w, bias, mul, B, gs, na, attrs, och, fused=True, xobj=xobj)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref)
    assert torch.equal(xobj, ref[..., och])
    # and against plain torch (fp32 accumulate of the same bf16 operands)
    t = (x.float() @ w.float().t() + bias)
    if with_mul:
        t = t * mul
    torch.testing.assert_close(got, t.view(B, gs, gs, na, attrs).permute(0, 3, 1, 2, 4).contiguous(), rtol=2e-4, atol=2e-3)
    got2 = _head_gemm(x, w, bias, mul, B, gs, na, attrs, och, fused=True)             # (the compact copy is optional)
    assert torch.equal(got2, ref)


def test_head_gemm_final_layout_rejects_what_it_cannot_run():
    x = torch.zeros(64, 32, dtype=torch.bfloat16, device=DEV)
    p = S.ConvGemmParams()
    p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = x.data_ptr(), 1, 8, 8, 32, 32
    p.W, p.Nout, p.wtaps = x.data_ptr(), 24, 1
    p.OH, p.OW, p.sh, p.sw = 8, 8, 1, 1
    p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, 8, 8
    p.nclasses = 1
    p.cls[0].ntaps = 1
    p.epi, p.out, p.zeros, p.pipe = S.EPI_F32_BIAS, x.data_ptr(), x.data_ptr(), 0x301
    p.head_attrs, p.head_och = 8, 5
    rows, kern = S.I(), S.I()
    hip.call("ryolo_conv_gemm_plan", p, rows, kern)                       # fine
    p.pipe = 0x300                                                        # register-staged mainloop: no 1x1 instantiation
    with pytest.raises(RuntimeError, match="unsupported|UNSUPPORTED|-"):
        hip.call("ryolo_conv_gemm_plan", p, rows, kern)
    p.pipe, p.head_attrs = 0x301, 5                                       # 24 columns are not whole rows of 5; rows shorter than 7
    with pytest.raises(RuntimeError):
        hip.call("ryolo_conv_gemm", p, hip.stream())


@pytest.mark.parametrize("Cout,K,with_bias,with_a", [(396, 256, True, True), (54, 1024, True, False), (24, 32, False, True)])
def test_head_wgrad_finish(Cout, K, with_bias, with_a):
    """Ge = G + s (x) a; dW += m Ge, db += m s, dm += rowdot(W, Ge) + b s, da += W^T (m s); G and s cleared (ryolo_head_wgrad_finish) against
    torch in double; and the folded bias b + W a (ryolo_head_bias_fold)."""
    g = torch.Generator().manual_seed(Cout + K)
    G = torch.randn(Cout, K, generator=g).to(DEV)
    s = torch.randn(Cout, generator=g).to(DEV)
    W = torch.randn(Cout, K, generator=g).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    m = (1 + 0.1 * torch.randn(Cout, generator=g)).to(DEV)
    a = (0.1 * torch.randn(K, generator=g)).to(DEV)
    dW = torch.randn(Cout, K, generator=g).to(DEV)
    db = torch.randn(Cout, generator=g).to(DEV)
    dm = torch.randn(Cout, generator=g).to(DEV)
    da = torch.randn(K, generator=g).to(DEV)
    Ge = G.double() + (s.double()[:, None] * a.double()[None, :] if with_a else 0)
    e_dW = dW.double() + m.double()[:, None] * Ge
    e_db = db.double() + m.double() * s.double()
    e_dm = dm.double() + (W.double() * Ge).sum(1) + (b.double() * s.double() if with_bias else 0)
    e_da = da.double() + W.double().t() @ (m.double() * s.double())
    da0 = da.clone()
    fb = torch.empty(Cout, device=DEV)
    hip.call("ryolo_head_bias_fold", W.data_ptr(), b.data_ptr() if with_bias else None, a.data_ptr(), Cout, K, fb.data_ptr(), hip.stream())
    torch.testing.assert_close(fb.double(), (b.double() if with_bias else 0) + W.double() @ a.double(), rtol=1e-5, atol=1e-5)
    hip.call("ryolo_head_wgrad_finish", G.data_ptr(), s.data_ptr(), W.data_ptr(), b.data_ptr() if with_bias else None, m.data_ptr(),
             a.data_ptr() if with_a else None, Cout, K, dW.data_ptr(), db.data_ptr(), dm.data_ptr(), da.data_ptr() if with_a else None, hip.stream())
    torch.testing.assert_close(dW.double(), e_dW, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(db.double(), e_db, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(dm.double(), e_dm, rtol=1e-5, atol=1e-4)
    if with_a:
        torch.testing.assert_close(da.double(), e_da, rtol=1e-5, atol=1e-4)
    else:
        assert torch.equal(da, da0)
    assert not G.abs().any() and not s.abs().any()
