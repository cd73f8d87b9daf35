"""GPU: the fused fold + finalize launches (csrc/elementwise.hip: bn_fold_finalize_kernel, bn_bwd_fold_finalize_kernel) hand partial sums
from FS x C/32 workgroups to the last-arriving one inside ONE launch (agent-scope stores -> ticket -> acquire -> agent-scope loads).  A
broken hand-off reads stale partials — LAST CALL's values when the buffer is reused, which a fixed-input test cannot see.  So: the SAME
statistics buffer, NEW random contents on every call (scale changing 100x between calls), a copy kernel streaming on a second stream to
make arrival order uneven, every coefficient compared with a float64 reference, every call."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(side, buf_a, buf_b, n):
    with torch.cuda.stream(side):
        for _ in range(n):
            buf_b.copy_(buf_a)


@pytest.mark.parametrize("rows,C,c0,ld", [(300, 64, 0, 64), (1000, 256, 0, 256), (20000, 128, 0, 128), (5000, 96, 160, 256), (160000, 32, 0, 32)])
def test_forward_fold_finalize_every_call_against_fp64(rows, C, c0, ld):
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    S.check_layouts()
    g = torch.Generator().manual_seed(rows + C)
    stats = torch.empty((rows + 64, 2, ld), dtype=torch.float32, device=DEV)          # the caller's layout: rows + 64 scratch rows
    gamma = torch.rand(C, generator=g).to(DEV) + 0.5
    beta = torch.randn(C, generator=g).to(DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    out = torch.empty((4, C), dtype=torch.float32, device=DEV)
    side = torch.cuda.Stream()
    big_a = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    big_b = torch.empty_like(big_a)
    count = float(rows * 128)
    for it in range(24):
        scale = 100.0 if it % 2 else 1.0
        s1 = (torch.randn(rows, ld, generator=g) * scale + 3.0 * scale)
        s2 = (s1.abs() * scale + torch.rand(rows, ld, generator=g) * 50.0 * scale * scale) * 128
        s1 = s1 * 128
        stats[:rows, 0].copy_(s1)
        stats[:rows, 1].copy_(s2)
        torch.cuda.synchronize()
        _load(side, big_a, big_b, 3)
        hip.call("ryolo_bn_finalize_slice", stats.data_ptr(), rows, ld, c0, C, count, 1e-5, 0.1, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
                 rv.data_ptr(), out.data_ptr(), hip.stream())
        torch.cuda.synchronize()
        mean = s1[:, c0:c0 + C].double().sum(0) / count
        var = (s2[:, c0:c0 + C].double().sum(0) / count - mean * mean).clamp(min=0)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        o = out.cpu().double()
        assert torch.allclose(o[0], mean, rtol=1e-6, atol=1e-7), (it, float((o[0] - mean).abs().max()))
        assert torch.allclose(o[1], invstd, rtol=1e-5), (it, float((o[1] / invstd - 1).abs().max()))
        assert torch.allclose(o[2], gamma.cpu().double() * invstd, rtol=1e-5)
        assert torch.allclose(o[3], beta.cpu().double() - mean * gamma.cpu().double() * invstd, rtol=1e-4, atol=1e-4 * float(mean.abs().max() * invstd.max()))


@pytest.mark.parametrize("M,C,act", [(640000, 64, 3), (160000, 256, 1), (2560000, 32, 3)])
def test_backward_bn_sums_every_call_against_a_second_evaluation(M, C, act):
    """ryolo_bn_act_bwd (reduce -> fused fold + finalize -> apply) on new data every call: dgamma / dbeta / the coefficient rows must equal
    what the SAME call produces on a fresh buffer (no reuse, so no stale data possible) — bit for bit (fixed summation order)."""
    from ryolov4_amd import hip
    from ryolov4_amd.engine import structs as S
    S.check_layouts()
    nblk, rpb = S.I(), S.I()
    hip.call("ryolo_bn_act_bwd_blocks", M, C, nblk, rpb)
    assert nblk.value > 256                                                             # the fused path
    g = torch.Generator().manual_seed(M // 1000 + C)
    co = torch.stack((torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1)).to(DEV)
    reused = torch.empty((nblk.value + 64, 2, C), dtype=torch.float32, device=DEV)
    side = torch.cuda.Stream()
    big_a = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    big_b = torch.empty_like(big_a)

    def run(partial, y, dz):
        dy = torch.empty_like(y)
        bco = torch.empty((3, C), dtype=torch.float32, device=DEV)
        dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        q = S.BnActParams()
        q.y1, q.ld1, q.co1 = y.data_ptr(), C, co.data_ptr()
        q.M, q.C, q.act = M, C, act
        q.dz, q.lddz, q.dy1, q.lddy1 = dz.data_ptr(), C, dy.data_ptr(), C
        q.partial = partial.data_ptr()
        hip.call("ryolo_bn_act_bwd", q, dgam.data_ptr(), dbet.data_ptr(), None, None, bco.data_ptr(), 0, hip.stream())
        torch.cuda.synchronize()
        return dgam.cpu(), dbet.cpu(), bco.cpu(), dy
    for it in range(6):
        scale = 30.0 if it % 2 else 1.0
        y = (torch.randn(M, C, generator=g, dtype=torch.float32) * scale).to(DEV).to(torch.bfloat16)
        dz = (torch.randn(M, C, generator=g, dtype=torch.float32) * scale).to(DEV).to(torch.bfloat16)
        _load(side, big_a, big_b, 3)
        a = run(reused, y, dz)
        fresh = torch.zeros((nblk.value + 64, 2, C), dtype=torch.float32, device=DEV)
        b = run(fresh, y, dz)
        for u, v, name in zip(a[:3], b[:3], ("dgamma", "dbeta", "bco")):
            assert torch.equal(u, v), (it, name, float((u - v).abs().max()))
        assert torch.equal(a[3], b[3])
        assert float(a[0].abs().sum()) > 0
