"""tests/wgrad_ref.py (the float64 weight-gradient reference of the GPU C-ABI tests) against torch's float64 autograd of conv2d, on the CPU."""
import torch

from tests.wgrad_ref import wgrad_fp64


def test_fp64_reference_equals_conv2d_autograd():
    g = torch.Generator().manual_seed(0)
    for (B, H, W, Cin, Cout, k, st) in [(2, 9, 7, 8, 5, (3, 3), 1), (2, 9, 7, 8, 5, (3, 3), 2), (2, 8, 8, 8, 5, (1, 1), 2), (1, 6, 10, 4, 3, (1, 3), 1), (3, 11, 5, 8, 8, (1, 1), 1)]:
        kh, kw = k
        ph, pw = (kh - 1) // 2, (kw - 1) // 2
        OH, OW = (H + 2 * ph - kh) // st + 1, (W + 2 * pw - kw) // st + 1
        x = torch.randn(B * H * W, Cin + 8, generator=g).bfloat16()          # (wider rows than Cin / Cout: concat strides)
        dy = torch.randn(B * OH * OW, Cout + 8, generator=g).bfloat16()
        ref = wgrad_fp64(x, dy, B, H, W, Cin, Cout, kh, kw, st, ph, pw)
        w0 = torch.zeros(Cout, Cin, kh, kw, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv2d(x[:, :Cin].double().view(B, H, W, Cin).permute(0, 3, 1, 2), w0, stride=st, padding=(ph, pw)).backward(
            dy[:, :Cout].double().view(B, OH, OW, Cout).permute(0, 3, 1, 2))
        assert torch.equal(ref, w0.grad.reshape(Cout, Cin, kh * kw)) or float((ref - w0.grad.reshape(Cout, Cin, kh * kw)).abs().max()) < 1e-12
