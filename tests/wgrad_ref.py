"""float64 reference for the weight-gradient C-ABI tests: dW[co, ci, tap] = sum_p dY[p, co] * X[in(p, tap), ci] as one float64 matrix product per
tap (rocBLAS dgemm on the device).  The tests used torch's fp32 conv2d backward (MIOpen) until r05: its solver — and with it its rounding — is chosen by
a benchmark at first use on a fresh box, and one cold-box run of the suite put a layer 2e-3 away from it once; a float64 product has no such freedom."""
import torch


def wgrad_fp64(x, dy, B, H, W, Cin, Cout, kh, kw, stride, ph, pw):
    """x [B*H*W, >= Cin] bf16, dy [B*OH*OW, >= Cout] bf16 (NHWC rows) -> [Cout, Cin, kh*kw] float64."""
    OH, OW = (H + 2 * ph - kh) // stride + 1, (W + 2 * pw - kw) // stride + 1
    xd = x[:, :Cin].double().view(B, H, W, Cin)
    xp = torch.zeros(B, H + 2 * ph + stride, W + 2 * pw + stride, Cin, dtype=torch.float64, device=x.device)
    xp[:, ph:ph + H, pw:pw + W] = xd
    g = dy[:, :Cout].double().view(B * OH * OW, Cout)
    out = torch.empty(Cout, Cin, kh * kw, dtype=torch.float64, device=x.device)
    for r in range(kh):
        for s in range(kw):
            xs = xp[:, r:r + stride * OH:stride, s:s + stride * OW:stride][:, :OH, :OW].reshape(B * OH * OW, Cin)
            out[:, :, r * kw + s] = g.t() @ xs
    return out
