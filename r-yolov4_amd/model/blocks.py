"""Parameter containers + plan emitters for the reference's building blocks (model/utils.py:6-282).

Each class keeps the reference's attribute names so `state_dict()` keys, `.apply(weights_init_normal)` (train.py:28-33 looks
for nn.Conv2d / nn.BatchNorm2d instances) and checkpoint loading are drop-in; but there is no per-module forward():
`emit(g, x, out=None)` appends the block's kernels to the static plan of engine/graph.py, writing its result into
`out` (a channel slice of a concat buffer) when the caller already knows where the tensor will be concatenated.
"""
import torch
import torch.nn as nn

_ACT_MODULE = {"mish": nn.Mish, "leaky": lambda: nn.LeakyReLU(0.1, inplace=True), "swish": nn.SiLU}


class Conv(nn.Module):
    """model/utils.py:6-32: Conv2d(k, s, pad=(k-1)//2) [+ BatchNorm2d] [+ Mish | LeakyReLU(0.1) | SiLU]."""

    def __init__(self, c1, c2, k, s, activation, bn=True, bias=False):
        super().__init__()
        mods = [nn.Conv2d(c1, c2, k, s, (k - 1) // 2, bias=bias)]
        if bn:
            mods.append(nn.BatchNorm2d(c2))
        if activation in _ACT_MODULE:
            mods.append(_ACT_MODULE[activation]())
        elif activation != "linear":
            raise NotImplementedError("Acativation function not found.")
        self.conv = nn.ModuleList(mods)
        self.act, self.has_bn = activation, bn

    def emit(self, g, x, out=None, residual=None, stem=False, pool_grad=None):
        assert self.has_bn, "head convs are emitted through Graph.head"
        return g.conv_bn_act(self.conv[0], self.conv[1], self.act, x, out=out, residual=residual, stem=stem, pool_grad=pool_grad)

    def forward(self, x):
        raise RuntimeError("ryolov4_amd blocks have no eager forward: run them through Yolo.forward (static HIP plan)")


class Bottleneck(nn.Module):            # model/utils.py:35-46
    def __init__(self, c1, c2, shortcut=True, e=0.5, act=None):
        super().__init__()
        h = int(c2 * e)
        self.cv1 = Conv(c1, h, 1, 1, act)
        self.cv2 = Conv(h, c2, 3, 1, act)
        self.add = shortcut and c1 == c2

    def emit(self, g, x, out=None):
        return self.cv2.emit(g, self.cv1.emit(g, x), out=out, residual=x if self.add else None)


def _siblings(g, a, b, x, out_a=None, out_b=None, fork=False):
    """cv1 + cv2 of a block: both read x with the same kernel — one shared GEMM launch (Graph.conv_bn_act_group) unless switched
    off (RYOLO_MERGE_SIBLINGS=0: two launches; `fork` then puts the first on the second forward stream as before; the caller joins)."""
    if g.rt.merge_siblings:
        return g.conv_bn_act_group([a, b], x, [out_a, out_b])
    if fork:
        with g.side_branch():
            za = a.emit(g, x, out=out_a)
    else:
        za = a.emit(g, x, out=out_a)
    return [za, b.emit(g, x, out=out_b)]


def _chain(n, c, shortcut, act):
    return nn.Sequential(*[Bottleneck(c, c, shortcut, e=1.0, act=act) for _ in range(n)])


def _emit_chain(g, seq, x, out=None):
    mods = list(seq)
    for i, m in enumerate(mods):
        x = m.emit(g, x, out=out if i == len(mods) - 1 else None)
    return x


class CSP(nn.Module):                   # model/utils.py:49-64
    def __init__(self, c1, c2, n=1, shortcut=True, e=0.5):
        super().__init__()
        h = int(c1 * e)
        self.cv1 = Conv(c1, h, 1, 1, "mish")
        self.cv2 = Conv(c1, h, 1, 1, "mish")
        self.cv3 = Conv(h, h, 1, 1, "mish")
        self.cv4 = Conv(2 * h, c2, 1, 1, "mish")
        self.m = _chain(n, h, shortcut, "mish")
        self.h = h

    def emit(self, g, x, out=None):
        cat = g.new(x.N, x.H, x.W, 2 * self.h)
        x1, _ = _siblings(g, self.cv1, self.cv2, x, None, cat.slice(self.h, self.h))
        self.cv3.emit(g, _emit_chain(g, self.m, x1), out=cat.slice(0, self.h))
        return self.cv4.emit(g, cat, out=out)


class C5(nn.Module):                    # model/utils.py:67-80
    def __init__(self, c1, c2, e=0.5):
        super().__init__()
        h = int(c1 * e)
        self.cv1 = Conv(c1, h, 1, 1, "leaky")
        self.cv2 = Conv(h, c1, 3, 1, "leaky")
        self.cv3 = Conv(c1, h, 1, 1, "leaky")
        self.cv4 = Conv(h, c1, 3, 1, "leaky")
        self.cv5 = Conv(c1, c2, 1, 1, "leaky")

    def emit(self, g, x, out=None):
        for m in (self.cv1, self.cv2, self.cv3, self.cv4):
            x = m.emit(g, x)
        return self.cv5.emit(g, x, out=out)


class C3(nn.Module):                    # model/utils.py:83-95
    def __init__(self, c1, c2, n=1, shortcut=True, e=0.5):
        super().__init__()
        h = int(c1 * e)
        self.cv1 = Conv(c1, h, 1, 1, "swish")
        self.cv2 = Conv(c1, h, 1, 1, "swish")
        self.cv3 = Conv(2 * h, c2, 1, 1, "swish")
        self.m = _chain(n, h, shortcut, "swish")
        self.h = h

    def emit(self, g, x, out=None):
        cat = g.new(x.N, x.H, x.W, 2 * self.h)
        x1, _ = _siblings(g, self.cv1, self.cv2, x, None, cat.slice(self.h, self.h))
        _emit_chain(g, self.m, x1, out=cat.slice(0, self.h))
        return self.cv3.emit(g, cat, out=out)


class ELAN1(nn.Module):                 # model/utils.py:98-118
    def __init__(self, c1, c2, e1=0.5, e2=0.5):
        super().__init__()
        h1, h2 = int(c1 * e1), int(c1 * e2)
        self.cv1 = Conv(c1, h1, 1, 1, "swish")
        self.cv2 = Conv(c1, h1, 1, 1, "swish")
        self.cv3 = Conv(h1, h2, 3, 1, "swish")
        self.cv4 = Conv(h1, h2, 3, 1, "swish")
        self.cv5 = Conv(h2, h2, 3, 1, "swish")
        self.cv6 = Conv(h2, h2, 3, 1, "swish")
        self.cv7 = Conv((h1 + h2) * 2, c2, 1, 1, "swish")
        self.h1, self.h2 = h1, h2

    def emit(self, g, x, out=None):
        h1, h2 = self.h1, self.h2
        cat = g.new(x.N, x.H, x.W, 2 * (h1 + h2))
        _, x2 = _siblings(g, self.cv1, self.cv2, x, cat.slice(0, h1), cat.slice(h1, h1), fork=True)
        x3 = self.cv4.emit(g, self.cv3.emit(g, x2), out=cat.slice(2 * h1, h2))
        self.cv6.emit(g, self.cv5.emit(g, x3), out=cat.slice(2 * h1 + h2, h2))
        g.join_side()
        return self.cv7.emit(g, cat, out=out)


class ELAN2(nn.Module):                 # model/utils.py:121-143
    def __init__(self, c1, c2, e1=0.5, e2=0.25):
        super().__init__()
        h1, h2 = int(c1 * e1), int(c1 * e2)
        self.cv1 = Conv(c1, h1, 1, 1, "swish")
        self.cv2 = Conv(c1, h1, 1, 1, "swish")
        self.cv3 = Conv(h1, h2, 3, 1, "swish")
        self.cv4 = Conv(h2, h2, 3, 1, "swish")
        self.cv5 = Conv(h2, h2, 3, 1, "swish")
        self.cv6 = Conv(h2, h2, 3, 1, "swish")
        self.cv7 = Conv(h1 * 2 + h2 * 4, c2, 1, 1, "swish")
        self.h1, self.h2 = h1, h2

    def emit(self, g, x, out=None):
        h1, h2 = self.h1, self.h2
        cat = g.new(x.N, x.H, x.W, 2 * h1 + 4 * h2)
        _, y = _siblings(g, self.cv1, self.cv2, x, cat.slice(0, h1), cat.slice(h1, h1), fork=True)
        for i, m in enumerate((self.cv3, self.cv4, self.cv5, self.cv6)):
            y = m.emit(g, y, out=cat.slice(2 * h1 + i * h2, h2))
        g.join_side()
        return self.cv7.emit(g, cat, out=out)


class MaxConv(nn.Module):               # model/utils.py:146-160
    def __init__(self, c1, e=0.5):
        super().__init__()
        h = int(c1 * e)
        self.m = nn.MaxPool2d(kernel_size=2, stride=2)
        self.cv1 = Conv(c1, h, 1, 1, "swish")
        self.cv2 = Conv(c1, h, 1, 1, "swish")
        self.cv3 = Conv(h, h, 3, 2, "swish")
        self.h = h

    def emit(self, g, x, out=None):
        cat = out if out is not None else g.new(x.N, x.H // 2, x.W // 2, 2 * self.h)
        if g.training and g.rt.fuse_pool_grad and x.H % 2 == 0 and x.W % 2 == 0 and x.C % 8 == 0 and x.W < 32768:
            # cv2 first: backward runs in reverse forward order, so cv2's data gradient comes AFTER cv1's (which produces the pooled
            # tensor's gradient) and adds the MaxPool gradient in its own store — no pass of its own over the full-resolution gradient
            holder = {}
            t2 = self.cv2.emit(g, x, pool_grad=holder)
            with g.side_branch():                           # pool -> 1x1 beside the 3x3 stride 2
                self.cv1.emit(g, g.maxpool(x, 2, 2, grad_into=holder), out=cat.slice(0, self.h))
            self.cv3.emit(g, t2, out=cat.slice(self.h, self.h))
            g.join_side()
            return cat
        with g.side_branch():                               # pool -> 1x1 beside 1x1 -> 3x3 stride 2
            self.cv1.emit(g, g.maxpool(x, 2, 2), out=cat.slice(0, self.h))
        self.cv3.emit(g, self.cv2.emit(g, x), out=cat.slice(self.h, self.h))
        g.join_side()
        return cat


class ImplicitA(nn.Module):             # model/utils.py:163-173
    def __init__(self, channel, mean=0., std=.02):
        super().__init__()
        self.implicit = nn.Parameter(torch.zeros(1, channel, 1, 1))
        nn.init.normal_(self.implicit, mean=mean, std=std)


class ImplicitM(nn.Module):             # model/utils.py:176-186
    def __init__(self, channel, mean=1., std=.02):
        super().__init__()
        self.implicit = nn.Parameter(torch.ones(1, channel, 1, 1))
        nn.init.normal_(self.implicit, mean=mean, std=std)


class RepConv(nn.Module):               # model/utils.py:189-215 (never re-parameterised by the reference)
    def __init__(self, c1, c2, k=3, s=1, p=1):
        super().__init__()
        self.silu = nn.SiLU()
        self.rbr_identity = nn.BatchNorm2d(num_features=c1) if c2 == c1 and s == 1 else None
        self.rbr_dense = nn.Sequential(nn.Conv2d(c1, c2, k, s, p, bias=False), nn.BatchNorm2d(num_features=c2))
        self.rbr_1x1 = nn.Sequential(nn.Conv2d(c1, c2, 1, s, 0, bias=False), nn.BatchNorm2d(num_features=c2))

    def emit(self, g, x):
        return g.repconv(self, x)


class SPP(nn.Module):                   # model/utils.py:218-244; cat order [m13, m9, m5, x]
    def __init__(self, c1, c2):
        super().__init__()
        h = c1 // 2
        self.cv1 = Conv(c1, h, 1, 1, "leaky")
        self.cv2 = Conv(h, c1, 3, 1, "leaky")
        self.cv3 = Conv(c1, h, 1, 1, "leaky")
        self.m1 = nn.MaxPool2d(kernel_size=5, stride=1, padding=5 // 2)
        self.m2 = nn.MaxPool2d(kernel_size=9, stride=1, padding=9 // 2)
        self.m3 = nn.MaxPool2d(kernel_size=13, stride=1, padding=13 // 2)
        self.cv4 = Conv(h * 4, h, 1, 1, "leaky")
        self.cv5 = Conv(h, c1, 3, 1, "leaky")
        self.cv6 = Conv(c1, c2, 1, 1, "leaky")
        self.h = h

    def emit(self, g, x, out=None):
        h = self.h
        cat = g.new(x.N, x.H, x.W, 4 * h)
        t = self.cv3.emit(g, self.cv2.emit(g, self.cv1.emit(g, x)), out=cat.slice(3 * h, h))
        for i, k in enumerate((13, 9, 5)):
            g.maxpool(t, k, 1, out=cat.slice(i * h, h))
        return self.cv6.emit(g, self.cv5.emit(g, self.cv4.emit(g, cat)), out=out)


class SPPF(nn.Module):                  # model/utils.py:247-261; cat order [x, y1, y2, m(y2)]
    def __init__(self, c1, c2, k=5):
        super().__init__()
        h = c1 // 2
        self.cv1 = Conv(c1, h, 1, 1, "swish")
        self.cv2 = Conv(h * 4, c2, 1, 1, "swish")
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)
        self.h, self.k = h, k

    def emit(self, g, x, out=None):
        h = self.h
        cat = g.new(x.N, x.H, x.W, 4 * h)
        t = self.cv1.emit(g, x, out=cat.slice(0, h))
        for i in range(1, 4):
            t = g.maxpool(t, self.k, 1, out=cat.slice(i * h, h))
        return self.cv2.emit(g, cat, out=out)


class SPPCSPC(nn.Module):               # model/utils.py:264-282; cat order [x1, m5, m9, m13]
    def __init__(self, c1, c2, e=0.5, k=(5, 9, 13)):
        super().__init__()
        h = int(2 * c2 * e)
        self.cv1 = Conv(c1, h, 1, 1, "swish")
        self.cv2 = Conv(c1, h, 1, 1, "swish")
        self.cv3 = Conv(h, h, 3, 1, "swish")
        self.cv4 = Conv(h, h, 1, 1, "swish")
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])
        self.cv5 = Conv(4 * h, h, 1, 1, "swish")
        self.cv6 = Conv(h, h, 3, 1, "swish")
        self.cv7 = Conv(2 * h, c2, 1, 1, "swish")
        self.h, self.k = h, k

    def emit(self, g, x, out=None):
        h = self.h
        cat4 = g.new(x.N, x.H, x.W, 4 * h)
        cat2 = g.new(x.N, x.H, x.W, 2 * h)
        t1, _ = _siblings(g, self.cv1, self.cv2, x, None, cat2.slice(h, h))
        x1 = self.cv4.emit(g, self.cv3.emit(g, t1), out=cat4.slice(0, h))
        for i, k in enumerate(self.k):
            g.maxpool(x1, k, 1, out=cat4.slice((i + 1) * h, h))
        self.cv6.emit(g, self.cv5.emit(g, cat4), out=cat2.slice(0, h))
        return self.cv7.emit(g, cat2, out=out)
