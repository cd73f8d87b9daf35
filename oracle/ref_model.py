"""ORACLE — test infrastructure only (see oracle/__init__.py).

torch-CPU fp32 restatement of the reference network (model/utils.py, model/backbone.py, model/neck.py,
model/yolo.py).  Same module tree => same state_dict keys as the reference (the key layout is an ABI),
which is how tests/golden/make_golden.py proves it equal to the imported reference: load_state_dict(strict) +
identical outputs on the same input.  Also the CPU baseline that bench.py times beside the HIP path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ref_ops

_ACT = {"mish": nn.Mish, "leaky": lambda: nn.LeakyReLU(0.1), "swish": nn.SiLU}


class Conv(nn.Module):
    """model/utils.py:6-32 — conv(+BN)(+act); `conv` is a ModuleList so keys are conv.0.*, conv.1.*"""

    def __init__(self, c1, c2, k, s, act, bn=True, bias=False):
        super().__init__()
        layers = [nn.Conv2d(c1, c2, k, s, (k - 1) // 2, bias=bias)]
        if bn:
            layers.append(nn.BatchNorm2d(c2))
        if act != "linear":
            if act not in _ACT:
                raise NotImplementedError("Acativation function not found.")
            layers.append(_ACT[act]())
        self.conv = nn.ModuleList(layers)

    def forward(self, x):
        for m in self.conv:
            x = m(x)
        return x


class Bottleneck(nn.Module):          # model/utils.py:35-46
    def __init__(self, c1, c2, shortcut=True, e=0.5, act=None):
        super().__init__()
        h = int(c2 * e)
        self.cv1, self.cv2 = Conv(c1, h, 1, 1, act), Conv(h, c2, 3, 1, act)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


def _chain(n, c, shortcut, act):
    return nn.Sequential(*[Bottleneck(c, c, shortcut, e=1.0, act=act) for _ in range(n)])


class CSP(nn.Module):                 # model/utils.py:49-64
    def __init__(self, c1, c2, n=1, shortcut=True, e=0.5):
        super().__init__()
        h = int(c1 * e)
        self.cv1, self.cv2 = Conv(c1, h, 1, 1, "mish"), Conv(c1, h, 1, 1, "mish")
        self.cv3, self.cv4 = Conv(h, h, 1, 1, "mish"), Conv(2 * h, c2, 1, 1, "mish")
        self.m = _chain(n, h, shortcut, "mish")

    def forward(self, x):
        return self.cv4(torch.cat((self.cv3(self.m(self.cv1(x))), self.cv2(x)), 1))


class C5(nn.Module):                  # model/utils.py:67-80
    def __init__(self, c1, c2, e=0.5):
        super().__init__()
        h = int(c1 * e)
        self.cv1, self.cv2 = Conv(c1, h, 1, 1, "leaky"), Conv(h, c1, 3, 1, "leaky")
        self.cv3, self.cv4 = Conv(c1, h, 1, 1, "leaky"), Conv(h, c1, 3, 1, "leaky")
        self.cv5 = Conv(c1, c2, 1, 1, "leaky")

    def forward(self, x):
        for m in (self.cv1, self.cv2, self.cv3, self.cv4, self.cv5):
            x = m(x)
        return x


class C3(nn.Module):                  # model/utils.py:83-95
    def __init__(self, c1, c2, n=1, shortcut=True, e=0.5):
        super().__init__()
        h = int(c1 * e)
        self.cv1, self.cv2 = Conv(c1, h, 1, 1, "swish"), Conv(c1, h, 1, 1, "swish")
        self.cv3 = Conv(2 * h, c2, 1, 1, "swish")
        self.m = _chain(n, h, shortcut, "swish")

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class ELAN1(nn.Module):               # model/utils.py:98-118
    def __init__(self, c1, c2, e1=0.5, e2=0.5):
        super().__init__()
        h1, h2 = int(c1 * e1), int(c1 * e2)
        self.cv1, self.cv2 = Conv(c1, h1, 1, 1, "swish"), Conv(c1, h1, 1, 1, "swish")
        self.cv3, self.cv4 = Conv(h1, h2, 3, 1, "swish"), Conv(h1, h2, 3, 1, "swish")
        self.cv5, self.cv6 = Conv(h2, h2, 3, 1, "swish"), Conv(h2, h2, 3, 1, "swish")
        self.cv7 = Conv((h1 + h2) * 2, c2, 1, 1, "swish")

    def forward(self, x):
        a, b = self.cv1(x), self.cv2(x)
        c = self.cv4(self.cv3(b))
        d = self.cv6(self.cv5(c))
        return self.cv7(torch.cat((a, b, c, d), 1))


class ELAN2(nn.Module):               # model/utils.py:121-143
    def __init__(self, c1, c2, e1=0.5, e2=0.25):
        super().__init__()
        h1, h2 = int(c1 * e1), int(c1 * e2)
        self.cv1, self.cv2 = Conv(c1, h1, 1, 1, "swish"), Conv(c1, h1, 1, 1, "swish")
        self.cv3 = Conv(h1, h2, 3, 1, "swish")
        self.cv4, self.cv5, self.cv6 = (Conv(h2, h2, 3, 1, "swish") for _ in range(3))
        self.cv7 = Conv(h1 * 2 + h2 * 4, c2, 1, 1, "swish")

    def forward(self, x):
        ys = [self.cv1(x), self.cv2(x)]
        for m in (self.cv3, self.cv4, self.cv5, self.cv6):
            ys.append(m(ys[-1]))
        return self.cv7(torch.cat(ys, 1))


class MaxConv(nn.Module):             # model/utils.py:146-160
    def __init__(self, c1, e=0.5):
        super().__init__()
        h = int(c1 * e)
        self.m = nn.MaxPool2d(2, 2)
        self.cv1, self.cv2, self.cv3 = Conv(c1, h, 1, 1, "swish"), Conv(c1, h, 1, 1, "swish"), Conv(h, h, 3, 2, "swish")

    def forward(self, x):
        return torch.cat((self.cv1(self.m(x)), self.cv3(self.cv2(x))), 1)


class ImplicitA(nn.Module):           # model/utils.py:163-173
    def __init__(self, c, mean=0., std=.02):
        super().__init__()
        self.implicit = nn.Parameter(torch.zeros(1, c, 1, 1))
        nn.init.normal_(self.implicit, mean=mean, std=std)

    def forward(self, x):
        return self.implicit + x


class ImplicitM(nn.Module):           # model/utils.py:176-186
    def __init__(self, c, mean=1., std=.02):
        super().__init__()
        self.implicit = nn.Parameter(torch.ones(1, c, 1, 1))
        nn.init.normal_(self.implicit, mean=mean, std=std)

    def forward(self, x):
        return self.implicit * x


class RepConv(nn.Module):             # model/utils.py:189-215 (never re-parameterised in the reference)
    def __init__(self, c1, c2, k=3, s=1, p=1):
        super().__init__()
        self.silu = nn.SiLU()
        self.rbr_identity = nn.BatchNorm2d(c1) if c2 == c1 and s == 1 else None
        self.rbr_dense = nn.Sequential(nn.Conv2d(c1, c2, k, s, p, bias=False), nn.BatchNorm2d(c2))
        self.rbr_1x1 = nn.Sequential(nn.Conv2d(c1, c2, 1, s, 0, bias=False), nn.BatchNorm2d(c2))

    def forward(self, x):
        y = self.rbr_dense(x) + self.rbr_1x1(x)
        if self.rbr_identity is not None:
            y = y + self.rbr_identity(x)
        return self.silu(y)


def _pools(ks):
    return [nn.MaxPool2d(k, 1, k // 2) for k in ks]


class SPP(nn.Module):                 # model/utils.py:218-244, cat order [m13, m9, m5, x]
    def __init__(self, c1, c2):
        super().__init__()
        h = c1 // 2
        self.cv1, self.cv2, self.cv3 = Conv(c1, h, 1, 1, "leaky"), Conv(h, c1, 3, 1, "leaky"), Conv(c1, h, 1, 1, "leaky")
        self.m1, self.m2, self.m3 = _pools((5, 9, 13))
        self.cv4, self.cv5, self.cv6 = Conv(h * 4, h, 1, 1, "leaky"), Conv(h, c1, 3, 1, "leaky"), Conv(c1, c2, 1, 1, "leaky")

    def forward(self, x):
        x = self.cv3(self.cv2(self.cv1(x)))
        x = torch.cat((self.m3(x), self.m2(x), self.m1(x), x), 1)
        return self.cv6(self.cv5(self.cv4(x)))


class SPPF(nn.Module):                # model/utils.py:247-261, cat order [x, y1, y2, m(y2)]
    def __init__(self, c1, c2, k=5):
        super().__init__()
        h = c1 // 2
        self.cv1, self.cv2 = Conv(c1, h, 1, 1, "swish"), Conv(h * 4, c2, 1, 1, "swish")
        self.m = nn.MaxPool2d(k, 1, k // 2)

    def forward(self, x):
        x = self.cv1(x)
        y1 = self.m(x)
        y2 = self.m(y1)
        return self.cv2(torch.cat((x, y1, y2, self.m(y2)), 1))


class SPPCSPC(nn.Module):             # model/utils.py:264-282, cat order [x1, m5, m9, m13]
    def __init__(self, c1, c2, e=0.5, k=(5, 9, 13)):
        super().__init__()
        h = int(2 * c2 * e)
        self.cv1, self.cv2 = Conv(c1, h, 1, 1, "swish"), Conv(c1, h, 1, 1, "swish")
        self.cv3, self.cv4 = Conv(h, h, 3, 1, "swish"), Conv(h, h, 1, 1, "swish")
        self.m = nn.ModuleList(_pools(k))
        self.cv5, self.cv6, self.cv7 = Conv(4 * h, h, 1, 1, "swish"), Conv(h, h, 3, 1, "swish"), Conv(2 * h, c2, 1, 1, "swish")

    def forward(self, x):
        x1 = self.cv4(self.cv3(self.cv1(x)))
        y1 = self.cv6(self.cv5(torch.cat([x1] + [m(x1) for m in self.m], 1)))
        return self.cv7(torch.cat((y1, self.cv2(x)), 1))


# ---------------------------------------------------------------------------------------------- backbones
class Backbonev4(nn.Module):          # model/backbone.py:4-36
    def __init__(self):
        super().__init__()
        self.cbm0 = Conv(3, 32, 3, 1, "mish")
        for i, (c1, c2, n) in enumerate(((32, 64, 1), (64, 128, 2), (128, 256, 8), (256, 512, 8), (512, 1024, 4)), 1):
            setattr(self, f"cbm{i}", Conv(c1, c2, 3, 2, "mish"))
            setattr(self, f"csp{i}", CSP(c2, c2, n))
        self.spp = SPP(1024, 512)

    def forward(self, x):
        x = self.cbm0(x)
        feats = []
        for i in range(1, 6):
            x = getattr(self, f"csp{i}")(getattr(self, f"cbm{i}")(x))
            feats.append(x)
        return feats[2], feats[3], self.spp(feats[4])


class Backbonev5(nn.Module):          # model/backbone.py:39-66
    def __init__(self):
        super().__init__()
        self.cbs0 = Conv(3, 64, 6, 2, "swish")
        for i, (c1, c2, n) in enumerate(((64, 128, 3), (128, 256, 6), (256, 512, 9), (512, 1024, 3)), 1):
            setattr(self, f"cbs{i}", Conv(c1, c2, 3, 2, "swish"))
            setattr(self, f"csp{i}", C3(c2, c2, n))
        self.spp = SPPF(1024, 1024)

    def forward(self, x):
        x = self.cbs0(x)
        feats = []
        for i in range(1, 5):
            x = getattr(self, f"csp{i}")(getattr(self, f"cbs{i}")(x))
            feats.append(x)
        return feats[1], feats[2], self.spp(feats[3])


class Backbonev7(nn.Module):          # model/backbone.py:69-101
    def __init__(self):
        super().__init__()
        self.cbs0, self.cbs1 = Conv(3, 32, 3, 1, "swish"), Conv(32, 64, 3, 2, "swish")
        self.cbs2, self.cbs3 = Conv(64, 64, 3, 1, "swish"), Conv(64, 128, 3, 2, "swish")
        self.elan1 = ELAN1(128, 256)
        self.mc1, self.elan2 = MaxConv(256), ELAN1(256, 512)
        self.mc2, self.elan3 = MaxConv(512), ELAN1(512, 1024)
        self.mc3, self.elan4 = MaxConv(1024), ELAN1(1024, 1024, e1=0.25, e2=0.25)
        self.spp = SPPCSPC(1024, 512)

    def forward(self, x):
        x = self.elan1(self.cbs3(self.cbs2(self.cbs1(self.cbs0(x)))))
        d3 = self.elan2(self.mc1(x))
        d4 = self.elan3(self.mc2(d3))
        d5 = self.elan4(self.mc3(d4))
        return d3, d4, self.spp(d5)


# ---------------------------------------------------------------------------------------------- necks
def _head(c, out_ch):
    return Conv(c, out_ch, 1, 1, "linear", bn=False, bias=True)


class Neckv4(nn.Module):              # model/neck.py:4-81
    def __init__(self, out_ch):
        super().__init__()
        self.conv7, self.up1 = Conv(512, 256, 1, 1, "leaky"), nn.Upsample(scale_factor=2)
        self.conv8, self.conv9 = Conv(512, 256, 1, 1, "leaky"), C5(512, 256)
        self.conv14, self.up2 = Conv(256, 128, 1, 1, "leaky"), nn.Upsample(scale_factor=2)
        self.conv15, self.conv16 = Conv(256, 128, 1, 1, "leaky"), C5(256, 128)
        self.conv21, self.conv22 = Conv(128, 256, 3, 1, "leaky"), _head(256, out_ch)
        self.conv23, self.conv24 = Conv(128, 256, 3, 2, "leaky"), C5(512, 256)
        self.conv29, self.conv30 = Conv(256, 512, 3, 1, "leaky"), _head(512, out_ch)
        self.conv31, self.conv32 = Conv(256, 512, 3, 2, "leaky"), C5(1024, 512)
        self.conv37, self.conv38 = Conv(512, 1024, 3, 1, "leaky"), _head(1024, out_ch)

    def forward(self, d5, d4, d3):
        p4 = self.conv9(torch.cat((self.conv8(d4), self.up1(self.conv7(d5))), 1))
        p3 = self.conv16(torch.cat((self.conv15(d3), self.up2(self.conv14(p4))), 1))
        small = self.conv22(self.conv21(p3))
        p4 = self.conv24(torch.cat((self.conv23(p3), p4), 1))
        mid = self.conv30(self.conv29(p4))
        p5 = self.conv32(torch.cat((self.conv31(p4), d5), 1))
        return small, mid, self.conv38(self.conv37(p5))


class Neckv5(nn.Module):              # model/neck.py:84-147
    def __init__(self, out_ch):
        super().__init__()
        self.conv7, self.up1 = Conv(1024, 512, 1, 1, "swish"), nn.Upsample(scale_factor=2, mode="nearest")
        self.csp1 = C3(1024, 512, 3, shortcut=False)
        self.conv14, self.up2 = Conv(512, 256, 1, 1, "swish"), nn.Upsample(scale_factor=2, mode="nearest")
        self.csp2, self.conv15 = C3(512, 256, 3, shortcut=False), _head(256, out_ch)
        self.conv16, self.csp3, self.conv17 = Conv(256, 256, 3, 2, "swish"), C3(512, 512, 3, shortcut=False), _head(512, out_ch)
        self.conv18, self.csp4, self.conv19 = Conv(512, 512, 3, 2, "swish"), C3(1024, 1024, 3, shortcut=False), _head(1024, out_ch)

    def forward(self, d5, d4, d3):
        t5 = self.conv7(d5)
        t4 = self.conv14(self.csp1(torch.cat((d4, self.up1(t5)), 1)))
        p3 = self.csp2(torch.cat((d3, self.up2(t4)), 1))
        p4 = self.csp3(torch.cat((t4, self.conv16(p3)), 1))
        p5 = self.csp4(torch.cat((t5, self.conv18(p4)), 1))
        return self.conv15(p3), self.conv17(p4), self.conv19(p5)


class Neckv7(nn.Module):              # model/neck.py:150-217
    def __init__(self, out_ch):
        super().__init__()
        self.conv1, self.up1, self.elan1 = Conv(512, 256, 1, 1, "swish"), nn.Upsample(scale_factor=2, mode="nearest"), ELAN2(512, 256)
        self.conv2, self.up2, self.elan2 = Conv(256, 128, 1, 1, "swish"), nn.Upsample(scale_factor=2, mode="nearest"), ELAN2(256, 128)
        self.conv3, self.conv4 = Conv(1024, 256, 1, 1, "swish"), Conv(512, 128, 1, 1, "swish")
        self.mc1, self.elan3 = MaxConv(128, e=1.0), ELAN2(512, 256)
        self.mc2, self.elan4 = MaxConv(256, e=1.0), ELAN2(1024, 512)
        for i, c in ((1, 128), (2, 256), (3, 512)):
            setattr(self, f"repVgg{i}", RepConv(c, 2 * c))
            setattr(self, f"ia{i}", ImplicitA(2 * c))
            setattr(self, f"conv{4 + i}", _head(2 * c, out_ch))
            setattr(self, f"im{i}", ImplicitM(out_ch))

    def _det(self, i, x):
        x = getattr(self, f"ia{i}")(getattr(self, f"repVgg{i}")(x))
        return getattr(self, f"im{i}")(getattr(self, f"conv{4 + i}")(x))

    def forward(self, d5, d4, d3):
        p4 = self.elan1(torch.cat((self.conv3(d4), self.up1(self.conv1(d5))), 1))
        p3 = self.elan2(torch.cat((self.conv4(d3), self.up2(self.conv2(p4))), 1))
        q4 = self.elan3(torch.cat((p4, self.mc1(p3)), 1))
        q5 = self.elan4(torch.cat((d5, self.mc2(q4)), 1))
        return self._det(1, p3), self._det(2, q4), self._det(3, q5)


_VER = {"yolov4": (Backbonev4, Neckv4), "yolov5": (Backbonev5, Neckv5), "yolov7": (Backbonev7, Neckv7)}


class Yolo(nn.Module):
    """model/yolo.py:9-51 (the YoloLayer has no parameters, so decode lives in ref_ops.decode)."""

    def __init__(self, n_classes, model_config, mode, ver):
        super().__init__()
        if mode == "csl":
            out_ch = (4 + 180 + 1 + n_classes) * 3
        elif mode == "kfiou":
            out_ch = (5 + 1 + n_classes) * 3 * 6
        else:
            raise NotImplementedError("Loss mode : {} not found.".format(mode))
        self.mode, self.nc = mode, n_classes
        self.anchors = ref_ops.make_anchors(model_config, mode)
        self.backbone, self.neck = _VER[ver][0](), _VER[ver][1](out_ch)

    def head_maps(self, x):
        d3, d4, d5 = self.backbone(x)
        return list(self.neck(d5, d4, d3))

    def forward(self, x, training):
        outs, infer = ref_ops.decode(self.head_maps(x), self.anchors, self.nc, self.mode)
        return outs if training else (outs, infer)
