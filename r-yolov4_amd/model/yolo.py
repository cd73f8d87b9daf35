"""Drop-in for the reference's model/yolo.py (Yolo, :9-72) with backbones (model/backbone.py:4-101) and necks
(model/neck.py:4-217) planned onto the MI355X static-graph engine.

Same constructor, attributes (.anchors, .nc, .backbone, .neck, .yolo), state_dict key layout and forward contract:
    Yolo(n_classes, model_config, mode, ver).forward(i[B,3,S,S] fp32, training)
        training=True  -> list of 3 [B, na, gs, gs, attrs] fp32 head maps
        training=False -> (that list, infer_out [B, sum(na*gs*gs), nc+6])
BatchNorm uses batch statistics iff the module is in .train() mode, exactly like nn.BatchNorm2d in the reference.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import hip
from .blocks import (C3, C5, CSP, ELAN1, ELAN2, SPP, SPPCSPC, SPPF, Conv, ImplicitA, ImplicitM, MaxConv, RepConv)
from .yololayer import YoloCSLLayer, YoloKFIoULayer


# ------------------------------------------------------------------------------------------------ backbones
class Backbonev4(nn.Module):            # model/backbone.py:4-36
    def __init__(self):
        super().__init__()
        self.cbm0 = Conv(3, 32, 3, 1, "mish")
        for i, (c1, c2, n) in enumerate(((32, 64, 1), (64, 128, 2), (128, 256, 8), (256, 512, 8), (512, 1024, 4)), 1):
            setattr(self, f"cbm{i}", Conv(c1, c2, 3, 2, "mish"))
            setattr(self, f"csp{i}", CSP(c2, c2, n))
        self.spp = SPP(1024, 512)

    def emit(self, g, d3_out=None, d4_out=None, d5_out=None):
        x = self.cbm0.emit(g, None, stem=True)
        feats = []
        for i in range(1, 6):
            x = getattr(self, f"csp{i}").emit(g, getattr(self, f"cbm{i}").emit(g, x), out={3: d3_out, 4: d4_out}.get(i))
            feats.append(x)
        return feats[2], feats[3], self.spp.emit(g, feats[4], out=d5_out)


class Backbonev5(nn.Module):            # model/backbone.py:39-66
    def __init__(self):
        super().__init__()
        self.cbs0 = Conv(3, 64, 6, 2, "swish")
        for i, (c1, c2, n) in enumerate(((64, 128, 3), (128, 256, 6), (256, 512, 9), (512, 1024, 3)), 1):
            setattr(self, f"cbs{i}", Conv(c1, c2, 3, 2, "swish"))
            setattr(self, f"csp{i}", C3(c2, c2, n))
        self.spp = SPPF(1024, 1024)

    def emit(self, g, d3_out=None, d4_out=None, d5_out=None):
        x = self.cbs0.emit(g, None, stem=True)
        feats = []
        for i in range(1, 5):
            x = getattr(self, f"csp{i}").emit(g, getattr(self, f"cbs{i}").emit(g, x), out={2: d3_out, 3: d4_out}.get(i))
            feats.append(x)
        return feats[1], feats[2], self.spp.emit(g, feats[3], out=d5_out)


class Backbonev7(nn.Module):            # model/backbone.py:69-101
    def __init__(self):
        super().__init__()
        self.cbs0 = Conv(3, 32, 3, 1, "swish")
        self.cbs1 = Conv(32, 64, 3, 2, "swish")
        self.cbs2 = Conv(64, 64, 3, 1, "swish")
        self.cbs3 = Conv(64, 128, 3, 2, "swish")
        self.elan1 = ELAN1(128, 256)
        self.mc1 = MaxConv(256)
        self.elan2 = ELAN1(256, 512)
        self.mc2 = MaxConv(512)
        self.elan3 = ELAN1(512, 1024)
        self.mc3 = MaxConv(1024)
        self.elan4 = ELAN1(1024, 1024, e1=0.25, e2=0.25)
        self.spp = SPPCSPC(1024, 512)

    def emit(self, g, d3_out=None, d4_out=None, d5_out=None):
        x = self.cbs0.emit(g, None, stem=True)
        x = self.cbs3.emit(g, self.cbs2.emit(g, self.cbs1.emit(g, x)))
        x = self.elan1.emit(g, x)
        d3 = self.elan2.emit(g, self.mc1.emit(g, x), out=d3_out)
        d4 = self.elan3.emit(g, self.mc2.emit(g, d3), out=d4_out)
        d5 = self.elan4.emit(g, self.mc3.emit(g, d4))
        return d3, d4, self.spp.emit(g, d5, out=d5_out)


# ------------------------------------------------------------------------------------------------ necks
def _head_conv(c, out_ch):
    return Conv(c, out_ch, 1, 1, "linear", bn=False, bias=True)


class Neckv4(nn.Module):                # model/neck.py:4-81
    D_CH = (256, 512, 512)              # channels of d3, d4, d5

    def __init__(self, output_ch):
        super().__init__()
        self.conv7 = Conv(512, 256, 1, 1, "leaky")
        self.up1 = nn.Upsample(scale_factor=2)
        self.conv8 = Conv(512, 256, 1, 1, "leaky")
        self.conv9 = C5(512, 256)
        self.conv14 = Conv(256, 128, 1, 1, "leaky")
        self.up2 = nn.Upsample(scale_factor=2)
        self.conv15 = Conv(256, 128, 1, 1, "leaky")
        self.conv16 = C5(256, 128)
        self.conv21 = Conv(128, 256, 3, 1, "leaky")
        self.conv22 = _head_conv(256, output_ch)
        self.conv23 = Conv(128, 256, 3, 2, "leaky")
        self.conv24 = C5(512, 256)
        self.conv29 = Conv(256, 512, 3, 1, "leaky")
        self.conv30 = _head_conv(512, output_ch)
        self.conv31 = Conv(256, 512, 3, 2, "leaky")
        self.conv32 = C5(1024, 512)
        self.conv37 = Conv(512, 1024, 3, 1, "leaky")
        self.conv38 = _head_conv(1024, output_ch)

    def plan_inputs(self, g, B, H8):
        """Concat buffers whose slices the backbone must write directly (cat([x2, x1]) at model/neck.py:72)."""
        self.cat5 = g.new(B, H8 // 4, H8 // 4, 1024)
        return None, None, self.cat5.slice(512, 512)

    def emit(self, g, d5, d4, d3, na, attrs):
        B = d5.N
        cat4 = g.new(B, d4.H, d4.W, 512)
        self.conv8.emit(g, d4, out=cat4.slice(0, 256))
        g.upsample(self.conv7.emit(g, d5), out=cat4.slice(256, 256))
        cat4b = g.new(B, d4.H, d4.W, 512)                                   # cat([conv23(p3), p4])  (neck.py:64)
        p4 = self.conv9.emit(g, cat4, out=cat4b.slice(256, 256))
        cat3 = g.new(B, d3.H, d3.W, 256)
        self.conv15.emit(g, d3, out=cat3.slice(0, 128))
        g.upsample(self.conv14.emit(g, p4), out=cat3.slice(128, 128))
        p3 = self.conv16.emit(g, cat3)
        o_small = g.head(self.conv22.conv[0], self.conv21.emit(g, p3), na, attrs)
        self.conv23.emit(g, p3, out=cat4b.slice(0, 256))
        p4 = self.conv24.emit(g, cat4b)
        o_mid = g.head(self.conv30.conv[0], self.conv29.emit(g, p4), na, attrs)
        self.conv31.emit(g, p4, out=self.cat5.slice(0, 512))
        p5 = self.conv32.emit(g, self.cat5)
        o_big = g.head(self.conv38.conv[0], self.conv37.emit(g, p5), na, attrs)
        return o_small, o_mid, o_big


class Neckv5(nn.Module):                # model/neck.py:84-147
    def __init__(self, output_ch):
        super().__init__()
        self.conv7 = Conv(1024, 512, 1, 1, "swish")
        self.up1 = nn.Upsample(scale_factor=2, mode="nearest")
        self.csp1 = C3(1024, 512, 3, shortcut=False)
        self.conv14 = Conv(512, 256, 1, 1, "swish")
        self.up2 = nn.Upsample(scale_factor=2, mode="nearest")
        self.csp2 = C3(512, 256, 3, shortcut=False)
        self.conv15 = _head_conv(256, output_ch)
        self.conv16 = Conv(256, 256, 3, 2, "swish")
        self.csp3 = C3(512, 512, 3, shortcut=False)
        self.conv17 = _head_conv(512, output_ch)
        self.conv18 = Conv(512, 512, 3, 2, "swish")
        self.csp4 = C3(1024, 1024, 3, shortcut=False)
        self.conv19 = _head_conv(1024, output_ch)

    def plan_inputs(self, g, B, H8):
        self.cat4 = g.new(B, H8 // 2, H8 // 2, 1024)        # cat([d4, up1(t5)])  (neck.py:119)
        self.cat3 = g.new(B, H8, H8, 512)                   # cat([d3, up2(t4)])  (neck.py:127)
        return self.cat3.slice(0, 256), self.cat4.slice(0, 512), None

    def emit(self, g, d5, d4, d3, na, attrs):
        B = d5.N
        cat5 = g.new(B, d5.H, d5.W, 1024)                   # cat([t5, conv18(p4)])  (neck.py:141)
        t5 = self.conv7.emit(g, d5, out=cat5.slice(0, 512))
        g.upsample(t5, out=self.cat4.slice(512, 512))
        cat4b = g.new(B, d4.H, d4.W, 512)                   # cat([t4, conv16(p3)])  (neck.py:135)
        t4 = self.conv14.emit(g, self.csp1.emit(g, self.cat4), out=cat4b.slice(0, 256))
        g.upsample(t4, out=self.cat3.slice(256, 256))
        p3 = self.csp2.emit(g, self.cat3)
        o_small = g.head(self.conv15.conv[0], p3, na, attrs)
        self.conv16.emit(g, p3, out=cat4b.slice(256, 256))
        p4 = self.csp3.emit(g, cat4b)
        o_mid = g.head(self.conv17.conv[0], p4, na, attrs)
        self.conv18.emit(g, p4, out=cat5.slice(512, 512))
        p5 = self.csp4.emit(g, cat5)
        o_big = g.head(self.conv19.conv[0], p5, na, attrs)
        return o_small, o_mid, o_big


class Neckv7(nn.Module):                # model/neck.py:150-217
    def __init__(self, output_ch):
        super().__init__()
        self.conv1 = Conv(512, 256, 1, 1, "swish")
        self.up1 = nn.Upsample(scale_factor=2, mode="nearest")
        self.elan1 = ELAN2(512, 256)
        self.conv2 = Conv(256, 128, 1, 1, "swish")
        self.up2 = nn.Upsample(scale_factor=2, mode="nearest")
        self.elan2 = ELAN2(256, 128)
        self.conv3 = Conv(1024, 256, 1, 1, "swish")
        self.conv4 = Conv(512, 128, 1, 1, "swish")
        self.mc1 = MaxConv(128, e=1.0)
        self.elan3 = ELAN2(512, 256)
        self.mc2 = MaxConv(256, e=1.0)
        self.elan4 = ELAN2(1024, 512)
        for i, c in ((1, 128), (2, 256), (3, 512)):
            setattr(self, f"repVgg{i}", RepConv(c, 2 * c))
            setattr(self, f"ia{i}", ImplicitA(2 * c))
            setattr(self, f"conv{4 + i}", _head_conv(2 * c, output_ch))
            setattr(self, f"im{i}", ImplicitM(output_ch))

    def plan_inputs(self, g, B, H8):
        self.cat5 = g.new(B, H8 // 4, H8 // 4, 1024)        # cat([d5, mc2(q4)])  (neck.py:210)
        return None, None, self.cat5.slice(0, 512)

    def _det(self, g, i, x, na, attrs):
        rep = getattr(self, f"repVgg{i}").emit(g, x)
        return g.head(getattr(self, f"conv{4 + i}").conv[0], rep, na, attrs, implicit_a=getattr(self, f"ia{i}").implicit,
                      implicit_m=getattr(self, f"im{i}").implicit)

    def emit(self, g, d5, d4, d3, na, attrs):
        B = d5.N
        cat4 = g.new(B, d4.H, d4.W, 512)                    # cat([conv3(d4), up1(conv1(d5))])  (neck.py:192)
        self.conv3.emit(g, d4, out=cat4.slice(0, 256))
        g.upsample(self.conv1.emit(g, d5), out=cat4.slice(256, 256))
        cat4b = g.new(B, d4.H, d4.W, 512)                   # cat([p4, mc1(p3)])  (neck.py:205)
        p4 = self.elan1.emit(g, cat4, out=cat4b.slice(0, 256))
        cat3 = g.new(B, d3.H, d3.W, 256)                    # cat([conv4(d3), up2(conv2(p4))])  (neck.py:198)
        self.conv4.emit(g, d3, out=cat3.slice(0, 128))
        g.upsample(self.conv2.emit(g, p4), out=cat3.slice(128, 128))
        p3 = self.elan2.emit(g, cat3)
        with g.side_branch(lane=2):                       # the heads are tails: nothing in the network reads them
            o_small = self._det(g, 1, p3, na, attrs)
        self.mc1.emit(g, p3, out=cat4b.slice(256, 256))
        q4 = self.elan3.emit(g, cat4b)
        with g.side_branch(lane=2):
            o_mid = self._det(g, 2, q4, na, attrs)
        self.mc2.emit(g, q4, out=self.cat5.slice(512, 512))
        q5 = self.elan4.emit(g, self.cat5)
        o_big = self._det(g, 3, q5, na, attrs)
        return o_small, o_mid, o_big


# ------------------------------------------------------------------------------------------------ Yolo
class Yolo(nn.Module):
    def __init__(self, n_classes, model_config, mode, ver):
        super().__init__()
        anchors = model_config["anchors"]
        angles = [a * np.pi / 180 for a in model_config["angles"]]
        strides = [8, 16, 32]
        if mode == "csl":
            output_ch = (4 + 180 + 1 + n_classes) * 3
            an = self._make_anchors(strides, anchors)
            layer = YoloCSLLayer(n_classes, an, strides)
        elif mode == "kfiou":
            output_ch = (5 + 1 + n_classes) * 3 * 6
            an = self._make_rotated_anchors(strides, anchors, angles)
            layer = YoloKFIoULayer(n_classes, an, strides)
        else:
            raise NotImplementedError("Loss mode : {} not found.".format(mode))
        self.anchors = an
        self.nc = n_classes
        self.mode, self.ver = mode, ver
        nets = {"yolov4": (Backbonev4, Neckv4), "yolov5": (Backbonev5, Neckv5), "yolov7": (Backbonev7, Neckv7)}
        self.backbone = nets[ver][0]()
        self.neck = nets[ver][1](output_ch)
        self.yolo = layer
        self._rt = None
        self._grad_hook = None                  # set by parallel.DataParallel: all-reduce of the flat gradient buffer
        self._flag = torch.zeros(1, requires_grad=True)
        self.frozen_bn = False                  # True: BatchNorm keeps its running statistics but gradients still flow (fine-tuning)

    # model/yolo.py:54-72
    @staticmethod
    def _make_anchors(strides, anchors):
        return [[[a[i] / s, a[i + 1] / s] for i in range(0, len(a), 2)] for s, a in zip(strides, anchors)]

    @staticmethod
    def _make_rotated_anchors(strides, anchors, angles):
        return [[[a[i] / s, a[i + 1] / s, ang] for i in range(0, len(a), 2) for ang in angles] for s, a in zip(strides, anchors)]

    # ------------------------------------------------------------------ engine plumbing
    def runtime(self, device=None):
        if self._rt is None:
            from ..engine.runtime import Runtime
            dev = device or next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("ryolov4_amd.Yolo runs on a HIP device only (model.to('cuda') first); there is no CPU path")
            self._rt = Runtime(self, dev)
        return self._rt

    def _emit(self, g):
        na = len(self.anchors[0])
        attrs = self.nc + (185 if self.mode == "csl" else 6)
        d3_out, d4_out, d5_out = self.neck.plan_inputs(g, g.B, g.Hin // 8)
        d3, d4, d5 = self.backbone.emit(g, d3_out=d3_out, d4_out=d4_out, d5_out=d5_out)
        self.neck.emit(g, d5, d4, d3, na, attrs)

    def forward(self, i, training):
        hip.require_device(i, "Yolo.forward")
        if i.dim() != 4 or i.size(1) != 3 or i.size(2) % 32 or i.size(3) % 32 or i.size(2) != i.size(3):
            raise RuntimeError("Yolo.forward: expected [B, 3, S, S] with S a multiple of 32")
        rt = self.runtime(i.device)
        g = rt.graph(i.size(0), i.size(2), i.size(3), self.training or self.frozen_bn, frozen=self.frozen_bn)
        from ..engine.runtime import NetFunction
        x = i.float().contiguous()
        if g.training and torch.is_grad_enabled():
            outs = NetFunction.apply(x, self._flag, rt, g)
        else:
            with torch.no_grad():
                outs = NetFunction.forward(_NoCtx(), x, None, rt, g)
        return self.yolo(list(outs), training)


    def capture_inference(self, batch, size, device=None, static_weights=True, post=None):
        """hipGraph-captured inference (BASELINE config C5): the eval forward tape (≈330 C-ABI launches: folded BN + activation
        GEMMs, pooling, heads) and the YoloLayer decode are captured ONCE into a hipGraph on static buffers; the returned callable
        copies a batch into the static input and replays the graph — one host call instead of hundreds, which is what batch-1
        latency is made of.  Returns fn(imgs[B,3,S,S]) -> (head maps list, detections [B, rows, nc+6]); the outputs are the
        graph's static buffers (valid until the next call).
        post = (conf_thres, iou_thres): post_process (score filter, radix-select top-K, rotated NMS with its on-device greedy reduce) is
        captured in the SAME graph on worst-case sized buffers (lib.general.PostProcessPlan: no allocation, no host read); fn then
        returns (head maps, detections, dets [B, max_det, 7] zero padded, num [B] int32 on the device).  As in the reference, whose
        post_process multiplies predictions[:, :, 6:] by the objectness in place (lib/general.py:155), the `detections` buffer returned
        WITH post holds class scores already multiplied by objectness — with post=None it holds the raw decode.
        static_weights: the fp32 -> bf16 weight repack and the folded-BN coefficient kernels (weights-only work, ≈100 launches) are
        left out of the graph; re-capture after changing the weights."""
        if self.training:
            raise RuntimeError("capture_inference: call .eval() first")
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        static_in = torch.zeros((batch, 3, size, size), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                                   # eager warm-up: plan build, lazy kernel attributes, allocator pools
                self(static_in, False)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        g = self.runtime(dev).graph(batch, size, size, False)
        g.static_weights = static_weights                        # bf16 weight images and folded BN coefficients are already in place
        plan = None
        if post is not None:
            from ..lib.general import PostProcessPlan
            with torch.no_grad():
                _, probe = self(static_in, False)
            plan = PostProcessPlan(batch, probe.shape[1], probe.shape[2] - 6, dev, post[0], post[1])
            plan.run(probe)                                      # warm-up outside the capture (lazy kernel attributes)
            torch.cuda.synchronize(dev)
        dets = num = None
        try:
            with torch.cuda.graph(graph), torch.no_grad():
                heads, infer = self(static_in, False)
                if plan is not None:
                    dets, num = plan.run(infer)
        finally:
            g.static_weights = False

        def run(imgs):
            static_in.copy_(imgs)
            graph.replay()
            return (heads, infer) if plan is None else (heads, infer, dets, num)
        run.graph, run.static_input, run.post_plan = graph, static_in, plan
        return run


class _NoCtx:
    pass
