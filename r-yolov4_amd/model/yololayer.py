"""YoloLayer on MI355X — same role as the reference's model/yololayer.py (YoloCSLLayer :8-56, YoloKFIoULayer :59-105).

forward(out, training): `out` is the list of 3 raw head maps.  Two input layouts are accepted:
  * [B, na*attrs, gs, gs]  (NCHW, what the reference's neck returns) -> permuted by ryolo_head_permute;
  * [B, na, gs, gs, attrs] (what this package's conv stack emits straight from the head-conv epilogue) -> used as is.
Like the reference, the list is updated in place (model/yololayer.py:25) and the return value is
`out` when training else `(out, infer_out[B, sum(na*gs*gs), nc+6])`.
"""
import torch
import torch.nn as nn

from .. import hip


class _YoloLayerBase(nn.Module):
    MODE = None

    def __init__(self, num_classes, anchors, stride):
        super().__init__()
        self.nc = num_classes
        self.anchors = anchors
        self.stride = stride

    def _attrs(self):
        return self.nc + (185 if self.MODE == 0 else 6)

    def forward(self, out, training):
        attrs = self._attrs()
        st = None
        for i in range(3):
            x = out[i]
            hip.require_device(x, "YoloLayer")
            na = len(self.anchors[i])
            if x.dim() == 4:
                bs, gs = x.size(0), x.size(2)
                x = x.float().contiguous()
                y = torch.empty((bs, na, gs, gs, attrs), dtype=torch.float32, device=x.device)
                st = st or hip.stream()
                hip.call("ryolo_head_permute", hip.ptr(x), hip.ptr(y), bs, na, attrs, gs, st)
                out[i] = y
            elif x.dim() != 5 or x.size(1) != na or x.size(4) != attrs:
                raise RuntimeError("YoloLayer: unexpected head map shape {}".format(tuple(x.shape)))
        if training:
            return out
        bs = out[0].size(0)
        rows = [len(self.anchors[i]) * out[i].size(2) * out[i].size(3) for i in range(3)]
        infer = torch.empty((bs, sum(rows), self.nc + 6), dtype=torch.float32, device=out[0].device)
        st = st or hip.stream()
        off = 0
        for i in range(3):
            an = []
            for a in self.anchors[i]:
                an += [float(a[0]), float(a[1]), float(a[2]) if len(a) > 2 else 0.0]
            arr = (hip._F * len(an))(*an)
            head = out[i].contiguous()                         # named: a temporary copy must outlive the launch
            hip.call("ryolo_decode", self.MODE, hip.ptr(head), hip.ptr(infer), bs, len(self.anchors[i]),
                     out[i].size(2), self.nc, float(self.stride[i]), arr, off, sum(rows), st)
            off += rows[i]
        return out, infer


class YoloCSLLayer(_YoloLayerBase):
    MODE = 0


class YoloKFIoULayer(_YoloLayerBase):
    MODE = 1


def make_layer(mode, nc, anchors, strides):
    return (YoloCSLLayer if mode == "csl" else YoloKFIoULayer)(nc, anchors, strides)
