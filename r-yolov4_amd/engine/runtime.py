"""Per-model device runtime: flat fp32 parameter / gradient / momentum buffers, bf16 packed weights, cached execution
plans, and the autograd bridge (ONE autograd node for the whole backbone + neck instead of ~600)."""
import ctypes as C
import os

import torch
import torch.nn as nn

from .. import hip
from . import structs as S
from .graph import Graph


def _compact_head_grad(t):
    from ..lib.loss import compact_head_grad        # (lazy: lib/ imports the engine)
    return compact_head_grad(t)


def _clear_handoff():
    from ..lib.loss import clear_handoff
    clear_handoff()


def _register_head_obj(heads):
    from ..lib.loss import register_head_obj
    for h in heads:
        if h.get("xobj") is not None and not isinstance(h["xobj"], int):
            register_head_obj(h["out"], h["xobj"], h["och"])


def _round_up(x, m):
    return (x + m - 1) // m * m


class Runtime:
    def __init__(self, model, device):
        hip.lib()
        S.check_layouts()
        self.model, self.device = model, device
        self.bn_counters = []
        self._flatten()
        self._packed = {}
        self._pack_table = None
        self._graphs = {}
        self._side_streams = {}
        self.momentum_buf = None
        self.adam_state = None            # [exp_avg, exp_avg_sq, step] of adam_step
        self.zeros = torch.zeros(256, dtype=torch.uint8, device=device)       # source of padded rows for the LDS-DMA GEMM loop
        # bits 0-7: generic mainloop (1 LDS-DMA ring [default], 0 register staged);
        # 0x200: 3x3 stride-1 layers run the halo-patch kernel (csrc/conv3x3.hip)
        # 0x100 (32-channel stages only in the generic tapped GEMM) is part of the default since the end of r04: the 64-channel stages (0x201) were
        # +1-2 % on their launches against the r02 side stream; with the lighter side stream of r04 the smaller LDS footprint wins (+0.5 % step, 3 runs)
        self.gemm_pipe = int(os.environ.get("RYOLO_GEMM_PIPE", str(1 | 0x200 | 0x100)), 0)
        self.fuse_stem_bn = os.environ.get("RYOLO_FUSE_STEM_BN", "1") != "0"      # BN + act backward applied inside the stem wgrad kernel
        # 3x3 stride-1 first layer (yolov4 / yolov7): the raw conv output is never stored — statistics pass + fused BN/activation
        # forward, and ONE backward pass over dz that recomputes it from the image (csrc/stem.hip: stem3x3_bwd_kernel)
        self.stem_recompute = os.environ.get("RYOLO_STEM_RECOMPUTE", "1") != "0"
        # narrow stride-2 data gradients (<= 32 input channels: the second conv of yolov4 / yolov7) as one space-to-depth GEMM
        self.s2d_dgrad = os.environ.get("RYOLO_S2D_DGRAD", "1") != "0"
        self.s2d_dgrad_maxc = int(os.environ.get("RYOLO_S2D_DGRAD_MAXC", "32"))     # widest input of a stride-2 3x3 layer that takes this form
        # MaxConv (model/utils.py:146-160): the MaxPool2d(2, 2) gradient is added inside the store of the sibling 1x1 conv's data
        # gradient (same input tensor) instead of a read-modify-write pass over the full-resolution gradient
        self.fuse_pool_grad = os.environ.get("RYOLO_FUSE_POOL_GRAD", "1") != "0"
        self._s2d = {}                    # id(conv) -> (conv, bf16 image [4 Cin][4][CoutP]) refreshed with the other packed weights
        self.side_event = None            # set by Graph.run around a gradient-bucket hook: event of the weight-gradient stream
        self.fwd_fork = os.environ.get("RYOLO_FWD_FORK", "1") != "0"              # sibling branches of ELAN / MaxConv blocks on two streams
        self.wgrad_stream = os.environ.get("RYOLO_WGRAD_STREAM", "1") != "0"      # weight gradients on a second stream (Graph.run)
        # activations and their gradients of a plan as liveness-placed slots of one arena (engine/arena.py); 0 = one tensor per buffer
        self.buffer_reuse = os.environ.get("RYOLO_BUFFER_REUSE", "1") != "0"
        self.wgrad_lanes = int(os.environ.get("RYOLO_WGRAD_LANES", "1"))          # weight gradients round-robin over this many side streams
        self.wgrad_lag = int(os.environ.get("RYOLO_WGRAD_LAG", "8"))             # weight gradients the side stream may fall behind by
        self.fold_repconv = os.environ.get("RYOLO_FOLD_REPCONV", "1") != "0"      # eval plans: RepConv as one re-parameterised 3x3 GEMM
        # sibling convolutions of a block that read the same input (ELAN / CSP / C3 / SPPCSPC cv1 + cv2) as ONE GEMM with
        # concatenated output channels: the input is read once instead of twice (forward and weight gradient) and its gradient is
        # written once instead of store + read-modify-write (Graph.conv_bn_act_group)
        self.merge_siblings = os.environ.get("RYOLO_MERGE_SIBLINGS", "1") != "0"

    def side_stream(self, lane):
        """The engine's extra HIP streams (lane 1: weight gradients / forked forward branches, lane 2: detection-head tails), created
        once per model and shared by all of its plans."""
        st = self._side_streams.get(lane)
        if st is None:
            # RYOLO_SIDE_PRIO: HIP stream priority of the engine's extra streams (0 = default; positive = lower than the main stream where
            # the runtime offers it): a lower-priority weight-gradient stream fills the tails of the main stream's launches instead of
            # competing with them for compute units (A/B in DESIGN.md)
            prio = int(os.environ.get("RYOLO_SIDE_PRIO", "0"))
            try:
                st = torch.cuda.Stream(device=self.device, priority=prio)
            except Exception:
                st = torch.cuda.Stream(device=self.device)
            self._side_streams[lane] = st
        return st

    # ------------------------------------------------------------------ parameters
    def _flatten(self):
        """Re-home every parameter into one flat fp32 buffer (views keep the nn.Parameter objects and state_dict keys)."""
        params = [p for p in self.model.parameters()]
        offs, total = [], 0
        for p in params:
            if p.dtype != torch.float32:
                raise RuntimeError("ryolov4_amd: parameters must be float32 masters (bf16 copies are made internally)")
            offs.append(total)
            total += _round_up(p.numel(), 64)               # 256-byte aligned slices
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.gflat = torch.zeros(total, dtype=torch.float32, device=self.device)
        self._pslice = {}
        for p, o in zip(params, offs):
            v = self.flat[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            self._pslice[id(p)] = (o, p.numel(), p)
        self.params = params
        self.n_flat = total
        # one flat counter tensor for every BatchNorm's num_batches_tracked (a single add per step)
        bns = [m for m in self.model.modules() if isinstance(m, nn.BatchNorm2d)]
        self.nbt = torch.zeros(len(bns), dtype=torch.int64, device=self.device)
        for i, m in enumerate(bns):
            self.nbt[i] = m.num_batches_tracked.to(self.device)
            m.num_batches_tracked.data = self.nbt[i]

    def check_resident(self):
        for o, n, p in self._pslice.values():
            if p.data_ptr() != self.flat.data_ptr() + 4 * o:
                return False
        return True

    def grad_ptr(self, p):
        o, n, _ = self._pslice[id(p)]
        return self.gflat.data_ptr() + 4 * o

    def grad_view(self, p):
        o, n, _ = self._pslice[id(p)]
        return self.gflat[o:o + n].view(p.shape)

    def prepare_grads(self):
        """Called at the start of every backward: parameters whose .grad is not (a view of) the flat gradient buffer are
        (re)attached; a freshly attached slice starts from zero (== optimizer.zero_grad(set_to_none=True) semantics),
        an attached one keeps accumulating (== the reference's gradient accumulation, train.py:198-202)."""
        fresh = [p for p in self.params if p.grad is None or p.grad.data_ptr() != self.grad_ptr(p)]
        if len(fresh) == len(self.params):
            self.gflat.zero_()
        else:
            for p in fresh:
                self.grad_view(p).zero_()
        for p in fresh:
            if p.grad is not None:
                raise RuntimeError("ryolov4_amd: a parameter's .grad was replaced by a foreign tensor")
            p.grad = self.grad_view(p)

    # ------------------------------------------------------------------ packed bf16 weights
    def packed(self, conv):
        pk = self._packed.get(id(conv))
        if pk is None:
            cout, cin = conv.out_channels, conv.in_channels
            taps = conv.kernel_size[0] * conv.kernel_size[1]
            small = cin % 32 != 0                               # stem: K = taps*cin padded, single tap
            cinp = _round_up(taps * cin, 32) if small else cin
            coutp = _round_up(cout, 32)
            wf = torch.zeros((cout, 1 if small else taps, cinp), dtype=torch.bfloat16, device=self.device)
            wd = None if small else torch.zeros((cin, taps, coutp), dtype=torch.bfloat16, device=self.device)
            pk = dict(conv=conv, wf=wf, wd=wd, Cout=cout, Cin=cin, taps=taps, CinP=cinp, CoutP=coutp)
            self._packed[id(conv)] = pk
            self._pack_table = None
        return pk

    def packed_group(self, convs):
        """One pair of bf16 GEMM images for several convolutions with the same input, kernel and stride: Wf [sum Cout][taps][Cin]
        (each member owns a row range) and Wd [Cin][taps][sum Cout] (each member owns a column range, PackEntry.ldWd).  The members'
        fp32 masters, state_dict keys and .grad stay separate tensors."""
        key = ("grp",) + tuple(id(c) for c in convs)
        grp = self._packed.get(key)
        if grp is None:
            c0 = convs[0]
            cin, taps = c0.in_channels, c0.kernel_size[0] * c0.kernel_size[1]
            assert cin % 32 == 0 and all(c.in_channels == cin and c.kernel_size == c0.kernel_size and c.out_channels % 32 == 0 for c in convs)
            ctot = sum(c.out_channels for c in convs)
            wf = torch.zeros((ctot, taps, cin), dtype=torch.bfloat16, device=self.device)
            wd = torch.zeros((cin, taps, ctot), dtype=torch.bfloat16, device=self.device)
            members, off = [], 0
            for c in convs:
                co = c.out_channels
                members.append(dict(conv=c, wf=wf[off:off + co], wd=wd[:, :, off:off + co], Cout=co, Cin=cin, taps=taps, CinP=cin, CoutP=co, ldWd=ctot))
                off += co
            grp = dict(group=True, wf=wf, wd=wd, Ctot=ctot, members=members)
            self._packed[key] = grp
            self._pack_table = None
        return grp

    def _pack_entries(self):
        ents = []
        for pk in self._packed.values():
            ents.extend(pk["members"] if pk.get("group") else [pk])
        return ents

    def pack(self):
        """fp32 masters -> bf16 GEMM images ([Cout][tap][Cin] for fwd/wgrad, [Cin][tap][Cout] for dgrad); one launch."""
        if self._pack_table is None:
            ents = self._pack_entries()
            arr = (S.PackEntry * len(ents))()
            start = 0
            for e, pk in zip(arr, ents):
                e.src, e.wf = pk["conv"].weight.data_ptr(), pk["wf"].data_ptr()
                e.wd = pk["wd"].data_ptr() if pk["wd"] is not None else None
                e.Cout, e.Cin, e.taps, e.CinP, e.CoutP, e.start = pk["Cout"], pk["Cin"], pk["taps"], pk["CinP"], pk["CoutP"], start
                e.ldWd = pk.get("ldWd", 0)
                e.wd_scale = pk["wd_scale"].data_ptr() if pk.get("wd_scale") is not None else None
                if pk["CinP"] != pk["Cin"]:
                    start += (pk["Cout"] * pk["CinP"] + 255) // 256          # stem: 256-element tiles
                else:
                    start += ((pk["Cout"] + 31) // 32) * (pk["Cin"] // 32)       # 32x32 (co, cin) tiles
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self._pack_table = (host.to(self.device), len(ents), start)
        tab, n, total = self._pack_table
        hip.call("ryolo_pack_weights", tab.data_ptr(), n, total, hip.stream())
        for conv, img in self._s2d.values():
            hip.call("ryolo_pack_s2d", conv.weight.data_ptr(), conv.out_channels, conv.in_channels, img.data_ptr(), hip.stream())

    def packed_s2d(self, conv):
        ent = self._s2d.get(id(conv))
        if ent is None:
            coutp = _round_up(conv.out_channels, 32)
            ent = (conv, torch.zeros((4 * conv.in_channels, 4, coutp), dtype=torch.bfloat16, device=self.device))
            self._s2d[id(conv)] = ent
        return ent[1]

    # ------------------------------------------------------------------ plans
    def graph(self, B, H, W, training, frozen=False):
        key = (B, H, W, bool(training), bool(frozen))
        g = self._graphs.get(key)
        if g is None:
            if not self.check_resident():
                raise RuntimeError("ryolov4_amd: parameters were moved after the first forward; build a new Yolo/runtime")
            layout = None
            if self.buffer_reuse:
                # liveness pass on virtual addresses, then the real plan on one arena (engine/arena.py)
                from . import arena
                dry = Graph(self, B, H, W, training, frozen, dry=True)
                dry.begin()
                self.model._emit(dry)
                dry.finish()
                layout = arena.plan(dry, self.wgrad_lag if self.wgrad_stream else 0)
                del dry
            g = Graph(self, B, H, W, training, frozen, layout=layout)
            g.begin()
            self.model._emit(g)
            g.finish()
            if layout is not None and layout.names != ([e[2] for e in g.fwd], [e[2] for e in g.bwd]):
                raise RuntimeError("engine: the two planning passes emitted different tapes")
            self._graphs[key] = g
        return g

    # ------------------------------------------------------------------ fused optimizer (bench / DP path)
    def sgd_step(self, lr, momentum=0.937, grad_scale=1.0, zero_grad=True):
        """torch.optim.SGD(lr, momentum=0.937, nesterov=True).step() [+ zero_grad()] of train.py:156,201-202 as ONE kernel over
        the flat parameter / gradient / momentum buffers."""
        if self.momentum_buf is None:
            self.momentum_buf = torch.zeros_like(self.flat)
        hip.call("ryolo_sgd_nesterov", self.flat.data_ptr(), self.gflat.data_ptr(), self.momentum_buf.data_ptr(), self.n_flat, float(lr),
                 float(momentum), float(grad_scale), 1 if zero_grad else 0, hip.stream())

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, zero_grad=True):
        """torch.optim.Adam(model.parameters(), lr).step() [+ zero_grad()] of train.py:153-154 (`--optimizer Adam`) as ONE kernel over the flat
        parameter / gradient / moment buffers (csrc/elementwise.hip adam_kernel); the step count lives here, as torch keeps it in the optimizer state."""
        if self.adam_state is None:
            self.adam_state = [torch.zeros_like(self.flat), torch.zeros_like(self.flat), 0]
        st = self.adam_state
        st[2] += 1
        hip.call("ryolo_adam", self.flat.data_ptr(), self.gflat.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), self.n_flat, float(lr), float(betas[0]),
                 float(betas[1]), float(eps), st[2], float(grad_scale), 1 if zero_grad else 0, hip.stream())


class NetFunction(torch.autograd.Function):
    """imgs -> 3 head maps [B, na, gs, gs, attrs] fp32.  The parameters are not autograd inputs: their gradients are
    accumulated by the backward tape straight into Runtime.gflat, which `.grad` of every parameter aliases."""

    @staticmethod
    def forward(ctx, imgs, anchor, rt, g):
        if imgs.dtype == torch.float32 and imgs.is_contiguous():
            g.set_image(imgs)
        else:
            g.img.copy_(imgs)
            g.set_image(g.img)
        if not g.static_weights:
            rt.pack()
            if g.wprep:
                g.run(g.wprep)
        g.run(g.fwd, g.timer)
        if g.batch_stats:
            rt.nbt += 1
        # the plan's buffers hold THIS forward's activations until its backward ran: a later forward of the same plan overwrites them
        g.generation = getattr(g, "generation", 0) + 1
        ctx.rt, ctx.g, ctx.generation = rt, g, g.generation
        _register_head_obj(g.heads)
        return tuple(h["out"].detach() for h in g.heads)      # fresh tensor objects over the plan's output buffers

    @staticmethod
    def backward(ctx, *grads):
        rt, g = ctx.rt, ctx.g
        if getattr(ctx, "generation", None) != getattr(g, "generation", None):
            raise RuntimeError("ryolov4_amd: backward of a forward whose activations were overwritten — another forward with the same "
                               "(batch, size, mode) ran on this model in between; run forward -> backward pairs in order (or use a "
                               "different batch size for the interleaved forward)")
        rt.prepare_grads()
        for h, go in zip(g.heads, grads):
            if go is None:
                h["dout"].zero_()
                g.set_head_grad(h, h["dout"])
            elif go.dtype == torch.float32 and go.is_contiguous() and go.shape == h["dout"].shape:
                g.set_head_grad(h, go, _compact_head_grad(go))   # read the loss gradient in place (compact form if it is the fused loss's own map)
            else:
                h["dout"].copy_(go)
                g.set_head_grad(h, h["dout"])
        hook = rt.model._grad_hook
        after = hook.bucket_hooks(rt, g) if hook is not None and hasattr(hook, "bucket_hooks") else None
        g.run(g.bwd, g.timer, after)
        _clear_handoff()                                 # (the compact forms were consumed above; nothing keeps last step's gradient maps alive)
        if hook is not None:
            hook(rt)
        return None, None, None, None
