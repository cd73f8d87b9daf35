"""Drop-in for the reference's lib/loss.py: ComputeCSLLoss (:153-331) and ComputeKFIoULoss (:334-492) on the fused HIP
loss kernels (csrc/loss.hip).  Same constructor `(model, hyp)`, same `.loss_items` dict (present before the first call,
same keys, python floats), same `__call__(outputs, target) -> (loss[1], loss_items)`; works under torch.no_grad() and
with zero targets.  One device->host read per call (the reference does 4-5); pass `sync_items=False` to skip even that
(loss_items then holds 0-d device tensors) — the training benchmark uses this.
FocalLoss (lib/loss.py:10-33; inactive in the reference's configuration, fl_gamma: 0.0, data/hyp.yaml:12) wraps every BCE term
inside the same kernels when hyp['fl_gamma'] > 0.
"""
import os

import torch

from .. import hip
from ..engine import structs as S
from .general import norm_angle, xywhr2xywhrsigma  # noqa: F401  (re-exported like the reference's import at lib/loss.py:7)

# Compact head-gradient handoff (r05).  The dense gradient maps the fused loss returns to autograd are > 99.9 % "zero row with one objectness
# element"; the engine's head backward (csrc/elementwise.hip head_finish_bwd_kernel) can rebuild those rows from the compact per-cell
# objectness gradients + the loss's owner grid and read dense rows only for matched cells (1.33 GB less HBM read per step at the benchmark
# size).  The dense maps stay fully defined — any other consumer of the autograd gradient sees exactly what it saw before; this table only
# tells a consumer that receives the SAME tensor (data_ptr match, checked by engine/runtime.py) where the compact form lives.  One entry
# per dense map of the LAST loss call; replaced by the next call.
_HANDOFF = {}
_HEAD_SPARSE = os.environ.get("RYOLO_HEAD_SPARSE", "1") != "0"


_HEAD_OBJ = {}


def register_head_obj(out, xobj, och):
    """Engine side (engine/runtime.py, after every training forward): `xobj` [B, na, gs, gs] holds out[..., och] — written by the same kernel
    as `out` (ryolo_head_finish_fwd_obj), so the two always describe the same forward."""
    if len(_HEAD_OBJ) > 64:                         # (plans come and go with batch sizes; entries are a pointer and two tensor handles)
        _HEAD_OBJ.clear()
    _HEAD_OBJ[out.data_ptr()] = (out, xobj, och, out._version)


def head_obj_logits(o, och):
    """The compact objectness logits of head map `o` if `o` is — same storage, not modified through torch since — an output buffer of the
    engine, else None (the objectness pass then reads the strided elements of the map itself)."""
    h = _HEAD_OBJ.get(o.data_ptr())
    if h is None or not _HEAD_SPARSE or h[2] != och or o.shape[:4] != h[1].shape or o._version != h[3] or h[0]._version != h[3]:
        return None
    return h[1]


def compact_head_grad(t):
    """(objgrad tensor, owner-grid pointer, objectness channel) of the dense gradient map `t` if `t` is — same storage, untouched — one of the
    maps the last loss call produced, else None.  The entry is CONSUMED: the table drops its references (dense map, compact arrays, loss
    workspace — 1.3 GB at the benchmark size) as soon as the one consumer they were left for has taken them, instead of keeping them until
    the next loss backward.  In-place edits of the map through torch are detected by the version counter; raw-pointer writes into it from
    outside torch are NOT supported (nothing bumps the counter) — a caller that edits the map that way must pass a different tensor."""
    h = _HANDOFF.pop(t.data_ptr(), None)
    if h is None or t.shape != h[0].shape or t._version != h[5] or h[0]._version != h[5]:     # (in-place edits through torch bump the version)
        return None
    return h[1], h[2], h[3], h[4]


def clear_handoff():
    """Engine side, at the end of NetFunction.backward: whatever was not consumed (a head whose gradient was None, a foreign consumer) is dropped."""
    _HANDOFF.clear()


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, targets, o0, o1, o2):
        grads, items = owner._run([o0, o1, o2], targets, True)
        ctx.grads = grads
        ctx.crit, ctx.gen, ctx.compact = owner, owner._gen, owner._compact
        ctx.consumed = False
        owner._last_items = items
        return items[4:5].clone()

    @staticmethod
    def backward(ctx, go):
        if ctx.consumed:
            # the saved gradients are scaled IN PLACE below (they are 1.3 GB at the benchmark size: no second copy); a second backward
            # through the same node (retain_graph=True) would apply the incoming scale twice — refuse instead of returning wrong numbers
            raise RuntimeError("ryolov4_amd loss: backward through the same loss node twice is not supported (gradients are "
                               "scaled in place); call the criterion again for a second backward")
        ctx.consumed = True
        g = ctx.grads
        # d(total)/d(logits) was produced by the fused kernel; chain rule with the incoming gradient (a [1] device tensor, 1.0 after a
        # plain loss.backward()): scaled in place on the device, and skipped there when the scalar is exactly 1 (no host read)
        sc = go.detach().reshape(-1)[:1].to(device=g[0].device, dtype=torch.float32).contiguous()
        _HANDOFF.clear()
        arrs = list(g)
        if ctx.compact is not None and ctx.crit._gen == ctx.gen:         # (a later call of the criterion reused the workspace: owner grids gone)
            objgrad, owners, och, ws = ctx.compact
            arrs += list(objgrad)                                         # same scalar, same multiply: compact and dense forms stay bit-identical
            for gi, og, ow in zip(g, objgrad, owners):
                _HANDOFF[gi.data_ptr()] = (gi, og, ow, och, ws, gi._version)
        ptrs, lens = (S.P * 8)(), (S.L * 8)()
        for k, t in enumerate(arrs):
            ptrs[k], lens[k] = t.data_ptr(), t.numel()
        hip.call("ryolo_loss_grad_scale_multi", ptrs, lens, len(arrs), sc.data_ptr(), hip.stream())       # one launch (exits on the device when the scalar is 1)
        return (None, None) + tuple(g)


class _ComputeLossBase:
    MODE = None
    KEYS = ()

    def __init__(self, model, hyp):
        self.fl_gamma = float(hyp.get("fl_gamma", 0.0))        # > 0: FocalLoss around every BCE term (lib/loss.py:167-171, :345-348)
        self.hyp = {k: float(hyp[k]) for k in ("box", "obj", "cls", "obj_pw", "cls_pw")}
        self.lambda_theta = 0.5                     # lib/loss.py:160
        self.anchors_list = model.anchors
        self.na = len(model.anchors[0])
        self.nl = 3
        self.nc = model.nc
        self.loss_items = {k: 0 for k in self.KEYS}
        self._ws = None
        self._items = None
        self._last_items = None
        self._drop_host = self._drop_event = None
        self._drop_batch = 0
        self._gen = 0
        self._compact = None

    def _params(self, outputs, targets, compute_grad, grads):
        p = S.LossParams()
        p.mode, p.nc, p.na = self.MODE, self.nc, self.na
        p.batch = outputs[0].shape[0]
        p.nt = targets.shape[0]
        p.tcols = targets.shape[1] if targets.dim() == 2 else 0
        p.targets = targets.data_ptr() if p.nt else None
        for i in range(3):
            p.head[i] = outputs[i].data_ptr()
            p.grad[i] = grads[i].data_ptr() if compute_grad else None
            p.gs[i] = outputs[i].shape[2]
            for a, an in enumerate(self.anchors_list[i]):
                for k in range(len(an)):
                    p.anchors[i][a][k] = float(an[k])
        p.box, p.obj, p.cls = self.hyp["box"], self.hyp["obj"], self.hyp["cls"]
        p.theta_gain, p.obj_pw, p.cls_pw = self.lambda_theta, self.hyp["obj_pw"], self.hyp["cls_pw"]
        p.compute_grad = 1 if compute_grad else 0
        p.fl_gamma, p.fl_alpha = self.fl_gamma, 0.25             # FocalLoss(loss_fcn, gamma) keeps its default alpha (lib/loss.py:11)
        return p

    def _run(self, outputs, targets, compute_grad):
        S.check_layouts()
        dev = outputs[0].device
        attrs = self.nc + (185 if self.MODE == 0 else 6)              # (MODE 2 shares the kfiou head layout)
        outs = []
        for o in outputs:
            hip.require_device(o, "loss")
            if o.dim() != 5 or o.shape[1] != self.na or o.shape[4] != attrs or o.shape[2] != o.shape[3]:
                raise RuntimeError("loss: expected head maps [B, na, gs, gs, attrs], got {}".format(tuple(o.shape)))
            outs.append(o.detach().float().contiguous())
        targets = targets.to(dev).float().contiguous()
        grads = [torch.empty_like(o) for o in outs] if compute_grad else [None] * 3
        objgrad = [torch.empty(o.shape[:4], dtype=torch.float32, device=dev) for o in outs] if compute_grad and _HEAD_SPARSE else None
        self._gen += 1
        self._compact = None
        items = torch.empty(6, dtype=torch.float32, device=dev)     # reg, conf, cls, theta, total, dropped target rows
        p = self._params(outs, targets, compute_grad, grads)
        self._last_shape, self._last_gs = (p.nt, self.na, p.batch), [o.shape[2] for o in outs]
        need = S.Z()
        hip.call("ryolo_loss_workspace_bytes", p, need)
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != dev:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        p.ws, p.ws_bytes, p.items = self._ws.data_ptr(), self._ws.numel(), items.data_ptr()
        self._last_params = p
        if objgrad is not None:
            for i in range(3):
                p.objgrad[i] = objgrad[i].data_ptr()
        xobj = [head_obj_logits(o, 4 if self.MODE == 0 else 5) for o in outs]     # (kept alive until the launch below is enqueued)
        for i in range(3):
            p.headobj[i] = xobj[i].data_ptr() if xobj[i] is not None else None
        self._used_head_obj = [x is not None for x in xobj]
        hip.call("ryolo_loss", p, hip.stream())
        if objgrad is not None:
            own = (S.P * 3)()
            hip.call("ryolo_loss_owner_grids", p, own)
            self._compact = (objgrad, [int(own[i]) for i in range(3)], 4 if self.MODE == 0 else 5, self._ws)
        return grads, items

    def _check_previous_call(self):
        """Asynchronous path (sync_items=False): the dropped-row count of the previous call was copied to pinned host memory behind an
        event; by the next call that event has completed, so reading it does not stall the host.  A collate / shard re-indexing bug
        therefore raises one step late instead of training quietly on fewer labels.  `flush()` checks the last call explicitly."""
        ev = self._drop_event
        if ev is None:
            return
        self._drop_event = None
        ev.synchronize()
        n = int(self._drop_host[0])
        if n > 0:
            raise IndexError("loss: {} target row(s) of the previous call carried an image index outside [0, {}) (targets[:, 0])".format(n, self._drop_batch))

    def flush(self):
        """Wait for and check the dropped-target count of the last sync_items=False call (end of an epoch / before a checkpoint)."""
        self._check_previous_call()

    def debug_matches(self):
        """Test hook: the (b, a, gj, gi, cls, tidx, cell) records of the last call, per scale, read back from the workspace
        (layout = carve() in csrc/loss.hip)."""
        cnt_p, rec_p = (S.P * 3)(), (S.P * 3)()
        hip.call("ryolo_loss_match_records", self._last_params, cnt_p, rec_p)
        torch.cuda.synchronize()
        base, ws = self._ws.data_ptr(), self._ws
        out = []
        for i in range(3):
            cnt = int(ws[cnt_p[i] - base:cnt_p[i] - base + 4].view(torch.int32)[0])
            o = rec_p[i] - base
            out.append(ws[o:o + cnt * 32].view(torch.int32).reshape(-1, 8).cpu().numpy().astype("int64"))
        return out

    def __call__(self, outputs, target, sync_items=True):
        needs_grad = torch.is_grad_enabled() and any(o.requires_grad for o in outputs)
        if needs_grad:
            loss = _LossFn.apply(self, target, *outputs)
            items = self._last_items
        else:
            _, items = self._run(list(outputs), target, False)
            loss = items[4:5].clone()
        names = ("reg_loss", "conf_loss", "cls_loss", "theta_loss", "total_loss")
        self.dropped_targets = items[5]                # device scalar: rows whose image index is outside [0, batch)
        # inside a stream capture (a whole training step recorded into one hipGraph: tools/bench_graph_step.py) nothing may wait on the host and
        # the pinned read-back would be replayed without anyone looking at it: the count stays on the device (`dropped_targets`)
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            self._check_previous_call()                # sync_items=False: the PREVIOUS call's count, long finished by now (no stall)
        if not sync_items and not capturing:
            if self._drop_host is None:
                self._drop_host = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._drop_host.copy_(items[5:6], non_blocking=True)
            self._drop_event = torch.cuda.current_stream().record_event()
            self._drop_batch = outputs[0].shape[0]
        if sync_items:
            vals = items.tolist()                      # the single device->host read of the step
            if vals[5] > 0:
                # the reference indexes pi[b, a, gj, gi] (lib/loss.py:209,385) and raises here; a collate / shard re-indexing bug must
                # not train quietly on fewer labels
                raise IndexError("loss: {} target row(s) carry an image index outside [0, {}) (targets[:, 0])".format(int(vals[5]), outputs[0].shape[0]))
            self.loss_items.update({k: vals[names.index(k)] for k in self.KEYS})
        else:
            self.loss_items.update({k: items[names.index(k)] for k in self.KEYS})
        return loss, self.loss_items


class ComputeCSLLoss(_ComputeLossBase):
    MODE = 0
    KEYS = ("reg_loss", "theta_loss", "conf_loss", "cls_loss", "total_loss")      # lib/loss.py:183-189


class ComputeKFIoULoss(_ComputeLossBase):
    MODE = 1
    KEYS = ("reg_loss", "conf_loss", "cls_loss", "total_loss")                    # lib/loss.py:361-366


class ComputeSL1IoULoss(_ComputeLossBase):
    """EXTRA mode, not in the reference's code: the smooth-L1-IoU box regression its Readme names (Readme.md:4,12-13: "combining
    smooth-L1-IoU loss function proposed by R3Det", formula images only; BASELINE config C2).  Same constructor / call / loss_items
    contract as ComputeKFIoULoss and the SAME network (Yolo(mode='kfiou'): 18 rotated anchors, attrs = nc + 6); only the regression
    term differs: per match (L_sl1 / |L_sl1|) * |-log SkewIoU| with the exact rotated IoU of csrc/rotated_iou.h (definition:
    DESIGN.md §4.3; oracle: oracle/ref_ops.sl1iou_loss, fp64, written for this build — no reference oracle exists)."""
    MODE = 2
    KEYS = ("reg_loss", "conf_loss", "cls_loss", "total_loss")
