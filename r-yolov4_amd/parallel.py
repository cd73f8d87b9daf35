"""Plain data parallel over the GPUs of one node: one process per GPU, full replica, images sharded, gradients summed with
RCCL all-reduce over xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).  The reference has no distributed code at
all; the step semantics follow train.py:186-202 with the DDP convention (gradients averaged over ranks,
BatchNorm statistics per rank, per-rank loss means).

Because every parameter gradient lives in ONE flat fp32 buffer (engine/runtime.py), the all-reduce is a handful of large
messages instead of ~290 small ones: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so few, large, bucketed collectives
are what it wants.  The 1/world scaling is folded into the fused SGD kernel (grad_scale), not a separate pass.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        be = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if be == "nccl":
            # bind the communicator to the device the caller selected (torch.cuda.set_device before this call): no rank -> GPU guessing
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(be, rank=rank, world_size=world, **kw)
    return rank, local, world


def bucket_bounds(n, bucket_elems):
    """[0, n) cut into contiguous buckets of at most bucket_elems (the last one may be shorter)."""
    out, a = [], 0
    while a < n:
        b = min(n, a + bucket_elems)
        out.append((a, b))
        a = b
    return out


def allreduce_flat(flat, bucket_bytes=64 << 20, async_op=False):
    """Sum `flat` over all ranks in buckets of ~bucket_bytes; returns the work handles when async_op."""
    if not (dist.is_available() and dist.is_initialized()):
        return []
    elems = max(1, bucket_bytes // flat.element_size())
    works = []
    for a, b in bucket_bounds(flat.numel(), elems):
        w = dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def shard_range(n_items, rank, world):
    """Contiguous shard of a global batch (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _Reducer:
    """Gradient all-reduce overlapped with the backward tape.  The flat fp32 gradient buffer is cut into buckets at parameter
    boundaries (~bucket_bytes each); Graph.grad_ready_points tells after which backward launch a bucket is final, and right after
    that launch is enqueued the bucket's all-reduce is issued on a side stream behind an event — RCCL moves the tail layers'
    gradients over xGMI while the MFMA kernels of the earlier layers still run.  __call__ (end of backward) reduces whatever no
    launch claimed and makes the compute stream wait for the collectives."""

    def __init__(self, bucket_bytes):
        self.bucket_bytes = bucket_bytes
        self.bounds = None
        self.plans = {}
        self.works = []
        self.done = set()
        self.side = None

    def _bounds(self, rt):
        if self.bounds is None:
            target = max(1, self.bucket_bytes // 4)
            starts = sorted(v[0] for v in rt._pslice.values())        # parameter start offsets in the flat buffers
            cuts, last = [0], 0
            n = rt.gflat.numel()
            for o in starts:
                if o - last >= target:
                    cuts.append(o)
                    last = o
            cuts.append(n)
            self.bounds = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        return self.bounds

    def _launch(self, rt, k):
        if k in self.done:
            return
        self.done.add(k)
        a, b = self.bounds[k]
        view = rt.gflat[a:b]
        if view.is_cuda:
            if self.side is None:
                self.side = torch.cuda.Stream(device=view.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(view.device))
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                se = getattr(rt, "side_event", None)          # weight gradients of this bucket still running on the engine's
                if se is not None:                            # second stream (Graph.run): the collective waits for them too
                    self.side.wait_event(se)
                self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM)

    def bucket_hooks(self, rt, g):
        """{backward tape index: callable} for Graph.run."""
        self.works, self.done = [], set()
        bounds = self._bounds(rt)
        plan = self.plans.get(id(g))
        if plan is None:
            ready = g.grad_ready_points(bounds)
            plan = {}
            for k, idx in enumerate(ready):
                if idx >= 0:
                    plan.setdefault(idx, []).append(k)
            self.plans[id(g)] = plan
        return {idx: (lambda ks=ks: [self._launch(rt, k) for k in ks]) for idx, ks in plan.items()}

    def __call__(self, rt):
        for k in range(len(self._bounds(rt))):
            self._launch(rt, k)
        for w in self.works:
            w.wait()                                   # the current (compute) stream waits for the collective
        self.works = []


class _Hook:
    """What Yolo._grad_hook holds: forwards to the reducer unless the owner is inside `no_sync()`."""

    def __init__(self, owner, reducer, overlapped):
        self.owner, self.reducer, self.overlapped = owner, reducer, overlapped

    def bucket_hooks(self, rt, g):
        if not self.owner._sync or not self.overlapped:
            return None
        return self.reducer.bucket_hooks(rt, g)

    def __call__(self, rt):
        if self.owner._sync:
            self.reducer(rt)


class DataParallel:
    """Wraps a ryolov4_amd Yolo: broadcasts rank-0 parameters once, all-reduces the flat gradient buffer in ~25 MB buckets
    overlapped with the backward tape (overlap=False: one pass at the end of backward), and exposes `grad_scale` = 1/world for
    the fused SGD step.

    Gradient accumulation (train.py:198-202, `accumulate > 1`): the flat buffer keeps accumulating across backward passes, and the
    all-reduce is IN PLACE — reducing after every micro-step would sum the already-reduced earlier micro-steps over the ranks again
    (world*G1 + G2).  Run every micro-step but the last under `with dp.no_sync():` (same contract as torch DDP); the last backward
    reduces the accumulated sum once."""

    def __init__(self, model, bucket_bytes=25 << 20, overlap=True, force=False):
        self.model = model
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        rt = model.runtime()
        if self.world > 1:
            dist.broadcast(rt.flat, src=0)
            for b in model.buffers():
                if b.dtype.is_floating_point:
                    dist.broadcast(b, src=0)
        self.overlap = overlap
        self._sync = True
        self._reducer = _Reducer(bucket_bytes)
        # force: install the hooks even for a single rank (exercises the RCCL path on a one-GPU box: tools/dp_check.py)
        model._grad_hook = _Hook(self, self._reducer if overlap else self._reduce, overlap) if (self.world > 1 or force) else None
        self.grad_scale = 1.0 / self.world

    def _reduce(self, rt):
        allreduce_flat(rt.gflat, self.bucket_bytes)

    def no_sync(self):
        """Context manager: backward passes inside it only accumulate into the local flat gradient buffer (no collective)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self._sync = self._sync, False
            try:
                yield
            finally:
                self._sync = old
        return ctx()

    def __call__(self, *a, **k):
        return self.model(*a, **k)
