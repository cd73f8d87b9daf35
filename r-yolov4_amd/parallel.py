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
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
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
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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


class DataParallel:
    """Wraps a ryolov4_amd Yolo: broadcasts rank-0 parameters once, all-reduces the flat gradient buffer at the end of every
    backward (hooked inside the engine's single autograd node), and exposes `grad_scale` = 1/world for the fused SGD step."""

    def __init__(self, model, bucket_bytes=64 << 20):
        self.model = model
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        rt = model.runtime()
        if self.world > 1:
            dist.broadcast(rt.flat, src=0)
            for b in model.buffers():
                if b.dtype.is_floating_point:
                    dist.broadcast(b, src=0)
        model._grad_hook = self._reduce
        self.grad_scale = 1.0 / self.world

    def _reduce(self, rt):
        allreduce_flat(rt.gflat, self.bucket_bytes)

    def __call__(self, *a, **k):
        return self.model(*a, **k)
