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


def _parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def _fmt_cpulist(cpus):
    cpus, parts, i = sorted(cpus), [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(parts)


def _gpu_local_cpus(index):
    """CPUs of the NUMA node the GPU `index` hangs off: /sys/bus/pci/devices/<domain:bus:dev.0>/local_cpulist (the same topology
    `rocm-smi --showtoponuma` prints); None when the device or the file is not there (CPU-only box, container without sysfs)."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        cpus = _parse_cpulist(open(f"{base}/local_cpulist").read())
        try:
            node = int(open(f"{base}/numa_node").read())
        except (OSError, ValueError):
            node = -1
        return (cpus, node) if cpus else None
    except Exception:                                                   # no GPU / no sysfs: the caller splits what it has evenly
        return None


def plan_affinity(local_rank, n_local, device_of_rank=None, allowed=None, gpu_cpus=_gpu_local_cpus):
    """Which host cores local rank `local_rank` of `n_local` gets: the cores of ITS GPU's NUMA node, and when several ranks' GPUs share a
    node (MI355X chassis: 8 GPUs on 2 sockets) an equal contiguous slice of them per rank — loader decode threads, the launch thread and
    RCCL's proxy threads of a rank then stay on the socket whose PCIe root the GPU is on, and no two ranks fight for a core.  Pure
    function of the topology (every rank computes the same plan without talking): returns {"cpus": [...], "numa_node", "source"}."""
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    device_of_rank = device_of_rank or list(range(n_local))
    info = [gpu_cpus(device_of_rank[r]) for r in range(n_local)]
    if any(i is None for i in info):
        per = max(1, len(allowed) // n_local)
        mine = allowed[local_rank * per:(local_rank + 1) * per] or allowed
        return {"cpus": mine, "numa_node": None, "source": "even split of the allowed cores (no PCI topology in sysfs)"}
    sets = [tuple(sorted(set(c) & set(allowed)) or allowed) for c, _ in info]
    sharers = [r for r in range(n_local) if sets[r] == sets[local_rank]]
    pool = list(sets[local_rank])
    per = max(1, len(pool) // len(sharers))
    k = sharers.index(local_rank)
    mine = pool[k * per:(k + 1) * per] or pool
    return {"cpus": mine, "numa_node": info[local_rank][1], "source": "sysfs local_cpulist of the rank's GPU, split over the ranks of that node"}


def bind_rank(local_rank, n_local, device_of_rank=None, max_threads=16):
    """Pin this process (and the threads it will start) to its slice of plan_affinity() and size torch's host thread pool to it (the
    loader's decode pool reads `decode_threads`).  RYOLO_BIND=0 turns it off.  Returns the record bench.py prints per rank."""
    if os.environ.get("RYOLO_BIND", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return {"bound": False, "cpus": _fmt_cpulist(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    plan = plan_affinity(local_rank, n_local, device_of_rank)
    try:
        os.sched_setaffinity(0, plan["cpus"])
        bound = True
    except OSError:
        bound = False
    threads = max(1, min(max_threads, len(plan["cpus"])))
    torch.set_num_threads(threads)
    return {"bound": bound, "cpus": _fmt_cpulist(plan["cpus"]), "n_cpus": len(plan["cpus"]), "numa_node": plan["numa_node"],
            "host_threads": threads, "decode_threads": max(1, min(8, len(plan["cpus"]) - 1)), "source": plan["source"]}


SIDE_CUS_SINGLE = 96            # CUs the CU-exclusive 8-wave weight-gradient kernels hold when no collective runs beside them
RCCL_CHANNELS_DEFAULT = 8       # one ring channel per XCD: each channel is ONE workgroup of the collective's kernel


def plan_partition(world, channels=None, side_cus=None):
    """How the 256 CUs are shared in a data-parallel step (pure function: every rank computes the same answer without talking).

    Single GPU: the weight-gradient side stream owns SIDE_CUS_SINGLE = 96 CUs (one CU-exclusive 8-wave workgroup each: RYOLO_W3_V8_BLOCKS,
    RYOLO_WGRAD_8W_BLOCKS), the main stream the other 160.  With world > 1 the bucket all-reduces run on a THIRD stream while backward is
    still going, and an RCCL channel is a workgroup that needs a CU slot: behind 96 CU-exclusive workgroups + a main stream that fills the
    other 160 CUs a collective would queue until a weight-gradient launch drains.  So the side partition shrinks by the channel count:
    side = 96 - channels (88 for the default 8 channels), and the channel count is pinned (NCCL_MIN = NCCL_MAX) so that the sizing holds.
    Explicit RYOLO_W3_V8_BLOCKS / RYOLO_WGRAD_8W_BLOCKS / RYOLO_RCCL_CHANNELS in the environment win over the rule."""
    if world <= 1:
        return {"world": world, "rccl_channels": 0, "side_cus": SIDE_CUS_SINGLE if side_cus is None else int(side_cus),
                "main_cus": 256 - (SIDE_CUS_SINGLE if side_cus is None else int(side_cus))}
    ch = RCCL_CHANNELS_DEFAULT if channels is None else max(1, min(32, int(channels)))
    side = max(32, SIDE_CUS_SINGLE - ch) if side_cus is None else int(side_cus)
    return {"world": world, "rccl_channels": ch, "side_cus": side, "main_cus": 256 - side - ch}


_PARTITION = {}


def apply_partition(world):
    """Put plan_partition(world) into the environment BEFORE the HIP library reads its grid knobs (function-local statics, first plan) and
    before the communicator exists.  Values already set by the caller are kept and reported as such."""
    env = os.environ
    ch_req = env.get("RYOLO_RCCL_CHANNELS")
    plan = plan_partition(world, channels=int(ch_req) if ch_req else None)
    src = {}
    if world > 1:
        for k in ("RYOLO_W3_V8_BLOCKS", "RYOLO_WGRAD_8W_BLOCKS"):
            src[k] = "environment" if k in env else "rule"
            env.setdefault(k, str(plan["side_cus"]))
        env["RYOLO_RCCL_CHANNELS"] = str(plan["rccl_channels"])
    w3 = int(env.get("RYOLO_W3_V8_BLOCKS", SIDE_CUS_SINGLE))
    w1 = int(env.get("RYOLO_WGRAD_8W_BLOCKS", SIDE_CUS_SINGLE))
    _PARTITION.clear()
    _PARTITION.update(plan, w3_v8_blocks=w3, wgrad_8w_blocks=w1, set_by=src,
                      rule="side = 96 - rccl_channels when world > 1 (plan_partition); explicit RYOLO_* values win")
    return dict(_PARTITION)


def partition():
    """What apply_partition decided for this process (bench.py prints it in the `distributed` block)."""
    return dict(_PARTITION) if _PARTITION else apply_partition(int(os.environ.get("WORLD_SIZE", "1")))


def gpu_telemetry(index=0, timeout=10.0):
    """Clocks and socket power of one GPU as rocm-smi reports them right now (bench.py samples every rank once, in the middle of a run of
    steps: eight GPUs in one chassis at ~1.3 kW each is the first thing a SCALE run tests).  Never raises: {"error": ...} when rocm-smi is
    absent or does not answer in `timeout` seconds."""
    import json
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        r = subprocess.run([exe, "-d", str(index), "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=timeout)
        doc = json.loads(r.stdout[r.stdout.index("{"):])
        card = next(iter(doc.values()))
        out = {}
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl and "level" in kl:
                out["sclk"] = v
            elif "mclk" in kl and "level" in kl:
                out["mclk"] = v
            elif "power" in kl and "(w)" in kl:
                out["power_w"] = float(v)
        return out or {"error": "no clock / power fields in rocm-smi --json", "keys": sorted(card)[:12]}
    except Exception as e:                                                   # noqa: BLE001 (telemetry must never take a bench run down)
        return {"error": f"{type(e).__name__}: {str(e)[:120]}"}


def rccl_env():
    """RCCL knobs that matter beside two compute streams: the channel count bounds how many CUs the collective's kernels occupy while
    backward is still running (each channel is one workgroup).  RYOLO_RCCL_CHANNELS=n sets NCCL_MIN/MAX_NCHANNELS before the communicator
    exists (default when world > 1: 8, with the weight-gradient partition shrunk to match — plan_partition); whatever is in effect is
    reported by bench.py."""
    n = os.environ.get("RYOLO_RCCL_CHANNELS")
    if n:
        os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(int(n))
    return {k: os.environ[k] for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_ALGO", "NCCL_PROTO", "GPU_MAX_HW_QUEUES") if k in os.environ}


def pick_wire(batch_per_gpu, world, requested="auto"):
    """fp32 or bf16 gradient buckets on xGMI, by rule: the all-reduce moves 151.5 MB (fp32) per step whatever the batch; against the
    ~73 ms step of 64 images per GPU it is hidden under backward, against the ~15 ms step of 8 images per GPU (the reference's nominal
    batch 64 over 8 GPUs) it is not — there bf16 on the wire (fp32 accumulate on receive, replicas still bit-identical) halves it."""
    if requested != "auto":
        return requested
    return "bf16" if (world > 1 and batch_per_gpu <= 16) else "fp32"


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    apply_partition(world)                       # CU shares of the two compute streams and the collective (before the library plans anything)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        rccl_env()
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


_A2A = {}


def _has_all_to_all(probe):
    """Decided ONCE per (backend, device type), collectively (every rank runs the same probe at the same point), never by catching an
    error around a real bucket: on RCCL a failed collective on some ranks followed by a different collective on the others is a hang."""
    key = (dist.get_backend(), probe.device.type)
    if key not in _A2A:
        if key[0] == "nccl":
            _A2A[key] = True
        else:
            w = dist.get_world_size()
            a = torch.zeros(w, dtype=probe.dtype, device=probe.device)
            try:
                dist.all_to_all_single(torch.empty_like(a), a)
                _A2A[key] = True
            except RuntimeError:
                _A2A[key] = False
    return _A2A[key]


def allreduce_bf16_wire(view, scratch=None):
    """Sum the fp32 tensor `view` over all ranks moving bf16 over the wire and accumulating in fp32 ON RECEIVE (75.8 MB
    instead of 151.5 MB per step for yolov7).  A direct reduce-scatter + all-gather over the point-to-point links, not a ring:
      1. the bucket is cast to bf16 and cut into `world` shards; all_to_all sends shard j to rank j (7/8 of the bf16 bucket leaves a GPU);
      2. rank j adds the `world` versions of its shard in fp32, in rank order — one owner per element, so every rank ends up with the
         same bits and the result does not depend on the collective's internal schedule — and rounds the sum to bf16 once;
      3. all_gather of the summed shards (another 7/8 of the bf16 bucket), cast back into the fp32 buffer.
    What differs from the fp32 all-reduce: each rank's contribution is rounded to bf16 (8 bits of mantissa) before the sum and the sum once
    after it; the accumulation itself is exact fp32.  Runs on the caller's current stream (collectives are stream-ordered on RCCL)."""
    world = dist.get_world_size()
    n = view.numel()
    shard = -(-n // world)
    padded = shard * world
    if scratch is None or scratch[0].numel() < padded:
        dev = view.device
        scratch = (torch.empty(padded, dtype=torch.bfloat16, device=dev), torch.empty(padded, dtype=torch.bfloat16, device=dev),
                   torch.empty(shard, dtype=torch.bfloat16, device=dev))
    send, recv, mine = scratch[0][:padded], scratch[1][:padded], scratch[2][:shard]
    send[:n].copy_(view)                                           # fp32 -> bf16 (round to nearest even)
    if padded > n:
        send[n:].zero_()
    if _has_all_to_all(send):
        dist.all_to_all_single(recv, send)
    else:                                                          # gloo has no all_to_all on some builds: same data movement by all_gather
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send)
        r = dist.get_rank()
        recv.copy_(torch.cat([p[r * shard:(r + 1) * shard] for p in parts]))
    acc = recv.view(world, shard)[0].float()
    for k in range(1, world):                                      # fixed order: rank 0, 1, 2, ...
        acc += recv.view(world, shard)[k].float()
    mine.copy_(acc)
    dist.all_gather_into_tensor(send, mine)
    view.copy_(send[:n])
    return scratch


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

    def __init__(self, bucket_bytes, wire="fp32"):
        if wire not in ("fp32", "bf16"):
            raise ValueError("wire must be 'fp32' or 'bf16'")
        self.bucket_bytes = bucket_bytes
        self.wire = wire
        self.scratch = {}                 # bf16 staging buffers PER STREAM (side stream / caller's stream): never shared across streams
        self.events = []
        self.bounds = None
        self.plans = {}
        self.done = set()
        self.side = None
        self.timing = False               # bench.py: HIP-event pairs around every bucket's collective on the stream it runs on
        self.stamps = []

    def collective_ms(self):
        """Sum of the bucket collectives' durations since the last call (needs .timing and a device synchronize before): time on the
        side stream from 'bucket final' to 'sum back in the buffer', waiting for slower peers included."""
        ms = sum(a.elapsed_time(b) for a, b in self.stamps)
        self.stamps = []
        return ms

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

    def _collective(self, view, key):
        if self.wire == "bf16":
            self.scratch[key] = allreduce_bf16_wire(view, self.scratch.get(key))
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True).wait()     # RCCL: the CURRENT stream waits, the host does not

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
                if self.timing:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record(self.side)
                self._collective(view, "side")                # buckets run in order on this stream and share its scratch
                if self.timing:
                    t1 = torch.cuda.Event(enable_timing=True)
                    t1.record(self.side)
                    self.stamps.append((t0, t1))
                self.events.append(self.side.record_event())
        else:
            self._collective(view, "host")

    def bucket_hooks(self, rt, g):
        """{backward tape index: callable} for Graph.run."""
        self.events, self.done = [], set()
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
        for ev in self.events:
            torch.cuda.current_stream().wait_event(ev)    # the compute stream waits for the collectives, the host does not
        self.events = []


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
    the fused SGD step.  wire="bf16" halves the bytes on xGMI (bf16 on the wire, fp32 accumulation on receive).

    Gradient accumulation (train.py:198-202, `accumulate > 1`): the flat buffer keeps accumulating across backward passes, and the
    all-reduce is IN PLACE — reducing after every micro-step would sum the already-reduced earlier micro-steps over the ranks again
    (world*G1 + G2).  Run every micro-step but the last under `with dp.no_sync():` (same contract as torch DDP); the last backward
    reduces the accumulated sum once."""

    def __init__(self, model, bucket_bytes=25 << 20, overlap=True, force=False, wire="fp32"):
        """wire="bf16": gradients cross xGMI as bf16 and are accumulated in fp32 on receive (allreduce_bf16_wire): half the bytes of the
        fp32 all-reduce; every rank still ends with bit-identical gradients."""
        self.model = model
        self.wire = wire
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
        self._reducer = _Reducer(bucket_bytes, wire)
        # force: install the hooks even for a single rank (exercises the RCCL path on a one-GPU box: tools/dp_check.py)
        model._grad_hook = _Hook(self, self._reducer if overlap else self._reduce, overlap) if (self.world > 1 or force) else None
        self.grad_scale = 1.0 / self.world

    def _reduce(self, rt):
        if self.wire == "bf16":
            elems = max(1, self.bucket_bytes // 4)
            sc = self._reducer.scratch
            for a, b in bucket_bounds(rt.gflat.numel(), elems):
                sc["main"] = allreduce_bf16_wire(rt.gflat[a:b], sc.get("main"))     # the caller's stream has its own staging buffers
        else:
            allreduce_flat(rt.gflat, self.bucket_bytes)

    def set_wire(self, wire):
        """Switch the bucket format between steps (bench.py: fp32 for the 64-image step, bf16 for the 8-image one)."""
        if wire not in ("fp32", "bf16"):
            raise ValueError("wire must be 'fp32' or 'bf16'")
        self.wire = self._reducer.wire = wire

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
