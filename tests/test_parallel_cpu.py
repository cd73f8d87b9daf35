"""CPU tests of the data-parallel host logic with world_size 2 over gloo (the N>1 path the driver runs over RCCL):
bucketed all-reduce of a flat gradient buffer, shard arithmetic, rendezvous from the torch.distributed.run env contract."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ryolov4_amd import parallel
    r, l, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    parallel.allreduce_flat(flat, bucket_bytes=1024)                    # 4 buckets of 256 floats (last one shorter)
    expect = torch.arange(1000, dtype=torch.float32) * sum(range(1, world + 1))
    ok = torch.equal(flat, expect)
    works = parallel.allreduce_flat(flat, bucket_bytes=4096, async_op=True)
    for wk in works:
        wk.wait()
    ok = ok and torch.equal(flat, expect * world)
    # gradient averaging convention: sum over ranks then grad_scale = 1/world inside the SGD step
    p, g, buf = torch.ones(8), torch.full((8,), float(rank + 1)), torch.zeros(8)
    dist.all_reduce(g)
    gs = g / world
    buf = 0.937 * buf + gs
    p = p - 0.01 * (gs + 0.937 * buf)
    q.put((rank, ok, p.tolist()))
    dist.destroy_process_group()


def test_allreduce_flat_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]                                        # replicas stay identical after the step
    g_mean = 1.5
    assert abs(res[0][2][0] - (1 - 0.01 * (g_mean + 0.937 * g_mean))) < 1e-6


def test_shard_and_buckets():
    from ryolov4_amd import parallel
    for n, w in ((64, 8), (10, 4), (7, 8)):
        spans = [parallel.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    assert parallel.bucket_bounds(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert parallel.bucket_bounds(0, 4) == []


def test_bucket_ready_points_cover_every_gradient():
    """Plan built on CPU (no launches): every parameter's gradient is written by some backward launch, and the bucket that holds
    it becomes ready no earlier than that launch; the tail layers' buckets are ready long before the first layers' (that window is
    what the overlapped all-reduce uses)."""
    from ryolov4_amd import parallel
    from ryolov4_amd.engine.runtime import Runtime
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG
    m = Yolo(16, CFG, "kfiou", "yolov7")
    m.train()
    rt = Runtime(m, torch.device("cpu"))
    g = rt.graph(2, 64, 64, True)
    red = parallel._Reducer(bucket_bytes=8 << 20)
    bounds = red._bounds(rt)
    assert bounds[0][0] == 0 and bounds[-1][1] == rt.gflat.numel() and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    assert len(bounds) >= 8
    ready = g.grad_ready_points(bounds)
    assert min(ready) >= 0
    written = {}
    for idx, offs in g.grad_writes:
        for o in offs:
            written[o] = max(written.get(o, -1), idx)
    starts = sorted(v[0] for v in rt._pslice.values())
    assert set(starts) <= set(written), "a parameter gradient that no backward launch writes"
    for o, idx in written.items():
        k = max(i for i, (a, _) in enumerate(bounds) if a <= o)
        assert ready[k] >= idx
    assert ready[-1] < ready[0], "last layers' gradients are final before the first layers'"
    assert ready[-1] < len(g.bwd) // 2


def _reducer_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ryolov4_amd import parallel
    parallel.init_from_env(backend="gloo")

    class RT:                                                     # the two attributes the reducer reads
        gflat = torch.arange(5000, dtype=torch.float32) * (rank + 1)
        _pslice = {i: (o, 10, None) for i, o in enumerate(range(0, 5000, 64))}

    class G:
        grad_writes = [(3, [4992]), (7, [2048, 1024]), (9, [0])]

        def grad_ready_points(self, bounds):
            from ryolov4_amd.engine.graph import Graph
            return Graph.grad_ready_points(self, bounds)

    red = parallel._Reducer(bucket_bytes=4096)                    # 1024-float buckets cut at parameter starts
    hooks = red.bucket_hooks(RT, G())
    fired = []
    for i in range(12):                                           # stand-in for Graph.run
        if i in hooks:
            hooks[i]()
            fired.append(i)
    red(RT)                                                       # end of backward: unclaimed buckets + wait
    expect = torch.arange(5000, dtype=torch.float32) * sum(range(1, world + 1))
    q.put((rank, torch.equal(RT.gflat, expect), fired, len(red.bounds)))
    dist.destroy_process_group()


def test_overlapped_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for _, ok, fired, nb in res:
        assert ok and fired == [3, 7, 9] and nb == 5


def _nosync_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ryolov4_amd import parallel
    parallel.init_from_env(backend="gloo")

    class RT:
        flat = torch.zeros(300)
        gflat = torch.zeros(300)
        _pslice = {i: (o, 10, None) for i, o in enumerate(range(0, 300, 64))}

    class Model:
        _grad_hook = None

        def runtime(self):
            return RT

        def buffers(self):
            return []

    m = Model()
    dp = parallel.DataParallel(m, bucket_bytes=512, overlap=False)
    g1 = torch.arange(300, dtype=torch.float32) * (rank + 1)
    g2 = torch.ones(300) * (10 + rank)
    # two micro-steps of gradient accumulation (train.py:198-202): backward = "accumulate into gflat, then the hook"
    with dp.no_sync():
        RT.gflat += g1
        assert m._grad_hook.bucket_hooks(RT, None) is None
        m._grad_hook(RT)                                          # no collective inside no_sync
    RT.gflat += g2
    m._grad_hook(RT)
    expect = torch.arange(300, dtype=torch.float32) * 3 + 21.0    # sum over both ranks of (g1 + g2), each counted ONCE
    q.put((rank, torch.equal(RT.gflat, expect)))
    dist.destroy_process_group()


def test_no_sync_accumulation_gloo_world2():
    """the in-place all-reduce after EVERY micro-step would give world*G1 + G2; with no_sync() the accumulated sum is
    reduced once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 150)
    procs = [ctx.Process(target=_nosync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _bf16_wire_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ryolov4_amd import parallel
    parallel.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(7)
    grads = [torch.randn(5003, generator=g) * 10 ** torch.randint(-3, 3, (5003,), generator=g).float() for _ in range(world)]   # every rank knows all
    # (a) the wire-format reduce itself, odd length (padding), against its definition: bf16(sum_k fp32(bf16(g_k))) with fp32 accumulation
    mine = grads[rank].clone()
    parallel.allreduce_bf16_wire(mine)
    acc = grads[0].bfloat16().float()
    for k in range(1, world):
        acc = acc + grads[k].bfloat16().float()
    expect = acc.bfloat16().float()
    ok_def = torch.equal(mine, expect)
    exact = sum(g_.double() for g_ in grads)
    rel = float((mine.double() - exact).norm() / exact.norm())

    # (b) DataParallel(wire="bf16") through the overlapped reducer and through the end-of-backward pass: DP vs single-process parity
    class RT:
        flat = torch.zeros(5003)
        gflat = grads[rank].clone()
        _pslice = {i: (o, 10, None) for i, o in enumerate(range(0, 5003, 64))}

    class G:
        grad_writes = [(3, [4992]), (7, [2048, 1024]), (9, [0])]

        def grad_ready_points(self, bounds):
            from ryolov4_amd.engine.graph import Graph
            return Graph.grad_ready_points(self, bounds)

    class Model:
        _grad_hook = None

        def runtime(self):
            return RT

        def buffers(self):
            return []

    m = Model()
    dp = parallel.DataParallel(m, bucket_bytes=4096, overlap=True, wire="bf16")
    hooks = m._grad_hook.bucket_hooks(RT, G())
    for i in range(12):
        if i in hooks:
            hooks[i]()
    m._grad_hook(RT)
    ok_overlap = torch.equal(RT.gflat, expect)                    # bucket boundaries do not change the element-wise definition
    RT.gflat = grads[rank].clone()
    dp2 = parallel.DataParallel(m, bucket_bytes=4096, overlap=False, wire="bf16")
    m._grad_hook(RT)
    ok_serial = torch.equal(RT.gflat, expect)
    # the SGD step on the averaged gradient: replicas identical, and within bf16 rounding of the single-process big-batch step
    p_dp = torch.ones(5003) - 0.01 * (RT.gflat * dp2.grad_scale)
    p_single = torch.ones(5003) - 0.01 * (exact / world).float()
    q.put((rank, ok_def, ok_overlap, ok_serial, rel, float((p_dp - p_single).abs().max()), p_dp[:16].tolist()))
    dist.destroy_process_group()


def test_bf16_wire_gradient_buckets_gloo_world2():
    """SURVEY §8(e) / VERDICT r3 item 9: bf16 gradient buckets with fp32 accumulation on receive.  The result equals its definition bit for
    bit on every rank (so replicas stay identical), through the overlapped bucket schedule and the serial one, and the DP step matches the
    single-process step to bf16 rounding (2^-8 relative per element)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29200 + (os.getpid() % 150)
    procs = [ctx.Process(target=_bf16_wire_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for _, ok_def, ok_overlap, ok_serial, rel, dmax, _ in res:
        assert ok_def and ok_overlap and ok_serial
        assert rel < 2 ** -7 and dmax < 0.01 * 2 ** -6 * 1e3            # gradients up to 1e3: |dp - single| <= lr * |g| * 2^-7
    assert res[0][-1] == res[1][-1]                                     # replicas bit-identical
    from ryolov4_amd import parallel
    with pytest.raises(ValueError):
        parallel._Reducer(1024, wire="fp8")


# ---- round 5: 8-GPU readiness that needs no 8-GPU node and the unequal-shard hang -----------------
def test_affinity_plan_splits_numa_nodes_between_the_ranks_that_share_them():
    """MI355X chassis shape: 8 GPUs on 2 sockets.  Ranks 0-3 hang off node 0 (cores 0-63), ranks 4-7 off node 1 (64-127): every rank gets
    a quarter of ITS node, no two ranks share a core, and every rank computes the same plan on its own."""
    from ryolov4_amd import parallel
    topo = lambda dev: (list(range(0, 64)) if dev < 4 else list(range(64, 128)), 0 if dev < 4 else 1)
    plans = [parallel.plan_affinity(r, 8, allowed=range(128), gpu_cpus=topo) for r in range(8)]
    sets = [set(p["cpus"]) for p in plans]
    assert all(len(s) == 16 for s in sets)
    assert all(sets[a].isdisjoint(sets[b]) for a in range(8) for b in range(a + 1, 8))
    assert all(max(sets[r]) < 64 for r in range(4)) and all(min(sets[r]) >= 64 for r in range(4, 8))
    assert [p["numa_node"] for p in plans] == [0] * 4 + [1] * 4
    # two ranks on ONE device (bench.py --same-device plumbing runs): still distinct halves
    two = [parallel.plan_affinity(r, 2, device_of_rank=[0, 0], allowed=range(128), gpu_cpus=topo) for r in range(2)]
    assert set(two[0]["cpus"]).isdisjoint(two[1]["cpus"]) and len(two[0]["cpus"]) == 32
    # a cgroup that allows only part of the node: the plan stays inside it
    part = [parallel.plan_affinity(r, 8, allowed=range(0, 8), gpu_cpus=topo)["cpus"] for r in range(8)]
    assert sorted(sum(part, [])) == list(range(8))          # (node 1's cores are outside the cgroup: its ranks share what is allowed)
    # no PCI topology (this container): even split of what the process may use
    flat = [parallel.plan_affinity(r, 4, allowed=range(8), gpu_cpus=lambda d: None) for r in range(4)]
    assert [p["cpus"] for p in flat] == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert parallel._fmt_cpulist([0, 1, 2, 5, 7, 8]) == "0-2,5,7-8" and parallel._parse_cpulist("0-2,5,7-8\n") == [0, 1, 2, 5, 7, 8]


def test_wire_rule():
    from ryolov4_amd import parallel
    assert parallel.pick_wire(64, 8) == "fp32" and parallel.pick_wire(8, 8) == "bf16" and parallel.pick_wire(8, 1) == "fp32"
    assert parallel.pick_wire(8, 8, "fp32") == "fp32" and parallel.pick_wire(64, 8, "bf16") == "bf16"


def test_shards_are_equal_when_the_file_count_does_not_divide():
    """n = 129 files, world 2, batch 64 was 2 batches on rank 0 and 1 on rank 1 (an all-reduce without a peer).  Now both shards hold
    ceil(129 / 2) = 65 files (the short one wraps around), every file is in some shard, and the loaders agree on the batch count."""
    from ryolov4_amd.datasets.base_dataset import BaseDataset, DeviceLoader
    hyp = {"hsv_h": 0, "hsv_s": 0, "hsv_v": 0, "rotate": 0, "translate": 0, "scale": 0, "flipud": 0, "fliplr": 0, "mosaic": 0, "mixup": 0}
    for n, world, bs in ((129, 2, 64), (10, 4, 3), (7, 8, 2), (64, 8, 8)):
        lens, seen, nb = [], set(), []
        for rank in range(world):
            ds = BaseDataset(hyp, 64, True, False, False, device="cpu")          # augment=True: a TRAINING dataset
            ds.img_files = [f"img{i}" for i in range(n)]
            ds.label_files = [f"lab{i}" for i in range(n)]
            ld = DeviceLoader(ds, bs, shuffle=False, side_stream=False, rank=rank, world_size=world)
            lens.append(len(ds))
            nb.append(len(ld))
            seen |= set(ds.img_files)
            assert [f.replace("img", "lab") for f in ds.img_files] == ds.label_files
        assert len(set(lens)) == 1 and lens[0] == -(-n // world) and len(set(nb)) == 1, (n, world, lens, nb)
        assert seen == {f"img{i}" for i in range(n)}


def test_evaluation_shards_are_disjoint_and_exact():
    """a sharded EVALUATION set (augment=False, test.py:167-222) must hold every image exactly once over all ranks — a
    wrapped-around duplicate would count its detections and ground truth twice in the mAP statistics.  Training sets keep the padding."""
    from ryolov4_amd.datasets.base_dataset import BaseDataset, DeviceLoader
    hyp = {"hsv_h": 0, "hsv_s": 0, "hsv_v": 0, "rotate": 0, "translate": 0, "scale": 0, "flipud": 0, "fliplr": 0, "mosaic": 0, "mixup": 0}
    for n, world in ((129, 2), (10, 4), (7, 8), (64, 8)):
        files = []
        for rank in range(world):
            ds = BaseDataset(hyp, 64, False, False, False, device="cpu")         # augment=False: an evaluation dataset
            ds.img_files = [f"img{i}" for i in range(n)]
            ds.label_files = [f"lab{i}" for i in range(n)]
            DeviceLoader(ds, 4, shuffle=False, side_stream=False, rank=rank, world_size=world)
            assert [f.replace("img", "lab") for f in ds.img_files] == ds.label_files
            files += ds.img_files
        assert sorted(files) == sorted(f"img{i}" for i in range(n)), (n, world)          # every image once, none twice
    ds = BaseDataset(hyp, 64, False, False, False, device="cpu")
    ds.img_files, ds.label_files = [f"img{i}" for i in range(5)], [f"lab{i}" for i in range(5)]
    assert len(ds.shard(1, 2, pad=True)) == 3                                           # the explicit override still pads


def _uneven_worker(rank, world, port, q):
    """The loop of train.py over a DeviceLoader-shaped iteration with one all-reduce per batch: with unequal shards this deadlocks."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ryolov4_amd import parallel
    from ryolov4_amd.datasets.base_dataset import BaseDataset, DeviceLoader
    parallel.init_from_env(backend="gloo")
    hyp = {"hsv_h": 0, "hsv_s": 0, "hsv_v": 0, "rotate": 0, "translate": 0, "scale": 0, "flipud": 0, "fliplr": 0, "mosaic": 0, "mixup": 0}
    ds = BaseDataset(hyp, 64, True, False, False, device="cpu")          # a training dataset (augment=True): padded, equal shards
    ds.img_files = [f"img{i}" for i in range(129)]
    ds.label_files = list(ds.img_files)
    ld = DeviceLoader(ds, 64, shuffle=False, side_stream=False, rank=rank, world_size=world)
    n, sizes = len(ds), []
    for a in range(0, n, ld.batch_size):                                 # (DeviceLoader.__iter__'s slicing, without touching pixels)
        sizes.append(len(range(a, min(n, a + ld.batch_size))))
        g = torch.ones(4) * (rank + 1)
        dist.all_reduce(g)
    q.put((rank, sizes))
    dist.destroy_process_group()


def test_uneven_file_count_does_not_hang_the_allreduce_loop():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 40)
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res[0] == res[1] == [64, 1]


# ---- round 6: the side-stream partition shrinks by the RCCL channel count when world > 1 ----------------------------------
def test_partition_rule_shrinks_the_side_stream_by_the_channel_count(monkeypatch):
    from ryolov4_amd import parallel
    one = parallel.plan_partition(1)
    assert one["side_cus"] == 96 and one["main_cus"] == 160 and one["rccl_channels"] == 0
    for world in (2, 4, 8):
        p = parallel.plan_partition(world)
        assert p["rccl_channels"] == 8 and p["side_cus"] == 88 and p["main_cus"] == 160
        assert p["side_cus"] + p["main_cus"] + p["rccl_channels"] == 256                 # collectives never queue behind CU-exclusive workgroups
    assert parallel.plan_partition(8, channels=16)["side_cus"] == 80
    assert parallel.plan_partition(8, channels=4)["side_cus"] == 92
    assert parallel.plan_partition(8, channels=200)["rccl_channels"] == 32              # clamped
    assert parallel.plan_partition(8, channels=32)["side_cus"] == 64
    # applied to the environment the HIP library reads its grids from; explicit values win and are reported as such
    for k in ("RYOLO_W3_V8_BLOCKS", "RYOLO_WGRAD_8W_BLOCKS", "RYOLO_RCCL_CHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS"):
        monkeypatch.delenv(k, raising=False)
    got = parallel.apply_partition(8)
    assert os.environ["RYOLO_W3_V8_BLOCKS"] == os.environ["RYOLO_WGRAD_8W_BLOCKS"] == "88" and os.environ["RYOLO_RCCL_CHANNELS"] == "8"
    assert got["w3_v8_blocks"] == got["wgrad_8w_blocks"] == 88 and got["set_by"] == {"RYOLO_W3_V8_BLOCKS": "rule", "RYOLO_WGRAD_8W_BLOCKS": "rule"}
    assert parallel.rccl_env()["NCCL_MIN_NCHANNELS"] == parallel.rccl_env()["NCCL_MAX_NCHANNELS"] == "8"
    monkeypatch.setenv("RYOLO_W3_V8_BLOCKS", "64")
    monkeypatch.delenv("RYOLO_WGRAD_8W_BLOCKS")
    monkeypatch.setenv("RYOLO_RCCL_CHANNELS", "12")
    got = parallel.apply_partition(4)
    assert got["w3_v8_blocks"] == 64 and got["wgrad_8w_blocks"] == 84 and got["rccl_channels"] == 12
    assert got["set_by"] == {"RYOLO_W3_V8_BLOCKS": "environment", "RYOLO_WGRAD_8W_BLOCKS": "rule"}
    for k in ("RYOLO_W3_V8_BLOCKS", "RYOLO_WGRAD_8W_BLOCKS", "RYOLO_RCCL_CHANNELS"):
        monkeypatch.delenv(k, raising=False)
    got = parallel.apply_partition(1)                                                    # single GPU: nothing is written, the library defaults hold
    assert "RYOLO_W3_V8_BLOCKS" not in os.environ and got["side_cus"] == 96 and got["w3_v8_blocks"] == 96


def test_gpu_telemetry_never_raises():
    from ryolov4_amd import parallel
    t = parallel.gpu_telemetry(0, timeout=5.0)
    assert isinstance(t, dict) and (("error" in t) or ("power_w" in t or "sclk" in t))
