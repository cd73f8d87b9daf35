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
