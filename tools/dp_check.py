"""2-rank end-to-end check of the overlapped gradient all-reduce: two training steps with overlap on and off from the same
initial state must leave bit-identical parameters (the reduction itself is the same sum; only its schedule differs).
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_check.py
(gloo + one shared GPU is enough: the plumbing, bucket hooks and stream ordering are what is being checked)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from ryolov4_amd import parallel
from ryolov4_amd.lib.loss import ComputeKFIoULoss
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, HYP, synth_batch

torch.cuda.set_device(0)
backend = os.environ.get("BACKEND", "gloo")
if int(os.environ.get("WORLD_SIZE", "1")) == 1:
    # single rank: still create the process group so the collectives really go through the backend (RCCL with BACKEND=nccl)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend, rank=0, world_size=1)
    rank, world = 0, 1
else:
    rank, _, world = parallel.init_from_env(backend=backend)
dev = torch.device("cuda:0")
res = {}
for overlap in (True, False):
    torch.manual_seed(0)
    m = Yolo(16, CFG, "kfiou", "yolov7")
    for p in m.parameters():
        torch.nn.init.normal_(p, 0.0, 0.05)
    m.to(dev).train()
    dp = parallel.DataParallel(m, bucket_bytes=8 << 20, overlap=overlap, force=True)
    rt = m.runtime()
    crit = ComputeKFIoULoss(m, HYP)
    for step in range(2):
        imgs, tg = synth_batch(2, 256, 16, False, seed=100 + 10 * step + rank)
        outs = dp(imgs.to(dev), training=True)
        loss, _ = crit(outs, tg.to(dev), sync_items=False)
        loss.backward()
        rt.sgd_step(0.01, 0.937, grad_scale=dp.grad_scale, zero_grad=True)
    torch.cuda.synchronize()
    res[overlap] = rt.flat.clone()
    if overlap:
        print(f"rank {rank}: buckets {len(dp._reducer.bounds)} hooks at {sorted(dp._reducer.plans[next(iter(dp._reducer.plans))])[:6]}...", flush=True)
same = torch.equal(res[True], res[False])
other = res[True].clone()
dist.broadcast(other, src=0)
print(f"rank {rank}: overlap == serial: {same}; replicas identical: {torch.equal(other, res[True])}; finite: {bool(torch.isfinite(res[True]).all())}", flush=True)
dist.destroy_process_group()
