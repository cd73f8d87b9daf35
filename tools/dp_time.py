"""Single-process timing of the full data-parallel step path (RCCL process group of one rank, bucket hooks forced on): main stream +
weight-gradient stream + the collective stream in one process, the shape every rank of an N-GPU run has.
usage: [RYOLO_WGRAD_STREAM=0|1] python tools/dp_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from ryolov4_amd import parallel
from ryolov4_amd.lib.loss import ComputeKFIoULoss
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, HYP, synth_batch

torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group(os.environ.get("BACKEND", "nccl"), rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda:0")
m = Yolo(16, CFG, "kfiou", "yolov7"); m.apply(bench.weights_init_normal); m.to(dev).train()
dp = parallel.DataParallel(m, force=True)
rt = m.runtime(); crit = ComputeKFIoULoss(m, HYP)
imgs, tg = synth_batch(64, 800, 16, False, seed=42); imgs, tg = imgs.to(dev), tg.to(dev)


def step():
    outs = dp(imgs, training=True); loss, _ = crit(outs, tg, sync_items=False); loss.backward()
    rt.sgd_step(0.01, 0.937, grad_scale=dp.grad_scale, zero_grad=True)


for _ in range(3):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8):
    step()
torch.cuda.synchronize()
print("wgrad_stream", rt.wgrad_stream, "ms/step", round((time.perf_counter() - t0) / 8 * 1e3, 2), "finite", bool(torch.isfinite(rt.flat).all()))
dist.destroy_process_group()
