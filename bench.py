#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: one training step (forward + fused loss + backward [+ RCCL all-reduce] + fused
Nesterov-SGD step) of the DOTA yolov7 KFIoU configuration at 800x800 (BASELINE.json config C4; at N=1 the same per-GPU
workload), synthetic data of SURVEY.md §8(d), bf16 activations / fp32 accumulate / fp32 master weights.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...)

Prints ONE JSON line on rank 0: metric/value (whole-job img/s, inputs resident in HBM), `roofline` for the dominant conv
kernel (algorithmic FLOPs / HIP-event time of its launches inside the timed region, vs the dense bf16 MFMA peak), and
`cpu_baseline` (the torch-CPU oracle restatement timed on this host's cores on a bounded sample).
"""
import argparse
import json
import math
import os
import sys
import time

# Before the HIP runtime starts: the engine runs backward on two streams (weight gradients beside the data-gradient / BatchNorm chain)
# and RCCL adds its own; with the default of 4 hardware queues per process the extra stream gets multiplexed onto the main stream's
# queue and the overlap is lost (measured with the RCCL path active: 95.4 ms/step at 4 queues, 91.8 at 8).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E ~8 TB/s (≈6.3 achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0        # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md (2495 measured)
TRAIN_GFLOP_PER_IMG = {("yolov7", "kfiou", 800): 499.6, ("yolov7", "csl", 800): 505.1}      # BASELINE.md §2 (3 x forward)


class EventTimer:
    """Pairs of HIP events around individual kernel launches, recorded on the stream the kernels are launched on."""

    def __init__(self):
        self.pool, self.used, self.notes = [], 0, []

    def pair(self):
        if self.used + 2 > len(self.pool):
            self.pool.extend(torch.cuda.Event(enable_timing=True) for _ in range(1024))
        a, b = self.pool[self.used], self.pool[self.used + 1]
        self.used += 2
        return a, b

    def note(self, kind, flops, e0, e1, nbytes=0):
        self.notes.append((kind, flops, e0, e1, nbytes))

    def summary(self):
        agg = {}
        for kind, fl, e0, e1, nb in self.notes:
            d = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
            d[0] += e0.elapsed_time(e1) * 1e-3
            d[1] += fl
            d[2] += 1
            d[3] += nb
        return {k: dict(seconds=v[0], flops=v[1], launches=v[2], bytes=v[3]) for k, v in agg.items()}


def weights_init_normal(m):            # train.py:28-33
    if isinstance(m, torch.nn.Conv2d):
        torch.nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif isinstance(m, torch.nn.BatchNorm2d):
        torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
        torch.nn.init.constant_(m.bias.data, 0.0)


PMC_CLASSES = {
    "conv_gemm_kernel<128x128>": ("conv_gemm_kernel<128, 128, 2, 2, 1, 32>", "conv_gemm_kernel<128, 128, 2, 2, 1, 64>"),
    "conv_gemm_kernel<128x64>": ("conv_gemm_kernel<128, 64, 2, 2, 1, 32>", "conv_gemm_kernel<128, 64, 2, 2, 1, 64>"),
    "conv_gemm_kernel<256x32>": ("conv_gemm_kernel<256, 32, 4, 1, 1, 32>",),
    "conv3x3_patch_kernel<256x128>": ("conv3x3_patch_kernel<128, 2, 2>",),
    "conv3x3_patch_kernel<256x64>": ("conv3x3_patch_kernel<64, 4, 1>",),
    "conv_wgrad_kernel<128>": ("conv_wgrad_kernel<128>",),
    "conv_wgrad_kernel<64>": ("conv_wgrad_kernel<64>",),
    "conv3x3_wgrad_kernel<128x9x32>": ("conv3x3_wgrad_kernel<false>", "conv3x3_wgrad_kernel<true>"),
}


def pmc_traffic(args):
    """HBM bytes per launch of each timed kernel class from the committed rocprofv3 PMC profile of THIS workload
    (profiles/r01_pmc_step_traffic.json, made by tools/pmc_step.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2
    correction on FETCH_SIZE).  Counters cannot be read from inside the timed run; any other configuration reports null."""
    if (args.ver, args.mode, args.size, args.nc, args.batch) != ("yolov7", "kfiou", 800, 16, 64):
        return {}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_step_traffic.json")
    if not os.path.exists(path):
        return {}
    kern = json.load(open(path))["kernels"]
    res = {}
    for cls, names in PMC_CLASSES.items():
        rows = [kern[n] for n in names if n in kern]
        n = sum(r["launches"] for r in rows)
        if n:
            res[cls] = int(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n)
    return res


def pmc_step_bytes(args):
    """HBM bytes of ONE training step summed over every kernel of the committed PMC profile (NMS timing kernels excluded)."""
    if (args.ver, args.mode, args.size, args.nc, args.batch) != ("yolov7", "kfiou", 800, 16, 64):
        return 0
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_step_traffic.json")
    if not os.path.exists(path):
        return 0
    doc = json.load(open(path))
    steps = doc.get("steps_profiled", 3)
    return int(sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in doc["kernels"].items() if not k.startswith("nms_")) / steps)


def cpu_baseline(args, budget_s=25.0):
    """The oracle restatement (oracle/ref_model.py + ref_ops.py, pinned to the imported reference by the golden fixtures)
    doing the same training step on the host cores: fp32, SGD nesterov.  Bounded sample: batch 1, at most 2 steps."""
    from oracle import ref_model, ref_ops
    from ryolov4_amd.synth import CFG, HYP, synth_batch
    cores = min(os.cpu_count() or 1, 32)       # torch-CPU convs stop scaling (and oversubscribe) beyond ~32 threads at batch 1
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    net = ref_model.Yolo(args.nc, CFG, args.mode, args.ver)
    net.apply(weights_init_normal)
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.937, nesterov=True)
    imgs, tg = synth_batch(1, args.size, args.nc, args.mode == "csl", seed=42, per_image=64)
    times = []
    t_start = time.time()
    for _ in range(2):
        t0 = time.time()
        outs = net(imgs, True)
        loss, _ = ref_ops.compute_loss(outs, tg, net.anchors, args.nc, args.mode, HYP)
        loss.backward()
        opt.step()
        opt.zero_grad()
        times.append(time.time() - t0)
        if time.time() - t_start > budget_s:
            break
    return {"value": round(1.0 / min(times), 4), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} training step(s) of batch 1 at {args.size}x{args.size} ({args.ver} {args.mode} nc={args.nc}), fp32 torch-CPU oracle, best step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (weak scaling); 64 = the reference nominal batch (train.py:150)")
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--ver", default="yolov7")
    ap.add_argument("--mode", default="kfiou")
    ap.add_argument("--nc", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); gloo only for plumbing tests")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses cuda:0")
    args = ap.parse_args()

    import __graft_entry__ as ge
    from ryolov4_amd import parallel
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    local = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)                       # before the communicator is created: one process per GPU
    dev = torch.device("cuda", local)
    rank, _, world = parallel.init_from_env(backend=args.backend)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    if rank == 0:
        ge.build()                                     # one rank checks/builds the shared library, the others wait
    if world > 1:
        dist.barrier()

    from ryolov4_amd.lib.general import _nms_sorted_batched
    from ryolov4_amd.lib.loss import ComputeCSLLoss, ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, synth_batch, synth_nms_boxes

    torch.manual_seed(42)              # train.py:20-25
    model = Yolo(args.nc, CFG, args.mode, args.ver)
    model.apply(weights_init_normal)
    model.to(dev).train()
    dp = parallel.DataParallel(model)
    rt = model.runtime()
    crit = (ComputeCSLLoss if args.mode == "csl" else ComputeKFIoULoss)(model, HYP)
    imgs, targets = synth_batch(args.batch, args.size, args.nc, args.mode == "csl", seed=42 + rank, per_image=64)
    imgs, targets = imgs.to(dev), targets.to(dev)          # inputs resident in HBM before the timed region
    lr = 0.01

    def step():
        outs = model(imgs, training=True)
        loss, _ = crit(outs, targets, sync_items=False)
        loss.backward()                                   # engine backward + (N>1) RCCL all-reduce of the flat gradient buffer
        rt.sgd_step(lr, 0.937, grad_scale=dp.grad_scale, zero_grad=True)

    def read_loss():                                          # one extra, untimed step whose loss is read back (host sync)
        outs = model(imgs, training=True)
        loss, items = crit(outs, targets)
        loss.backward()
        rt.sgd_step(lr, 0.937, grad_scale=dp.grad_scale, zero_grad=True)
        return float(items["total_loss"])

    loss_first = read_loss()
    for _ in range(args.warmup):
        step()
    g = rt.graph(args.batch, args.size, args.size, True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: EXACTLY K steps, nothing else on the stream ---------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # ---- the same K steps again with a HIP-event pair around every conv launch (per-kernel durations for the rooflines).  Kept out
    # of the timed region: ~600 event records per step cost ~3 % (measured 688 vs 710 img/s), the kernels themselves run unchanged
    # (the durations agree with the rocprofv3 summary of the same command, profiles/).
    timer, dt_inst = None, None
    if not args.no_kernel_timing:
        timer = EventTimer()
        g.timer = timer
        g.serial = True                                   # every kernel alone on the GPU: the second stream is folded back into
        barrier()                                         # the main stream for this pass (overlapped kernels stretch each other)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_inst = time.perf_counter() - t1
        g.timer = None
        g.serial = False
    loss_last = read_loss()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return

    ms_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt
    out = {
        "metric": "training img/s @800x800 bf16", "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"C4: DOTA {args.ver} {args.mode} nc={args.nc} {args.size}x{args.size}, batch {args.batch}/GPU, fwd+loss+bwd+SGD-nesterov step"
                               + (", RCCL all-reduce of 37.9M fp32 grads" if world > 1 else ""),
                   "global_batch": args.batch * world, "parallelism": f"dp{world}", "targets_per_image": 64,
                   "weights": "random init N(0,0.02) (train.py:28-33)"},
    }
    # the synthetic batch is fixed, so real training shows as a falling loss; NaN / inf would void the run (rank 0's view)
    out["train_loss"] = {"first_step": round(loss_first, 4), "after_timed_steps": round(loss_last, 4),
                         "finite": bool(math.isfinite(loss_first) and math.isfinite(loss_last) and bool(torch.isfinite(rt.flat).all()))}
    gf = TRAIN_GFLOP_PER_IMG.get((args.ver, args.mode, args.size))
    if gf:
        out["config"]["train_gflop_per_img"] = gf
        out["mfma_roofline_frac_whole_step"] = round(value * gf * 1e9 / (world * MFMA_BF16_PEAK_TFLOPS * 1e12), 4)
    step_bytes = pmc_step_bytes(args)
    if step_bytes and world == 1:
        # whole-step HBM view: PMC bytes of every kernel of one step (committed profile of THIS workload) over the measured step time
        out["hbm_whole_step"] = {"traffic_gb_per_step": round(step_bytes / 1e9, 1), "achieved_gbs": round(step_bytes / (dt / args.steps) / 1e9, 1),
                                 "peak_gbs": HBM_PEAK_GBS, "frac": round(step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
    if timer is not None:
        summ = timer.summary()
        pmc = pmc_traffic(args)

        def roof(k, v):
            # the roofline that binds the class: algorithmic FLOPs against the dense bf16 MFMA peak, or algorithmic bytes (every
            # operand element once) against the HBM peak — whichever fraction is larger (the generic GEMM class is mostly 1x1
            # layers with 8-32 K steps: memory streams)
            tf = v["flops"] / v["seconds"] / 1e12
            gb = v["bytes"] / v["seconds"] / 1e9
            fm, fh = tf / MFMA_BF16_PEAK_TFLOPS, gb / HBM_PEAK_GBS
            head = ({"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fh, 4)} if fh > fm else
                    {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fm, 4)})
            return {"kernel": k, **head, "frac_mfma": round(fm, 4), "frac_hbm": round(fh, 4),
                    "algorithmic_bytes_per_launch": int(v["bytes"] / max(1, v["launches"])), "traffic": pmc.get(k),
                    "launches_per_step": v["launches"] // args.steps,
                    "avg_launch_us": round(v["seconds"] / v["launches"] * 1e6, 2),
                    "share_of_step": round(v["seconds"] / dt_inst, 4)}
        # dominant kernel class by time (HIP events on the launch stream)
        out["roofline"] = roof(*max(summ.items(), key=lambda kv: kv[1]["seconds"]))
        # BASELINE.json north_star quotes the MFMA fraction of the 3x3 convs separately: the halo-patch kernel (fwd + dgrad)
        k33 = "conv3x3_patch_kernel<256x128>"
        if k33 in summ:
            out["roofline_3x3"] = roof(k33, summ[k33])
        out["kernel_timing"] = {"how": "HIP events around every conv launch, on the launch stream, in a second pass of the same K steps right after the timed region, with the two backward streams serialized (a kernel is timed alone on the GPU; rocprofv3 summary of the same: RYOLO_WGRAD_STREAM=0)",
                                "ms_per_step_instrumented": round(dt_inst / args.steps * 1e3, 3)}
        out["kernels"] = {kk: {"tflops": round(vv["flops"] / vv["seconds"] / 1e12, 2), "ms_per_step": round(vv["seconds"] / args.steps * 1e3, 3),
                               "launches_per_step": vv["launches"] // args.steps} for kk, vv in summ.items()}
    # secondary metric of BASELINE.json: rotated-NMS latency at 10k boxes (device time, median of 30; clustered set, thr 0.65)
    b, _ = synth_nms_boxes(10000, "C", seed=0)
    tb = torch.from_numpy(b).to(dev).unsqueeze(0).contiguous()
    ts = []
    for i in range(35):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _nms_sorted_batched(tb, None, 0.65, True, None)
        e1.record()
        torch.cuda.synchronize()
        if i >= 5:
            ts.append(e0.elapsed_time(e1))
    out["nms_ms_10k_boxes"] = round(sorted(ts)[len(ts) // 2], 4)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
