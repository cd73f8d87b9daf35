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
N_CUS = 256                                # MI355X: 8 XCDs x 32 CUs
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

    def note(self, kind, flops, e0, e1, nbytes=0, sub=None, cus=None, fam=None):
        """cus: CUs the launch holds when it runs alone (engine/graph.py _describe; None = the whole chip); fam: layer family ("3x3s1" /
        "3x3s2": forward, data-gradient and weight-gradient launches of the 3x3 convolutions, whichever kernel serves them)."""
        self.notes.append((kind, flops, e0, e1, nbytes, sub, cus, fam))

    def summary(self, by_sub=False, by_fam=False):
        """by_sub: only the launches that carry a sub-class label (engine/graph.py _describe), keyed (class, sub); by_fam: only the launches
        of a layer family, keyed (family, class)."""
        agg = {}
        for kind, fl, e0, e1, nb, sub, cus, fam in self.notes:
            if (by_sub and sub is None) or (by_fam and fam is None):
                continue
            d = agg.setdefault((fam, kind) if by_fam else (kind, sub) if by_sub else kind, [0.0, 0.0, 0, 0.0, 0.0])
            t = e0.elapsed_time(e1) * 1e-3
            d[0] += t
            d[1] += fl
            d[2] += 1
            d[3] += nb
            d[4] += t * (N_CUS if cus is None else min(cus, N_CUS)) / N_CUS      # chip-seconds: the share of the chip the launch held
        return {k: dict(seconds=v[0], flops=v[1], launches=v[2], bytes=v[3], chip_seconds=v[4]) for k, v in agg.items()}


def weights_init_normal(m):            # train.py:28-33
    if isinstance(m, torch.nn.Conv2d):
        torch.nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif isinstance(m, torch.nn.BatchNorm2d):
        torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
        torch.nn.init.constant_(m.bias.data, 0.0)


def _gemm_re(bm, bn, t1="(true|false)"):
    """conv_gemm_kernel<BM, BN, WM, WN, PIPE, KB, EP, ID, T1, NST> as rocprofv3 prints it (every template argument, defaults included)."""
    return rf"conv_gemm_kernel<{bm}, {bn}, \d+, \d+, \d+, \d+, \d+, (true|false), {t1}, \d+>"


PMC_CLASSES = {        # bench class -> regex over the kernel names of the rocprofv3 / PMC files: the class follows the tile and the 1x1 instantiation (T1);
                       # the other template arguments (stage width, epilogue form, identity grid, ring depth) vary inside it
    "conv_gemm_kernel<128x128,1x1>": _gemm_re(128, 128, "true"),
    "conv_gemm_kernel<128x128>": _gemm_re(128, 128, "false"),
    "conv_gemm_kernel<128x64,1x1>": _gemm_re(128, 64, "true"),
    "conv_gemm_kernel<128x64>": _gemm_re(128, 64, "false"),
    "conv_gemm_kernel<256x64>": _gemm_re(256, 64),
    "conv_gemm_kernel<256x32>": _gemm_re(256, 32),
    "gemm256_kernel<256x256>": r"gemm256_kernel<256, ",                                # 8-wave 256-wide pointwise GEMM for long reductions (gemm256.hip)
    "gemm256_kernel<256x128>": r"gemm256_kernel<128, ",
    "gemm1x1_ws_kernel": r"gemm1x1_ws_kernel<",                                       # weight-stationary persistent 1x1 (gemm1x1.hip, RYOLO_GEMM_WS)
    "conv3x3_patch_kernel<256x128>": r"conv3x3_patch_kernel<128, 2, 2[,>]",       # (+ epilogue / BatchNorm-fold template arguments since r04)
    "conv3x3_patch_kernel<256x64>": r"conv3x3_patch_kernel<64, 4, 1[,>]",
    "conv3x3s2_c32_dgrad_kernel": r"conv3x3s2_c32_dgrad_kernel",                      # ... and its space-to-depth data gradient
    "conv3x3s2_c32_kernel": r"conv3x3s2_c32_kernel<",                                 # streaming 3x3 stride-2 forward of the 32-channel layer (conv3x3s2_c32.hip, r06)
    "conv3x3_ws64_kernel": r"conv3x3_ws64_kernel<",                                   # persistent weight-stationary 64 -> <= 64 channel 3x3 (conv3x3_ws.hip)
    "conv_wgrad_kernel<128>": r"conv_wgrad_kernel<128, |wgrad1x1_dma_kernel|wgrad_taps_dma_kernel",   # (register-staged + the LDS-DMA pointwise / tapped forms)
    "conv_wgrad_kernel<64>": r"conv_wgrad_kernel<64, ",
    "wgrad1x1_8w_kernel<256x256>": r"wgrad1x1_8w_kernel",
    "conv3x3_wgrad_kernel<128x9x32>": r"conv3x3_wgrad(64|8)?_kernel<",
}


PMC_PROFILE = "profiles/r06_pmc_step_traffic.json"        # regenerated by tools/pmc_step.sh whenever a kernel of the step changes
MFMA_BUSY_PROFILE = "profiles/r06_pmc_step_mfma_busy.json"   # whole-step SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES per kernel class (tools/pmc_step.sh)
INFER_GFLOP_PER_IMG = {("yolov7", "kfiou", 800): 166.53, ("yolov7", "kfiou", 1024): 272.84}


def source_sha256():
    """Hash of everything that decides which kernels a step launches and what they do (csrc, include, engine, model): the PMC profile
    records it (tools/pmc_step.sh), and `traffic` is reported only while it still matches — a stale profile reports null, never old bytes."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "r-yolov4_amd")
    files = (glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) +
             glob.glob(os.path.join(pkg, "engine", "*.py")) + glob.glob(os.path.join(pkg, "model", "*.py")))
    for f in sorted(files):
        h.update(open(f, "rb").read())
    return h.hexdigest()


_PMC_STATE = {}


def _pmc_doc(args):
    if (args.ver, args.mode, args.size, args.nc, args.batch) != ("yolov7", "kfiou", 800, 16, 64):
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_PROFILE)
    if not os.path.exists(path):
        return None
    doc = json.load(open(path))
    if "stale" not in _PMC_STATE:
        _PMC_STATE["stale"] = doc.get("source_sha256") != source_sha256()
    return None if _PMC_STATE["stale"] else doc


def pmc_traffic(args):
    """HBM bytes per launch of each timed kernel class from the committed rocprofv3 PMC profile of THIS workload
    (PMC_PROFILE, made by tools/pmc_step.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2
    correction on FETCH_SIZE).  Counters cannot be read from inside the timed run; any other configuration reports null."""
    doc = _pmc_doc(args)
    if doc is None:
        return {}
    kern = doc["kernels"]
    res = {}
    import re
    for cls, pat in PMC_CLASSES.items():
        rows = [v for k, v in kern.items() if re.match(pat, k)]
        n = sum(r["launches"] for r in rows)
        if n:
            res[cls] = int(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n)
    return res


def pmc_step_bytes(args):
    """HBM bytes of ONE training step summed over every kernel of the committed PMC profile (NMS timing kernels excluded)."""
    doc = _pmc_doc(args)
    if doc is None:
        return 0
    steps = doc.get("steps_profiled", 3)
    skip = ("nms_", "bitonic_", "compose_kernel", "topk_")         # the NMS timing block of this script, not part of a training step
    return int(sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in doc["kernels"].items() if not k.startswith(skip)) / steps)


def _cpu_baseline_worker(ver, mode, nc, size, threads, budget_s):
    """Child process: the oracle restatement's training step on the host cores; prints one JSON line."""
    from oracle import ref_model, ref_ops
    from ryolov4_amd.synth import CFG, HYP, synth_batch
    if hasattr(os, "sched_setaffinity"):                       # the parent may be pinned to its GPU's cores (parallel.bind_rank): the baseline is not
        try:
            os.sched_setaffinity(0, range(os.cpu_count() or 1))
        except OSError:
            pass
    torch.set_num_threads(threads)
    torch.manual_seed(42)
    net = ref_model.Yolo(nc, CFG, mode, ver)
    net.apply(weights_init_normal)
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.937, nesterov=True)
    CB = 2
    imgs, tg = synth_batch(CB, size, nc, mode == "csl", seed=42, per_image=64)
    times = []
    t_start = time.time()
    for it in range(4):
        t0 = time.time()
        outs = net(imgs, True)
        loss, _ = ref_ops.compute_loss(outs, tg, net.anchors, nc, mode, HYP)
        loss.backward()
        opt.step()
        opt.zero_grad()
        if it > 0:
            times.append(time.time() - t0)
        if time.time() - t_start > budget_s and times:
            break
    print("CPUBASE " + json.dumps({"batch": CB, "times": times}), flush=True)


def cpu_baseline(args, budget_s=20.0):
    """The oracle restatement (oracle/ref_model.py + ref_ops.py, pinned to the imported reference by the golden fixtures) doing the
    same training step on the host cores (BASELINE.md §3 / SURVEY §8(d)): fp32, batch 2, SGD nesterov; one untimed step, then steps until
    ~budget_s is spent, best step reported.  BOUNDED: a child process with a hard time limit.  Threads: min(32, os.cpu_count()) — SURVEY
    §8(d) says torch.set_num_threads(os.cpu_count()), and rounds 2-4 tried exactly that first on every run: on the 256-thread GPU hosts
    torch-CPU's convolution backward is oversubscribed there (one batch-2 step ~300 s = 0.007 img/s in r02; no timed step within 50 s in
    any run of r03 / r04; 128 threads: 0.31 img/s) and the attempt only cost every bench run 50 s.  32 threads is where torch-CPU scales to on
    these hosts (1.0-1.4 img/s); `cores` / `sample` say what ran."""
    import subprocess
    ncpu = os.cpu_count() or 1
    threads = min(32, ncpu)
    limit = 3 * budget_s + 15
    code = ("import sys; sys.path.insert(0, %r); import bench; bench._cpu_baseline_worker(%r, %r, %d, %d, %d, %f)"
            % (ROOT, args.ver, args.mode, args.nc, args.size, threads, budget_s))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    note = f"torch.set_num_threads({threads}) did not finish a warm-up + a timed step within {int(limit)} s"
    try:
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=limit)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("CPUBASE ")]
        if r.returncode == 0 and line:
            d = json.loads(line[-1][len("CPUBASE "):])
            value = d["batch"] / min(d["times"])
            return {"value": round(value, 4), "unit": "img/s", "cores": threads, "kind": "port", "host_cpu_count": ncpu,
                    "sample": f"best of {len(d['times'])} timed training step(s) (after 1 untimed) of batch {d['batch']} at {args.size}x{args.size} "
                              f"({args.ver} {args.mode} nc={args.nc}), fp32 torch-CPU oracle, torch.set_num_threads({threads}) of {ncpu} host threads "
                              "(all-threads runs never finished a step on these hosts in rounds 2-4: docstring)"}
    except subprocess.TimeoutExpired:
        pass
    return {"value": None, "unit": "img/s", "cores": 0, "kind": "port", "host_cpu_count": ncpu, "sample": note}


def nms_block(dev):
    """Secondary metric of BASELINE.json: rotated-NMS latency at 10 k boxes: both synthetic sets (U uniform, C
    clustered), both thresholds (0.65 test.py:270, 0.2 detect.py:91), device time (HIP events, median of 30):
      mask_reduce_ms   prep + mask + on-device greedy reduce on PRE-SORTED boxes (what detectron2's kernel covers after its sort)
      end_to_end_ms    nms_rotated(boxes, scores, thr) as the reference calls it (lib/general.py:177): score sort + gather + the above
                       + the keep count read back."""
    from ryolov4_amd.lib.general import _nms_sorted_batched, nms_rotated
    from ryolov4_amd.synth import synth_nms_boxes

    def med(fn, n=30, skip=5):
        ts = []
        for i in range(n + skip):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            if i >= skip:
                ts.append(e0.elapsed_time(e1))
        return round(sorted(ts)[len(ts) // 2], 4)

    res = {"n_boxes": 10000, "mask_reduce_ms": {}, "end_to_end_ms": {}}
    for kind in ("U", "C"):
        b, sc = synth_nms_boxes(10000, kind, seed=0)
        tb = torch.from_numpy(b).to(dev).unsqueeze(0).contiguous()
        # the same boxes in a random order with their scores: what nms_rotated receives
        perm = torch.randperm(10000, generator=torch.Generator().manual_seed(1))
        ub, us = torch.from_numpy(b)[perm].to(dev).contiguous(), torch.from_numpy(sc)[perm].to(dev).contiguous()
        for thr in (0.65, 0.2):
            res["mask_reduce_ms"][f"{kind}_{thr}"] = med(lambda: _nms_sorted_batched(tb, None, thr, True, None))
            res["end_to_end_ms"][f"{kind}_{thr}"] = med(lambda: nms_rotated(ub, us, thr))
    return res


def infer_block(args, dev):
    """Inference in front of the driver (BASELINE config C5 and the eval branch of test.py:188-191 / detect.py:57-61).
      infer_c5        yolov7 kfiou nc=16, 1024 x 1024, 8 images: the eval forward (folded BatchNorm + activation epilogues, RepConv
                      re-parameterised) + YoloLayer decode + score filter + radix-select top-K + rotated NMS (lib/general.py:136-183) captured in ONE
                      hipGraph (conf 0.001 / iou 0.65: test.py:270-271); random-init weights put EVERY row past the confidence filter, so the
                      top-K and the NMS run at their worst-case sizes (max_nms candidates per image);
      infer_800_b64   the same network's eager batch-64 forward at 800 x 800 (the training benchmark's shape).
    Device time by HIP events over n replays; forward-only MFMA fraction from BASELINE.md section 2's GFLOP per image."""
    import gc
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, synth_batch
    torch.manual_seed(42)
    net = Yolo(args.nc, CFG, "kfiou", "yolov7")
    net.apply(weights_init_normal)
    net.to(dev).eval()

    def timed(fn, n, skip=3):
        for _ in range(skip):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    out = {}
    t_wall = time.perf_counter()
    # ---- C5: 1024^2 x 8, everything in one captured graph (one live graph at a time: DESIGN.md section 6, "Inference") ---------------------
    B, SZ, CONF, IOU = 8, 1024, 0.001, 0.65
    imgs, _ = synth_batch(B, SZ, args.nc, False, seed=42)
    imgs = imgs.to(dev)
    cap = net.capture_inference(B, SZ)
    t_f = timed(lambda: cap(imgs), 20)
    del cap
    gc.collect()
    torch.cuda.synchronize()
    capp = net.capture_inference(B, SZ, post=(CONF, IOU))
    t_p = timed(lambda: capp(imgs), 20)
    _, _, dets, num = capp(imgs)
    plan = capp.post_plan
    gf = INFER_GFLOP_PER_IMG[("yolov7", "kfiou", 1024)]
    out["infer_c5"] = {"workload": f"C5: yolov7 kfiou nc={args.nc} {SZ}x{SZ}, batch {B}, hipGraph-captured eval forward + decode + top-K + rotated NMS "
                                   f"(conf {CONF}, iou {IOU}); random-init weights: every row passes the confidence filter (worst-case post_process)",
                       "ms_per_batch": round(t_p, 3), "img_s": round(B / t_p * 1e3, 1), "forward_decode_ms": round(t_f, 3), "forward_img_s": round(B / t_f * 1e3, 1),
                       "post_process_ms": round(t_p - t_f, 3), "rows_per_image": int(plan.M), "nms_input_cap": int(plan.K),
                       "detections_per_image": num.cpu().tolist(), "gflop_per_img_forward": gf,
                       "frac_mfma_forward": round(B / t_f * 1e3 * gf * 1e9 / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4)}
    del capp, dets, num, plan, imgs
    gc.collect()
    torch.cuda.synchronize()
    # ---- the training benchmark's shape: batch 64 at 800^2, eager forward (+ eager post_process) --------------------------------------------
    from ryolov4_amd.lib.general import post_process
    B, SZ = 64, 800
    imgs, _ = synth_batch(B, SZ, args.nc, False, seed=42)
    imgs = imgs.to(dev)

    def fwd():
        with torch.no_grad():
            return net(imgs, training=False)

    def full():
        with torch.no_grad():
            _, inf = net(imgs, training=False)
            return post_process(inf, CONF, IOU)
    t_f = timed(fwd, 8, skip=2)
    t_a = timed(full, 4, skip=1)
    gf = INFER_GFLOP_PER_IMG[("yolov7", "kfiou", 800)]
    out["infer_800_b64"] = {"workload": f"yolov7 kfiou nc={args.nc} {SZ}x{SZ}, batch {B}, eager eval forward (+ decode); forward_post = + post_process at conf {CONF} / iou {IOU}",
                            "forward_ms": round(t_f, 3), "forward_img_s": round(B / t_f * 1e3, 1), "forward_post_ms": round(t_a, 3),
                            "forward_post_img_s": round(B / t_a * 1e3, 1), "post_process_ms": round(t_a - t_f, 3), "gflop_per_img_forward": gf,
                            "frac_mfma_forward": round(B / t_f * 1e3 * gf * 1e9 / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4)}
    out["wall_s"] = round(time.perf_counter() - t_wall, 2)
    del net, imgs
    gc.collect()
    torch.cuda.empty_cache()
    return out


def infer_child(args, limit=300):
    """The inference blocks in a process of their own, while this one sits idle (its streams synchronized): a captured hipGraph replays
    measurably slower beside another runtime's streams in the same process (the training runtime's two side streams + the loader's stream + the
    inference runtime's own: past the process's hardware queues — first r06 run: C5 9.2 ms per batch in-process, 6.8-7.0 ms alone).  Never
    loses the training line: errors come back as {"infer_c5": {"error": ...}}."""
    import subprocess
    torch.cuda.synchronize()
    cmd = [sys.executable, os.path.abspath(__file__), "--infer-only", "--nc", str(args.nc), "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=limit)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("INFERBLOCK ")]
        if r.returncode == 0 and line:
            d = json.loads(line[-1][len("INFERBLOCK "):])
            d["infer_process"] = "child process of bench.py (python bench.py --infer-only), run after the training blocks with this process idle"
            return d
        return {"infer_c5": {"error": (r.stderr or r.stdout)[-300:]}}
    except subprocess.TimeoutExpired:
        return {"infer_c5": {"error": f"inference child did not finish within {limit} s"}}


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
    ap.add_argument("--no-b8", action="store_true", help="skip the second regime (8 images per GPU)")
    ap.add_argument("--no-loader", action="store_true", help="skip the loader-in-the-loop block (the same step fed by DeviceLoader)")
    ap.add_argument("--no-infer", action="store_true", help="skip the inference blocks (infer_c5: captured 1024^2 x 8 forward + NMS; infer_800_b64)")
    ap.add_argument("--infer-only", action="store_true", help="internal: run only the inference blocks and print them (the child process of the default run)")
    ap.add_argument("--wire", default="auto", choices=["auto", "fp32", "bf16"],
                    help="gradient bucket format on xGMI (bf16: fp32 accumulate on receive); auto = parallel.pick_wire: bf16 for <= 16 images per GPU")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); gloo only for plumbing tests")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses cuda:0")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one process per GPU, RCCL rendezvous on 127.0.0.1) — a
        # single-rank run must never be reported under --gpus N
        if not args.same_device and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible (launch with --same-device for plumbing tests)")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    import __graft_entry__ as ge
    from ryolov4_amd import parallel
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    if args.infer_only:
        torch.cuda.set_device(0)
        print("INFERBLOCK " + json.dumps(infer_block(args, torch.device("cuda", 0))), flush=True)
        return
    local = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)                       # before the communicator is created: one process per GPU
    dev = torch.device("cuda", local)
    rank, _, world = parallel.init_from_env(backend=args.backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    # every rank on the cores of ITS GPU's NUMA node (one node: LOCAL_WORLD_SIZE ranks share the sockets); the loader's decode pool is sized to it
    n_local = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    aff = parallel.bind_rank(int(os.environ.get("LOCAL_RANK", "0")), n_local, device_of_rank=[0] * n_local if args.same_device else None)
    if rank == 0:
        ge.build()                                     # one rank checks/builds the shared library, the others wait
    if world > 1:
        dist.barrier()

    from ryolov4_amd.lib.loss import ComputeCSLLoss, ComputeKFIoULoss
    from ryolov4_amd.model.yolo import Yolo
    from ryolov4_amd.synth import CFG, HYP, synth_batch

    torch.manual_seed(42)              # train.py:20-25
    model = Yolo(args.nc, CFG, args.mode, args.ver)
    model.apply(weights_init_normal)
    model.to(dev).train()
    dp = parallel.DataParallel(model, wire=parallel.pick_wire(args.batch, world, args.wire))
    dp._reducer.timing = world > 1
    rt = model.runtime()
    crit = (ComputeCSLLoss if args.mode == "csl" else ComputeKFIoULoss)(model, HYP)
    lr = 0.01

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(batch, steps, warmup, timing):
        """One workload: `warmup` untimed steps, EXACTLY `steps` timed steps between barrier + synchronize, then (timing) the same
        steps again with a HIP-event pair around every conv launch for the rooflines."""
        imgs, targets = synth_batch(batch, args.size, args.nc, args.mode == "csl", seed=42 + rank, per_image=64)
        imgs, targets = imgs.to(dev), targets.to(dev)          # inputs resident in HBM before the timed region
        dp.set_wire(parallel.pick_wire(batch, world, args.wire))

        def step():
            outs = model(imgs, training=True)
            loss, _ = crit(outs, targets, sync_items=False)
            loss.backward()                                   # engine backward + (N>1) RCCL all-reduce of the flat gradient buffer
            rt.sgd_step(lr, 0.937, grad_scale=dp.grad_scale, zero_grad=True)

        def read_loss():                                      # one extra, untimed step whose loss is read back (host sync)
            outs = model(imgs, training=True)
            loss, items = crit(outs, targets)
            loss.backward()
            rt.sgd_step(lr, 0.937, grad_scale=dp.grad_scale, zero_grad=True)
            return float(items["total_loss"])

        loss_first = read_loss()
        for _ in range(warmup):
            step()
        g = rt.graph(batch, args.size, args.size, True)
        # ---- the timed region: EXACTLY K steps, nothing else on the stream --------------------------------------------------------
        barrier()
        dp._reducer.collective_ms()                           # (drop the warm-up steps' stamps)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt_rank = time.perf_counter() - t0                    # this rank's own K steps (before it waits for the others)
        barrier()
        dt = time.perf_counter() - t0
        telemetry = None
        if world > 1:
            # clocks / power of this rank's GPU UNDER LOAD, outside the timed region: a few more steps are enqueued (asynchronously) and rocm-smi is
            # read while they run — eight ~1.3 kW GPUs in one chassis is the first thing a SCALE run tests
            for _ in range(max(4, min(200, int(1.2 / (dt / steps)) + 1))):      # ~1.2 s of queued work: rocm-smi answers in 0.3-0.8 s
                step()
            telemetry = parallel.gpu_telemetry(local)
            barrier()
        coll_ms = dp._reducer.collective_ms() / steps if world > 1 else None
        # ---- the same K steps again with a HIP-event pair around every conv launch (per-kernel durations for the rooflines).  Kept
        # out of the timed region: ~600 event records per step cost ~3 %, the kernels themselves run unchanged (the durations agree
        # with the rocprofv3 summary of the same command, profiles/).
        timer, dt_inst = None, None
        if timing:
            timer = EventTimer()
            g.timer = timer
            g.serial = True                                   # every kernel alone on the GPU: the second stream is folded back into
            barrier()                                         # the main stream for this pass (overlapped kernels stretch each other)
            t1 = time.perf_counter()
            for _ in range(steps):
                step()
            barrier()
            dt_inst = time.perf_counter() - t1
            g.timer = None
            g.serial = False
        loss_last = read_loss()
        per_rank = None
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            mine = torch.tensor([dt_rank / steps * 1e3, coll_ms], dtype=torch.float64, device=dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank = {"ms_per_step": [round(float(e[0]), 3) for e in every], "allreduce_ms_per_step": [round(float(e[1]), 3) for e in every],
                        "wire": dp.wire}
        return dict(dt=dt, dt_inst=dt_inst, timer=timer, loss_first=loss_first, loss_last=loss_last, batch=batch, steps=steps, per_rank=per_rank,
                    telemetry=telemetry)

    def loader_fed(batch, steps, warmup, budget_frac=None, npool=256):
        """the SAME training step fed by the device-side loader instead of one resident synthetic batch.  A synthetic
        DOTA-like split (256 images of 1024x1024 uint8 BGR with 40 polygons each; the reference's augmentation hyper-parameters of
        data/hyp.yaml: mosaic 1.0, mixup 0.15, hsv, rotate / scale / translate, flips) is served by lib.load's DeviceLoader: host planning
        + ~10 launches per batch on the loader's own stream while the previous step runs.  `budget_frac`: pool byte budget as a fraction
        of the decoded split (None = everything stays resident after its first use; < 1 = slabs are recycled LRU and images re-uploaded).
        Image DECODE is an in-memory array lookup here (no PNG files on the GPU box): host decode time is not part of this number."""
        import random as _r
        import numpy as np
        from ryolov4_amd.datasets.base_dataset import BaseDataset, DeviceLoader
        hyp = {"hsv_h": 0.015, "hsv_s": 0.7, "hsv_v": 0.4, "rotate": 45, "translate": 0.1, "scale": 0.5, "flipud": 0.5, "fliplr": 0.5,
               "mosaic": 1.0, "mixup": 0.15}
        rs = np.random.RandomState(rank)
        side, nobj = 1024, 40
        base = rs.randint(0, 256, size=(side + 64, side + 64, 3)).astype(np.uint8)
        images = [np.ascontiguousarray(base[o:o + side, o:o + side]) for o in rs.randint(0, 64, size=npool)]
        polys, labels = [], []
        for _ in range(npool):
            c = rs.rand(nobj, 2) * side
            d = np.stack([np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], np.float64) * (8 + 40 * rs.rand()) * [1.0, 0.4 + 0.6 * rs.rand()] for _ in range(nobj)])
            polys.append((c[:, None, :] + d).reshape(nobj, 8).astype(np.float32))
            labels.append(rs.randint(0, args.nc, size=nobj).astype(np.float32))
        total = npool * side * side * 3
        kw = {} if budget_frac is None else dict(pool_budget_bytes=int(total * budget_frac))        # (one image per slab: per-image LRU)
        ds = BaseDataset(hyp, args.size, True, args.mode == "csl", False, device=dev, decode_workers=aff.get("decode_threads", 8), **kw)
        ds.set_arrays(images, polys, labels)
        loader = DeviceLoader(ds, batch, shuffle=True)
        _r.seed(1234 + rank)
        np.random.seed(1234 + rank)

        def stream():
            while True:
                for b in loader:
                    if b[1].shape[0] == batch:
                        yield b
        it = stream()
        ntg = 0

        def step():
            nonlocal ntg
            _, imgs, targets = next(it)
            ntg += targets.shape[0]
            outs = model(imgs, training=True)
            loss, _ = crit(outs, targets, sync_items=False)
            loss.backward()
            rt.sgd_step(lr, 0.937, grad_scale=dp.grad_scale, zero_grad=True)
        for _ in range(warmup):
            step()
        ntg = 0
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        crit.flush()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        st = ds.cache().stats
        return {"value": round(batch * world * steps / dt, 2), "unit": "img/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
                "targets_per_batch": ntg // max(1, steps), "pool": {"images": npool, "decoded_mb": round(total / 1e6, 1),
                "budget_mb": None if budget_frac is None else round(total * budget_frac / 1e6, 1), "resident_mb": round(ds.cache().resident_bytes() / 1e6, 1),
                "uploads": st["decoded"], "evicted_slabs": st["evicted_slabs"]}}

    def rooflines(m, pmc):
        """roofline / roofline_3x3 / kernels of one measured workload (HIP events of its instrumented pass)."""
        summ = m["timer"].summary()
        steps, dt_inst = m["steps"], m["dt_inst"]

        def roof(k, v):
            # the roofline that binds the class: algorithmic FLOPs against the dense bf16 MFMA peak, or algorithmic bytes (every
            # operand element once) against the HBM peak — whichever fraction is larger (the generic GEMM class is mostly 1x1
            # layers with 8-32 K steps: memory streams)
            tf = v["flops"] / v["seconds"] / 1e12
            gb = v["bytes"] / v["seconds"] / 1e9
            fm, fh = tf / MFMA_BF16_PEAK_TFLOPS, gb / HBM_PEAK_GBS
            head = ({"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fh, 4)} if fh > fm else
                    {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fm, 4)})
            out = {"kernel": k, **head, "frac_mfma": round(fm, 4), "frac_hbm": round(fh, 4),
                   "algorithmic_bytes_per_launch": int(v["bytes"] / max(1, v["launches"])), "traffic": pmc.get(k),
                   "launches_per_step": v["launches"] // steps,
                   "avg_launch_us": round(v["seconds"] / v["launches"] * 1e6, 2),
                   "share_of_step": round(v["seconds"] / dt_inst, 4)}
            if v.get("chip_seconds", v["seconds"]) < 0.999 * v["seconds"]:
                # 8-wave weight-gradient kernels (conv3x3_wgrad8.hip, wgrad1x1_8w.hip): a workgroup holds its CU exclusively and the grid covers
                # PART of the chip (96 of 256 CUs by default: in the timed step the side stream owns those CUs and the main stream the rest — same-box
                # step +1.2 % / +1.7 % over whole-chip grids).  Timed ALONE in this pass such a launch leaves the other CUs idle, so `frac`
                # (whole-chip peak, as the contract defines it) understates the kernel: `frac_of_occupied_cus` prices it against the CUs it holds.
                occ = v["chip_seconds"] / v["seconds"]
                out["cus_occupied_avg"] = round(occ * N_CUS, 1)
                out["frac_of_occupied_cus"] = round(max(fm, fh) / occ, 4)
                out["frac_mfma_of_occupied_cus"] = round(fm / occ, 4)
                out["chip_ms_per_step"] = round(v["chip_seconds"] / steps * 1e3, 3)
                out["note"] = ("CU-exclusive 8-wave kernel sized to part of the chip (RYOLO_W3_V8_BLOCKS / RYOLO_WGRAD_8W_BLOCKS = 96 workgroups = 96 CUs); "
                               "timed alone here, the other 160 CUs idle: frac = whole-chip peak, frac_of_occupied_cus = peak of the CUs held; the same "
                               "kernels on whole-chip grids, alone: profiles/r05_wgrad_isolated.txt (DESIGN.md section 3, round 5)")
            elif "wgrad" in k:
                out["note"] = "side-stream kernel (RYOLO_WGRAD_BLOCKS=512: two 4-wave workgroups per CU); timed alone here"
            elif k == "conv3x3_patch_kernel<256x128>" and out["launches_per_step"] > 44:
                out["note"] = ("class average over EVERY launch on the halo-patch kernel's 128-column tile; since r06 the one-round grids of the 25^2 maps run "
                               "here too (RYOLO_P3_MIN_WGS; batch 64: 15 more launches per step than in r05, 745 TF/s each where the generic kernel ran "
                               "them at 610) and pull the average below the r05 set's (DESIGN.md 3.4, 6)")
            return out
        # dominant kernel class, BOTH ways: `roofline` = the largest share of the CHIP's time (seconds x the
        # fraction of the CUs a launch holds: a kernel that runs on 96 CUs for 10 ms costs the step what a whole-chip kernel costs in 3.75 ms;
        # for whole-chip kernels this is plain time) — the r05 definition; `roofline_by_wall_time` = the largest sum of launch durations timed
        # alone — the r01-r04 definition.  In both objects `frac` / `achieved` are against the WHOLE chip's peak over wall time, as the
        # contract defines them; the occupied-CU view has its own keys (frac_of_occupied_cus, chip_ms_per_step).
        res = {"roofline": dict(roof(*max(summ.items(), key=lambda kv: kv[1]["chip_seconds"])), selected_by="largest chip time (seconds x CUs held / 256)"),
               "roofline_by_wall_time": dict(roof(*max(summ.items(), key=lambda kv: kv[1]["seconds"])), selected_by="largest wall time of its launches timed alone")}
        # BASELINE.json north_star quotes the MFMA fraction of the 3x3 convs separately: the halo-patch kernel (fwd + dgrad)
        k33 = "conv3x3_patch_kernel<256x128>"
        if k33 in summ:
            res["roofline_3x3"] = roof(k33, summ[k33])
        if "conv3x3_ws64_kernel" in summ:                       # the 64 -> 64 channel 3x3 layers on the persistent weight-stationary kernel (r04)
            res["roofline_3x3_ws64"] = roof("conv3x3_ws64_kernel", summ["conv3x3_ws64_kernel"])
        # ALL 3x3 work — stride 1 AND stride 2, forward + data gradient + weight gradient, whichever kernel serves it (r05 left the seven
        # stride-2 layers out and divided by chip time only).  Two denominators, both printed:
        #   frac_wall_alone  FLOPs / sum of the launch durations timed alone (a 96-CU launch counts its full duration)  = `frac`
        #   frac_chip_time   FLOPs / sum of duration x CUs held / 256 (what the launches cost the two-stream step's resources)
        fam = m["timer"].summary(by_fam=True)
        if fam:
            def fam_block(keys):
                fl = sum(fam[k]["flops"] for k in keys)
                sec = sum(fam[k]["seconds"] for k in keys)
                cs = sum(fam[k]["chip_seconds"] for k in keys)
                return {"gflop_per_step": round(fl / steps / 1e9, 1), "ms_per_step": round(sec / steps * 1e3, 3), "chip_ms_per_step": round(cs / steps * 1e3, 3),
                        "frac_wall_alone": round(fl / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "frac_chip_time": round(fl / cs / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                        "launches_per_step": sum(fam[k]["launches"] for k in keys) // steps}
            k3 = sorted(fam)
            allb = fam_block(k3)
            res["roofline_3x3_all"] = {"what": "every 3x3 convolution launch of the step (stride 1 and stride 2; forward, data gradient, weight gradient)",
                                       "bound": "mfma", "achieved": round(allb["frac_wall_alone"] * MFMA_BF16_PEAK_TFLOPS, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": allb["frac_wall_alone"], **allb,
                                       "stride1": fam_block([k for k in k3 if k[0] == "3x3s1"]) if any(k[0] == "3x3s1" for k in k3) else None,
                                       "stride2": fam_block([k for k in k3 if k[0] == "3x3s2"]) if any(k[0] == "3x3s2" for k in k3) else None,
                                       "kernels": {f"{f}:{k}": {"ms_per_step": round(v["seconds"] / steps * 1e3, 3), "chip_ms_per_step": round(v["chip_seconds"] / steps * 1e3, 3),
                                                                "tflops": round(v["flops"] / v["seconds"] / 1e12, 1), "launches_per_step": v["launches"] // steps}
                                                   for (f, k), v in sorted(fam.items())},
                                       "mfma_busy_counters": MFMA_BUSY_PROFILE if os.path.exists(os.path.join(ROOT, MFMA_BUSY_PROFILE)) else None}
        # the pointwise class by regime: K <= 256 layers are memory streams (HBM roof), K > 256 layers sit on the MFMA side
        split = m["timer"].summary(by_sub=True)
        if split:
            groups = {}
            for (k, sub), v in split.items():                       # a regime may be served by several kernels (generic 1x1 instantiation, gemm256)
                gsub = groups.setdefault(sub, {"seconds": 0.0, "flops": 0.0, "launches": 0, "bytes": 0.0, "chip_seconds": 0.0, "kernels": {}})
                for f in ("seconds", "flops", "launches", "bytes", "chip_seconds"):
                    gsub[f] += v[f]
                gsub["kernels"][k] = {"launches_per_step": v["launches"] // steps, "ms_per_step": round(v["seconds"] / steps * 1e3, 3),
                                      "tflops": round(v["flops"] / v["seconds"] / 1e12, 2)}
            res["roofline_1x1_split"] = {sub: dict(roof(f"pointwise layers, {sub}", gsub), kernels=gsub["kernels"]) for sub, gsub in sorted(groups.items())}
        res["kernels"] = {kk: {"tflops": round(vv["flops"] / vv["seconds"] / 1e12, 2), "ms_per_step": round(vv["seconds"] / steps * 1e3, 3),
                               "chip_ms_per_step": round(vv["chip_seconds"] / steps * 1e3, 3), "launches_per_step": vv["launches"] // steps}
                          for kk, vv in summ.items()}
        wg = [kk for kk in summ if "wgrad" in kk]
        if wg:                                      # all weight gradients together: wall time alone (`frac`, the contract's basis) and chip time
            fl, cs = sum(summ[kk]["flops"] for kk in wg), sum(summ[kk]["chip_seconds"] for kk in wg)
            sec = sum(summ[kk]["seconds"] for kk in wg)
            res["roofline_wgrad_all"] = {"kernels": wg, "bound": "mfma", "achieved": round(fl / sec / 1e12, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": round(fl / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "ms_per_step": round(sec / steps * 1e3, 3),
                                         "frac_chip_time": round(fl / cs / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "chip_ms_per_step": round(cs / steps * 1e3, 3),
                                         "what": "FLOPs of every weight-gradient launch; frac = over their wall time timed alone (whole-chip peak), "
                                                 "frac_chip_time = over seconds x CUs held / 256 (the side stream's kernels hold part of the chip)"}
        return res

    torch.cuda.reset_peak_memory_stats(dev)
    m = measure(args.batch, args.steps, args.warmup, not args.no_kernel_timing)
    g_main = rt.graph(args.batch, args.size, args.size, True)
    mem = {"peak_allocated_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
           "plan_arena_gb": round(g_main.layout.total / 1e9, 2) if g_main.layout is not None else None,
           "plan_buffers_one_tensor_each_gb": round(g_main.layout.sum_bytes / 1e9, 2) if g_main.layout is not None else None,
           "what": f"torch.cuda.max_memory_allocated over the batch-{args.batch} run (weights, optimizer state, plan, loss); the plan's "
                   "activations and activation gradients are liveness-placed slots of one arena (engine/arena.py)"}
    # second regime (VERDICT r1 #2 / SURVEY §8(d)): 8 images per GPU = the reference's nominal global batch 64 on 8 GPUs (train.py:150)
    m8 = None
    if not args.no_b8 and args.batch != 8:
        m8 = measure(8, 2 * args.steps, max(args.warmup, 5), not args.no_kernel_timing)
    fed = None
    if not args.no_loader and (args.ver, args.mode) == ("yolov7", "kfiou"):
        fed = {"resident_pool": loader_fed(args.batch, args.steps, max(args.warmup, 5))}
        try:
            # a split four times larger (1024 images: a mosaic batch of 64 touches ~300 of them) under a budget of HALF its decoded size: every
            # batch re-uploads what the LRU dropped (pinned staging + H2D on the loader's stream; the "decode" is a host memcpy here)
            fed["budget_half_of_1024_images"] = loader_fed(args.batch, args.steps, max(args.warmup, 5), budget_frac=0.5, npool=1024)
        except RuntimeError as e:
            fed["budget_half_of_1024_images"] = {"error": str(e)[:300]}
    dist_info = None
    if world > 1:
        # every rank takes part: the collective's own rank count and a bitwise comparison of the replicas after all the steps above
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)
        mx, mn = rt.flat.clone(), rt.flat.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        affs = [None] * world
        dist.all_gather_object(affs, aff)
        tele = [None] * world
        dist.all_gather_object(tele, m.get("telemetry"))

        def spread(pr):
            """What the first SCALE run needs to explain itself: every rank's own step time (before it waits for the others) and the time
            its gradient buckets spent in their collectives on the side stream (from 'bucket final' to 'sum back', peers' skew included)."""
            if pr is None:
                return None
            ms, ar = pr["ms_per_step"], pr["allreduce_ms_per_step"]
            return {"ms_per_step_min": min(ms), "ms_per_step_max": max(ms), "ms_per_step_by_rank": ms, "allreduce_ms_per_step_min": min(ar),
                    "allreduce_ms_per_step_max": max(ar), "allreduce_ms_per_step_by_rank": ar, "grad_wire": pr["wire"]}
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_in_allreduce": int(ones.item()),
                     "replicas_bit_identical": bool(torch.equal(mx, mn)), "devices": "all ranks on cuda:0 (--same-device)" if args.same_device else "one GPU per rank",
                     "grad_bucket_mb": round(dp.bucket_bytes / 2 ** 20, 1) if hasattr(dp, "bucket_bytes") else None,
                     "grad_wire": parallel.pick_wire(args.batch, world, args.wire), "grad_wire_rule": args.wire,
                     "per_rank": spread(m["per_rank"]), "per_rank_b8": spread(m8["per_rank"]) if m8 is not None else None,
                     "affinity": [{"rank": r, **a} for r, a in enumerate(affs)], "rccl_env": parallel.rccl_env(),
                     # how the 256 CUs are shared: weight-gradient side stream (CU-exclusive workgroups) / RCCL channels / main stream (parallel.plan_partition)
                     "cu_partition": parallel.partition(),
                     "gpu_telemetry_by_rank": tele,      # rocm-smi clocks + socket power of every rank's GPU, sampled while its steps were running
                     "note": "allreduce_ms = time of the bucket collectives on their side stream (overlapped with backward); the part of it the step "
                             "actually waits for is ms_per_step(N) - ms_per_step(1)"}
    if rank != 0:
        return
    dt, loss_first, loss_last = m["dt"], m["loss_first"], m["loss_last"]

    ms_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt
    out = {
        "metric": "training img/s @800x800 bf16", "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"C4: DOTA {args.ver} {args.mode} nc={args.nc} {args.size}x{args.size}, batch {args.batch}/GPU, fwd+loss+bwd+SGD-nesterov step"
                               + (f", RCCL all-reduce of 37.9M grads ({parallel.pick_wire(args.batch, world, args.wire)} on the wire)" if world > 1 else ""),
                   "global_batch": args.batch * world, "parallelism": f"dp{world}", "targets_per_image": 64,
                   "weights": "random init N(0,0.02) (train.py:28-33)"},
    }
    # the synthetic batch is fixed, so real training shows as a falling loss; NaN / inf would void the run (rank 0's view)
    out["train_loss"] = {"first_step": round(loss_first, 4), "after_timed_steps": round(loss_last, 4),
                         "finite": bool(math.isfinite(loss_first) and math.isfinite(loss_last) and bool(torch.isfinite(rt.flat).all()))}
    out["hbm_resident"] = mem
    out["host_affinity"] = aff
    if dist_info is not None:
        out["distributed"] = dist_info
    gf = TRAIN_GFLOP_PER_IMG.get((args.ver, args.mode, args.size))
    if gf:
        out["config"]["train_gflop_per_img"] = gf
        out["mfma_roofline_frac_whole_step"] = round(value * gf * 1e9 / (world * MFMA_BF16_PEAK_TFLOPS * 1e12), 4)
    step_bytes = pmc_step_bytes(args)
    if step_bytes and world == 1:
        # whole-step HBM view: PMC bytes of every kernel of one step (committed profile of THIS workload) over the measured step time
        out["hbm_whole_step"] = {"traffic_gb_per_step": round(step_bytes / 1e9, 1), "achieved_gbs": round(step_bytes / (dt / args.steps) / 1e9, 1),
                                 "peak_gbs": HBM_PEAK_GBS, "frac": round(step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                 "source": PMC_PROFILE}
    if _PMC_STATE.get("stale"):
        out["pmc_profile"] = {"file": PMC_PROFILE, "stale": True, "note": "sources changed since the PMC passes were taken: traffic / hbm_whole_step are "
                              "not reported (rerun tools/pmc_step.sh)"}
    if m["timer"] is not None:
        out.update(rooflines(m, pmc_traffic(args)))
        out["kernel_timing"] = {"how": "HIP events around every conv launch, on the launch stream, in a second pass of the same K steps right after the timed region, with the two backward streams serialized (a kernel is timed alone on the GPU; rocprofv3 summary of the same: RYOLO_WGRAD_STREAM=0)",
                                "ms_per_step_instrumented": round(m["dt_inst"] / args.steps * 1e3, 3)}
    if m8 is not None:
        v8 = 8 * world * m8["steps"] / m8["dt"]
        b8 = {"workload": f"the same network and step at 8 images/GPU (global batch {8 * world}; 64 = the reference's nominal batch on 8 GPUs, train.py:150)",
              "value": round(v8, 2), "unit": "img/s", "ms_per_step": round(m8["dt"] / m8["steps"] * 1e3, 3), "steps": m8["steps"],
              "train_loss": {"first_step": round(m8["loss_first"], 4), "after_timed_steps": round(m8["loss_last"], 4)}}
        if gf:
            b8["mfma_roofline_frac_whole_step"] = round(v8 * gf * 1e9 / (world * MFMA_BF16_PEAK_TFLOPS * 1e12), 4)
        if m8["timer"] is not None:
            b8.update(rooflines(m8, {}))
            b8["ms_per_step_instrumented"] = round(m8["dt_inst"] / m8["steps"] * 1e3, 3)
        out["b8"] = b8
    if fed is not None:
        for v in fed.values():
            if "value" in v:
                v["vs_synthetic"] = round(v["value"] / value, 4)
        out["loader_fed"] = {"what": "the same step with every batch assembled by lib.load's DeviceLoader (mosaic / mixup / hsv / warp / flips on the device, "
                                     "on the loader's own stream under the previous step) from a synthetic 256-image 1024x1024 split; `value` above stays the "
                                     "synthetic resident-batch number; image decode is an array lookup here (no files on the GPU box)", **fed}
    # secondary metric of BASELINE.json: rotated-NMS latency at 10k boxes (device time; both sets, both thresholds, with and without the sort)
    if world == 1 and not args.no_infer and (args.ver, args.mode) == ("yolov7", "kfiou"):
        out.update(infer_child(args))
    out["nms"] = nms_block(dev)
    out["nms_ms_10k_boxes"] = out["nms"]["mask_reduce_ms"]["C_0.65"]
    out["nms_ms_10k_boxes_end_to_end_worst"] = max(out["nms"]["end_to_end_ms"].values())
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
