"""Where the loader-fed training step spends its time: host planning vs launches vs waits, loader alone, step alone, both.
usage: python tools/bench_loader.py [batch] [size] [iters]   -> gpurun_out/loader_diag.json"""
import json
import os
import random
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ryolov4_amd.datasets.base_dataset import BaseDataset, DeviceLoader
from ryolov4_amd.lib.loss import ComputeKFIoULoss
from ryolov4_amd.model.yolo import Yolo
from ryolov4_amd.synth import CFG, HYP, synth_batch

AUG = {"hsv_h": 0.015, "hsv_s": 0.7, "hsv_v": 0.4, "rotate": 45, "translate": 0.1, "scale": 0.5, "flipud": 0.5, "fliplr": 0.5, "mosaic": 1.0, "mixup": 0.15}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 800
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    npool, side, nobj = 256, 1024, 40
    base = rs.randint(0, 256, size=(side + 64, side + 64, 3)).astype(np.uint8)
    images = [np.ascontiguousarray(base[o:o + side, o:o + side]) for o in rs.randint(0, 64, size=npool)]
    polys, labels = [], []
    for _ in range(npool):
        c = rs.rand(nobj, 2) * side
        d = (rs.rand(nobj, 4, 2) - 0.5) * 60
        polys.append((c[:, None, :] + d).reshape(nobj, 8).astype(np.float32))
        labels.append(rs.randint(0, 16, size=nobj).astype(np.float32))
    out = {}
    torch.manual_seed(42)
    model = Yolo(16, CFG, "kfiou", "yolov7").to(dev).train()
    rt = model.runtime()
    crit = ComputeKFIoULoss(model, HYP)
    simgs, stg = synth_batch(B, S, 16, False, seed=42, per_image=64)
    simgs, stg = simgs.to(dev), stg.to(dev)

    def step(imgs, tg):
        loss, _ = crit(model(imgs, training=True), tg, sync_items=False)
        loss.backward()
        rt.sgd_step(0.01, 0.937, zero_grad=True)

    for _ in range(4):
        step(simgs, stg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = 0.0
    for _ in range(iters):
        a = time.perf_counter()
        step(simgs, stg)
        th += time.perf_counter() - a
    torch.cuda.synchronize()
    out["step_alone_ms"] = (time.perf_counter() - t0) / iters * 1e3
    out["step_alone_host_enqueue_ms"] = th / iters * 1e3
    for side_stream in (True, False):
        ds = BaseDataset(AUG, S, True, False, False, device=dev, decode_workers=8)
        ds.set_arrays(images, polys, labels)
        loader = DeviceLoader(ds, B, shuffle=True, side_stream=side_stream)
        random.seed(1)
        np.random.seed(1)

        def stream():
            while True:
                for b in loader:
                    if b[1].shape[0] == B:
                        yield b
        it = stream()
        for _ in range(5):
            next(it)
        torch.cuda.synchronize()
        # loader alone
        t0 = time.perf_counter()
        for _ in range(iters):
            next(it)
        torch.cuda.synchronize()
        key = "side" if side_stream else "same"
        out[f"loader_alone_ms_{key}"] = (time.perf_counter() - t0) / iters * 1e3
        # phases of one batch on the host (planning + launches vs the count read-back), measured by patching finalize_batch
        from ryolov4_amd.datasets import base_dataset as BD
        orig = BD.finalize_batch
        marks = []

        def timed(*a, **k):
            marks.append(time.perf_counter())
            r = orig(*a, **k)
            marks.append(time.perf_counter())
            return r
        BD.finalize_batch = timed
        tt = []
        for _ in range(iters):
            a = time.perf_counter()
            next(it)
            tt.append((a, time.perf_counter()))
        BD.finalize_batch = orig
        torch.cuda.synchronize()
        plan = np.mean([marks[2 * i] - tt[i][0] for i in range(iters)]) * 1e3
        fin = np.mean([marks[2 * i + 1] - marks[2 * i] for i in range(iters)]) * 1e3
        out[f"host_plan_and_launch_ms_{key}"] = plan
        out[f"finalize_incl_count_readback_ms_{key}"] = fin
        # GPU time per C-ABI entry of one batch (HIP events around every hip.call of the loader)
        from ryolov4_amd import hip as H
        real_call, evs = H.call, []

        def traced(name, *a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            real_call(name, *a)
            e1.record()
            evs.append((name, e0, e1))
        H.call = traced
        for _ in range(4):
            next(it)
        torch.cuda.synchronize()
        H.call = real_call
        agg = {}
        for name, e0, e1 in evs:
            agg[name] = agg.get(name, 0.0) + e0.elapsed_time(e1) / 4
        out.update({f"gpu_ms_per_batch_{key}:{k}": v for k, v in agg.items()})
        # loader-fed step
        for _ in range(3):
            _, imgs, tg = next(it)
            step(imgs, tg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tn = ts = 0.0
        for _ in range(iters):
            a = time.perf_counter()
            _, imgs, tg = next(it)
            b = time.perf_counter()
            step(imgs, tg)
            c = time.perf_counter()
            tn += b - a
            ts += c - b
        torch.cuda.synchronize()
        out[f"fed_step_ms_{key}"] = (time.perf_counter() - t0) / iters * 1e3
        out[f"fed_host_next_ms_{key}"] = tn / iters * 1e3
        out[f"fed_host_step_enqueue_ms_{key}"] = ts / iters * 1e3
        del ds, loader
        BD._POOL_CACHE.clear()
    out = {k: round(v, 2) for k, v in out.items()}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/loader_diag.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
