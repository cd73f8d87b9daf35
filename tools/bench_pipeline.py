"""Feed rate of the device-side loader (datasets/base_dataset.py: BaseDataset.assemble_batch) at the benchmark's batch: 64 samples of
800x800 with the reference's augmentation hyper-parameters (data/hyp.yaml: mosaic 1.0, mixup 0.15, hsv, rotate / scale / translate,
flips) from a synthetic pool of 256 DOTA-sized images (1024x1024, 40 polygons each) resident in HBM.  Reports batches/s and img/s of
the whole assembler (host planning + ~10 launches), wall clock, against the training step it has to feed.
usage: python tools/bench_pipeline.py [batch] [size] [iters]  -> gpurun_out/pipeline.json"""
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ryolov4_amd.datasets.base_dataset import BaseDataset

HYP = {"hsv_h": 0.015, "hsv_s": 0.7, "hsv_v": 0.4, "rotate": 45, "translate": 0.1, "scale": 0.5, "flipud": 0.5, "fliplr": 0.5, "mosaic": 1.0, "mixup": 0.15}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 800
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    npool, side = 256, 1024
    base = rs.randint(0, 256, size=(side + 64, side + 64, 3)).astype(np.uint8)
    images = [base[o:o + side, o:o + side] for o in rs.randint(0, 64, size=npool)]
    polys, labels = [], []
    for _ in range(npool):
        c = rs.rand(40, 2) * side
        d = (rs.rand(40, 4, 2) - 0.5) * 60
        polys.append((c[:, None, :] + d).reshape(40, 8).astype(np.float32))
        labels.append(rs.randint(0, 16, size=40).astype(np.float32))
    res = {}
    for csl in (False, True):
        ds = BaseDataset(HYP, S, True, csl, False, device=dev)
        ds.set_arrays(images, polys, labels)
        random.seed(0)
        np.random.seed(0)
        for _ in range(3):
            ds.assemble_batch([random.randrange(npool) for _ in range(B)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nt = 0
        for _ in range(iters):
            _, imgs, tg = ds.assemble_batch([random.randrange(npool) for _ in range(B)])
            nt += tg.shape[0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        res["csl" if csl else "kfiou"] = {"ms_per_batch": round(dt * 1e3, 2), "img_per_s": round(B / dt, 1), "targets_per_batch": nt // iters}
    out = {"what": f"BaseDataset.assemble_batch, batch {B} at {S}x{S}, augment=True (mosaic 1.0, mixup 0.15, hsv, warp, flips), pool of {npool} "
                   f"{side}x{side} uint8 images in HBM ({npool * side * side * 3 / 1e6:.0f} MB), wall clock incl. host planning and the count read-back",
           "result": res, "training_step_to_feed_img_per_s": 750}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/pipeline.json", "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
