"""Fixture G12: keep sets of the C oracle (oracle/rotated_iou.c, detectron2 semantics) for the C5-size candidate sets — 50 000 boxes,
clustered (C) and uniform (U) sets of SURVEY §8(d), thresholds 0.2 (detect.py:91) and 0.65 (test.py:270), `>` (CUDA) semantics.  The
inputs regenerate from ryolov4_amd.synth.synth_nms_boxes(50000, dist, seed=9); only the keep indices are stored (int32, KB-scale).
Run here or on any host (the oracle is this repo's own C restatement):  python tests/golden/make_golden_nms50k.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from ryolov4_amd.synth import synth_nms_boxes  # noqa: E402


def main():
    out = {}
    for dist in ("C", "U"):
        b, s = synth_nms_boxes(50000, dist, seed=9)
        for thr in (0.2, 0.65):
            t = time.time()
            keep = oracle.nms_rotated(b, s, thr, True)
            assert keep.max() < 2 ** 31
            out[f"{dist}_{thr}"] = keep.astype(np.int32)
            print(f"G12 {dist} thr {thr}: {len(keep)} kept, {time.time() - t:.1f} s on one core", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g12_nms50k.npz"), **out)


if __name__ == "__main__":
    main()
