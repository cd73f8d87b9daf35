"""GPU: the bench.py contract — one JSON line on stdout with the keys the driver and the judge read, on a small workload."""
import gc
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16", "--size", "256"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "train_loss"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "img/s" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] - 16 / (d["ms_per_step"] / 1e3)) < 1e-2 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["train_loss"]["finite"] is True
    # batch 2, all host threads — tried first; a host where that oversubscribes torch-CPU falls back and says so
    assert "batch 2" in cb["sample"] and cb["host_cpu_count"] == os.cpu_count()
    assert f"torch.set_num_threads({os.cpu_count()})" in cb["sample"]                    # the all-threads attempt is always made and reported
    # second regime in the same line: 8 images per GPU, with its own roofline
    b8 = d["b8"]
    assert b8["unit"] == "img/s" and b8["value"] > 0 and abs(b8["value"] - 8 / (b8["ms_per_step"] / 1e3)) < 1e-2 * b8["value"]
    assert b8["roofline"]["bound"] in ("hbm", "mfma") and 0 < b8["roofline"]["frac"] < 1
    # rotated NMS at 10 k boxes: both sets, both thresholds, pre-sorted kernel time and end to end (sort included)
    nms = d["nms"]
    for part in ("mask_reduce_ms", "end_to_end_ms"):
        assert set(nms[part]) == {"U_0.65", "U_0.2", "C_0.65", "C_0.2"} and all(v > 0 for v in nms[part].values())
    assert d["nms_ms_10k_boxes"] == nms["mask_reduce_ms"]["C_0.65"]
