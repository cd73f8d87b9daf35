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
    # batch 2; threads = min(32, host threads) — the all-threads attempt of r02-r04 never finished a step on the 256-thread
    # hosts and is gone (bench.cpu_baseline docstring); the sample names the thread count and the host's
    assert "batch 2" in cb["sample"] and cb["host_cpu_count"] == os.cpu_count()
    assert cb["cores"] == min(32, os.cpu_count()) and f"torch.set_num_threads({cb['cores']})" in cb["sample"]
    assert d["host_affinity"]["cpus"]                                                    # the rank is pinned to its GPU's cores (or says why not)
    # second regime in the same line: 8 images per GPU, with its own roofline
    b8 = d["b8"]
    assert b8["unit"] == "img/s" and b8["value"] > 0 and abs(b8["value"] - 8 / (b8["ms_per_step"] / 1e3)) < 1e-2 * b8["value"]
    assert b8["roofline"]["bound"] in ("hbm", "mfma") and 0 < b8["roofline"]["frac"] < 1
    # rotated NMS at 10 k boxes: both sets, both thresholds, pre-sorted kernel time and end to end (sort included)
    nms = d["nms"]
    for part in ("mask_reduce_ms", "end_to_end_ms"):
        assert set(nms[part]) == {"U_0.65", "U_0.2", "C_0.65", "C_0.2"} and all(v > 0 for v in nms[part].values())
    assert d["nms_ms_10k_boxes"] == nms["mask_reduce_ms"]["C_0.65"]


def test_bench_gpus_2_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` WITHOUT a launcher must re-exec under torch.distributed.run (never report a single rank as n_gpus 2):
    two ranks over gloo on the one GPU of the test box, overlapped bucketed all-reduce of the flat gradient buffer, replicas compared
    bit for bit after the steps (train.py:150-152,198-202)."""
    import torch
    gc.collect()
    torch.cuda.empty_cache()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--same-device", "--batch", "2", "--size", "256",
                        "--steps", "2", "--warmup", "1", "--no-b8", "--no-kernel-timing"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4 and d["scaling"] == "weak"
    assert abs(d["value"] - 4 / (d["ms_per_step"] / 1e3)) < 1e-2 * d["value"]
    di = d["distributed"]
    assert di["world_size"] == 2 and di["ranks_in_allreduce"] == 2 and di["backend"] == "gloo" and di["replicas_bit_identical"] is True
    tl = d["train_loss"]
    assert tl["finite"] is True and tl["after_timed_steps"] < tl["first_step"]
    assert "cpu_baseline" not in d                                     # rank 0 at N = 1 only
    # round 5 (8-GPU readiness): what a first SCALE run needs to explain itself — every rank's own step time, the time its gradient buckets
    # spent in their collectives, the wire format chosen by rule (2 images per GPU -> bf16), and DISTINCT core sets per rank
    pr = di["per_rank"]
    assert len(pr["ms_per_step_by_rank"]) == 2 and 0 < pr["ms_per_step_min"] <= pr["ms_per_step_max"] <= d["ms_per_step"] * 1.05
    assert len(pr["allreduce_ms_per_step_by_rank"]) == 2 and pr["allreduce_ms_per_step_min"] > 0
    assert di["grad_wire_rule"] == "auto" and di["grad_wire"] == "bf16" and pr["grad_wire"] == "bf16"
    af = di["affinity"]
    assert [a["rank"] for a in af] == [0, 1] and all(a["cpus"] for a in af)
    if all(a["bound"] for a in af) and os.cpu_count() >= 2:
        from ryolov4_amd.parallel import _parse_cpulist
        assert set(_parse_cpulist(af[0]["cpus"])).isdisjoint(_parse_cpulist(af[1]["cpus"]))
    assert isinstance(di["rccl_env"], dict)
    # round 6: with world > 1 the CU-exclusive weight-gradient partition shrinks by the RCCL channel count, the channel
    # count is pinned, and every rank's GPU clocks / socket power (rocm-smi, sampled under load) are in the line
    cp = di["cu_partition"]
    assert cp["world"] == 2 and cp["rccl_channels"] == 8 and cp["side_cus"] == 88 and cp["w3_v8_blocks"] == 88 and cp["wgrad_8w_blocks"] == 88
    assert cp["side_cus"] + cp["main_cus"] + cp["rccl_channels"] == 256
    assert di["rccl_env"].get("NCCL_MIN_NCHANNELS") == di["rccl_env"].get("NCCL_MAX_NCHANNELS") == "8"
    tele = di["gpu_telemetry_by_rank"]
    assert len(tele) == 2 and all(isinstance(t, dict) and ("error" in t or "power_w" in t or "sclk" in t) for t in tele), tele


def test_bench_gpus_n_without_enough_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("an 8-GPU node can run it")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 8" in (r.stderr + r.stdout) and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
