#!/bin/bash
# Whole-step matrix-pipe occupancy ("settle the 40 % question with counters"): SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES
# and GRBM_GUI_ACTIVE for EVERY kernel of the training step, per kernel class and for the step.  One rocprofv3 --pmc pass (+ --kernel-trace only;
# never combined with sys / hip tracing).  Run on the GPU box from the repo root:
#   bash tools/pmc_step_mfma.sh [round tag]   ->  gpurun_out/<tag>_pmc_step_mfma_busy.json  (copy to profiles/)
# Counter semantics (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1 024
# SIMDs (= 32 x the number of 32x32x16 bf16 MFMAs); SQ_BUSY_CYCLES is summed over the 32 shader engines' SQs (4 per XCD x 8 XCDs);
# GRBM_GUI_ACTIVE = shader-clock cycles the GPU was busy in the dispatch, summed over the 8 XCDs (calibrated on the first r06 pass: GUI / ns = 15.9
# for every large dispatch = 8 x 1.99 GHz, the clock the guide quotes for profiled passes).  So, per dispatch:
#   matrix pipe busy fraction = (MFMA_BUSY / 1024) / (SQ_BUSY / 32)        [the convention of profiles/r05_pmc_wgrad_ring/SUMMARY.txt]
# rocprofv3 serializes dispatches while it collects counters: a kernel is counted ALONE on the chip (the 96-CU weight-gradient kernels with the
# other 160 CUs idle).  The step-level figure therefore divides the summed matrix cycles per SIMD by the cycles of the UN-profiled two-stream step
# (ms_per_step of a plain bench run on the same box x the mean shader clock of the profiled dispatches).
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
STEPS=3
# the plain step time on this box (two streams, no profiler)
python $R/bench.py --steps 10 --warmup 3 --no-kernel-timing --no-cpu-baseline --no-loader --no-b8 --no-infer > $R/gpurun_out/pmc_mfma_plain.json 2> $R/gpurun_out/pmc_mfma_plain.err
timeout 1200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step_mfma -o p -- \
  python $R/bench.py --steps $STEPS --warmup 1 --no-kernel-timing --no-cpu-baseline --no-loader --no-b8 --no-infer > $R/gpurun_out/pmc_step_mfma.log 2>&1
TAG=$TAG python - <<'PY'
import collections, csv, glob, json, os, re, sys
R = os.environ["GRAFT_REPO_ROOT"]
TAG = os.environ["TAG"]
sys.path.insert(0, R)
import bench
rows = collections.defaultdict(lambda: collections.defaultdict(float))
disp = {}
for f in glob.glob(R + "/gpurun_out/pmc_step_mfma/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
dur, cnt = collections.defaultdict(float), collections.Counter()
for (k, _), ns in disp.items():
    dur[k] += ns * 1e-9
    cnt[k] += 1
# shader clock of the profiled pass: GRBM_GUI_ACTIVE / 8 XCDs / duration over the LONG dispatches only (GUI_ACTIVE also ticks through a dispatch's
# set-up and drain: on 10-us kernels the quotient reads 2.4+ GHz, on every dispatch > 100 us it reads 1.95-2.0)
gui_long = dur_long = 0.0
for f in glob.glob(R + "/gpurun_out/pmc_step_mfma/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and ns > 100000:
            gui_long += float(r["Counter_Value"])
            dur_long += ns * 1e-9
CLOCK_GHZ = gui_long / 8 / dur_long / 1e9 if dur_long else 2.0
steps = cnt.get("sgd_nesterov_kernel", 0)
skip = ("nms_", "bitonic_", "compose_kernel", "topk_", "__amd_rocclr")


def block(keys):
    mf = sum(rows[k]["SQ_VALU_MFMA_BUSY_CYCLES"] for k in keys)
    sq = sum(rows[k]["SQ_BUSY_CYCLES"] for k in keys)
    gui = sum(rows[k]["GRBM_GUI_ACTIVE"] for k in keys)
    mops = sum(rows[k]["SQ_INSTS_VALU_MFMA_MOPS_BF16"] for k in keys)
    sec = sum(dur[k] for k in keys)
    n = sum(cnt[k] for k in keys)
    return {"launches_per_step": round(n / max(steps, 1), 1), "ms_per_step_profiled_alone": round(sec / max(steps, 1) * 1e3, 3),
            "mfma_busy_mcycles_per_step": round(mf / max(steps, 1) / 1e6, 2), "sq_busy_mcycles_per_step": round(sq / max(steps, 1) / 1e6, 3),
            "mfma_busy_frac_of_dispatch": round((mf / 1024) / (sq / 32), 4) if sq else None,
            "mfma_busy_frac_of_wall_clock": round((mf / 1024) / (gui / 8), 4) if gui else None,
            "shader_clock_ghz": round(gui / 8 / sec / 1e9, 3) if sec else None,
            "bf16_mfma_tflop_per_step": round(mops * 512 / max(steps, 1) / 1e12 * 1.0, 3) if mops else None}


kernels = {k: block([k]) for k in rows if not k.startswith(skip) and rows[k]["SQ_VALU_MFMA_BUSY_CYCLES"] > 0}
classes = {}
for cls, pat in bench.PMC_CLASSES.items():
    ks = [k for k in rows if re.match(pat, k)]
    if ks:
        classes[cls] = dict(block(ks), kernels=len(ks))
# the 3x3 family by kernel name (stride-2 3x3 layers run on the generic tapped kernels, which also serve other tapped launches: the name cannot
# separate them — bench.py's roofline_3x3_all does, from the launch descriptors)
allk = [k for k in rows if not k.startswith(skip)]
step = block(allk)
# every kernel that serves 3x3 convolutions, by NAME: the halo-patch / persistent / ring / streaming kernels (3x3 only) + the generic tapped kernels
# (stride-2 3x3 forward and data gradients, small-grid 3x3; the 1x1 instantiations are separate kernels and stay out; conv_wgrad_kernel<128> also
# serves the narrow pointwise weight gradients: included, stated)
pat33 = r"conv3x3|" + bench._gemm_re(128, 128, "false") + "|" + bench._gemm_re(256, 64) + r"|conv_wgrad_kernel<(64|128), |wgrad_taps_dma_kernel"
k33 = [k for k in allk if re.match(pat33, k)]
fam33 = block(k33)
fam33["kernels"] = sorted(k33)
plain = None
try:
    plain = json.loads([ln for ln in open(R + "/gpurun_out/pmc_mfma_plain.json") if ln.startswith("{")][-1])
except Exception as e:
    print("no plain run:", e)
out = {"_doc": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 over bench.py (batch 64, yolov7 kfiou nc=16 "
               "800^2); dispatches are serialized by the profiler (every kernel alone on the chip; the CU-exclusive weight-gradient kernels with 160 CUs idle). "
               "mfma_busy_frac_of_dispatch = (MFMA_BUSY / 1024 SIMDs) / (SQ_BUSY / 32 SQs); mfma_busy_frac_of_wall_clock = (MFMA_BUSY / 1024) / (GRBM_GUI_ACTIVE / 8 XCDs); shader clock = GRBM_GUI_ACTIVE / 8 / duration. "
               "A 32x32x16 bf16 MFMA holds its SIMD's pipe for 32 cycles at 8 passes x 4 cycles: busy 100 % = the dense peak at the running clock.",
       "steps_profiled": steps, "shader_clock_ghz_long_dispatches": round(CLOCK_GHZ, 3), "whole_step_profiled_alone": step,
       "all_3x3_kernels_by_name": fam33, "classes": classes, "kernels": kernels}
if plain:
    ms = plain["ms_per_step"]
    clk = round(CLOCK_GHZ, 3)
    mf_per_simd = step["mfma_busy_mcycles_per_step"] * 1e6 / 1024
    out["whole_step_two_streams"] = {"ms_per_step_unprofiled": ms, "img_s": plain["value"], "shader_clock_ghz_assumed": clk,
                                     "mfma_busy_frac": round(mf_per_simd / (ms * 1e-3 * clk * 1e9), 4),
                                     "what": "matrix cycles per SIMD of one step (counters) / (un-profiled two-stream step time x the profiled passes' mean shader clock)"}
import hashlib
out["source_sha256"] = bench.source_sha256()
json.dump(out, open(R + f"/gpurun_out/{TAG}_pmc_step_mfma_busy.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("shader_clock_ghz_long_dispatches", "whole_step_profiled_alone", "whole_step_two_streams") if k in out}, indent=1))
print("all 3x3 kernels by name:", {k: v for k, v in fam33.items() if k != "kernels"})
for c, v in sorted(classes.items(), key=lambda kv: -kv[1]["mfma_busy_mcycles_per_step"]):
    print(c.ljust(36), v["mfma_busy_frac_of_dispatch"], v["ms_per_step_profiled_alone"], "ms")
PY
