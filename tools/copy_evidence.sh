#!/bin/bash
# Copy the end-of-round evidence of tools/collect_evidence.sh (gpurun_out/<tag>m/, merged back by gpurun) into profiles/ under round names.
TAG=${1:-r06}
O=gpurun_out/${TAG}m
P=profiles
last() { tail -n 1 "$1"; }
last $O/bench_default.json > $P/${TAG}_bench_b64.json
cp $O/prof_b64/run_kernel_stats.csv $P/${TAG}_rocprofv3_kernel_stats_b64.csv
cp $O/prof_b8/run_kernel_stats.csv $P/${TAG}_rocprofv3_kernel_stats_b8.csv
last $O/prof_b8.json > $P/${TAG}_bench_b8.json
last $O/prof_b64.json > $P/${TAG}_bench_b64_under_rocprofv3_serialized.json
cp gpurun_out/${TAG}_pmc_step_traffic.json $P/${TAG}_pmc_step_traffic.json
cp $O/per_launch_b64.txt $P/${TAG}_per_launch_table_b64.txt
cp $O/per_launch_b8.txt $P/${TAG}_per_launch_table_b8.txt
cp $O/bnact_passes.txt $P/${TAG}_bnact_passes.txt
[ -f $O/wgrad_isolated.txt ] && grep -v "^+" $O/wgrad_isolated.txt > $P/${TAG}_wgrad_isolated.txt
[ -f $O/s2c32_isolated.txt ] && cp $O/s2c32_isolated.txt $P/${TAG}_s2c32_isolated.txt
[ -f gpurun_out/${TAG}_pmc_step_mfma_busy.json ] && cp gpurun_out/${TAG}_pmc_step_mfma_busy.json $P/${TAG}_pmc_step_mfma_busy.json
cp $O/highres_layers_vs_floor.txt $P/${TAG}_highres_layers_vs_floor.txt
[ -f $O/pmc_wgrad.txt ] && { mkdir -p $P/${TAG}_pmc_wgrad_ring; grep -v "^+" $O/pmc_wgrad.txt > $P/${TAG}_pmc_wgrad_ring/raw_counters.txt; }
( cat $O/power_clocks_bench.txt; echo; cat $O/power_clocks.txt ) > $P/${TAG}_power_clocks.txt
cp $O/nms_times.json $P/${TAG}_nms_times.json
cp $O/infer.json $P/${TAG}_infer_yolov7_kfiou_800.json
cp $O/infer_1024_b8.json $P/${TAG}_infer_yolov7_kfiou_1024_b8_graph.json
[ -f $O/infer_layers_b64_800.txt ] && cp $O/infer_layers_b64_800.txt $P/${TAG}_infer_layers_b64_800.txt
[ -f $O/infer_layers_b8_1024.txt ] && cp $O/infer_layers_b8_1024.txt $P/${TAG}_infer_layers_b8_1024.txt
[ -f $O/graph_step_b8.txt ] && grep -E "ms/step|loss after" $O/graph_step_b8.txt > $P/${TAG}_graph_step_b8.txt
python - <<PY
import json
O, P, T = "$O", "$P", "$TAG"
def last(f): return json.loads(open(f).read().strip().splitlines()[-1])
json.dump({k: last(f"{O}/cfg_{k}.json") for k in ("yolov4_kfiou", "yolov7_csl", "yolov5_kfiou")}, open(f"{P}/{T}_bench_other_configs_b64.json", "w"), indent=1)
json.dump({f"batch{b}": last(f"{O}/batch{b}.json") for b in (96, 128)}, open(f"{P}/{T}_bench_batch_sweep.json", "w"), indent=1)
PY
cp $O/overfit_kfiou.json $P/${TAG}_overfit_yolov7_kfiou_b64_800.json
cp $O/overfit_csl.json $P/${TAG}_overfit_yolov7_csl_b64_800.json
( echo "== 2 ranks, gloo, one shared GPU (tools/dp_check.py) =="; cat $O/dp_check_gloo2.txt; echo; echo "== 1 rank, RCCL (BACKEND=nccl tools/dp_check.py) =="; cat $O/dp_check_rccl1.txt ) > $P/${TAG}_dp_check.txt
cp $O/gpu_test_suite.txt $P/${TAG}_gpu_test_suite.txt
[ -f gpurun_out/${TAG}_map_parity.json ] && cp gpurun_out/${TAG}_map_parity.json $P/${TAG}_map_parity.json
cp $O/loader_diag.json $P/${TAG}_loader_in_the_loop.json
[ -f gpurun_out/${TAG}_iou_fuzz.json ] && cp gpurun_out/${TAG}_iou_fuzz.json $P/${TAG}_iou_fuzz.json
cp $O/pipeline.json $P/${TAG}_pipeline_feed_rate.json
for f in teacher_forced trajectory; do [ -f gpurun_out/r03_$f.json ] && cp gpurun_out/r03_$f.json $P/${TAG}_$f.json; done
[ -f gpurun_out/r02_parity_e2e.json ] && cp gpurun_out/r02_parity_e2e.json $P/${TAG}_parity_e2e.json
ls -la $P | grep ${TAG}_
