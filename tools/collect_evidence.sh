# End-of-round evidence on ONE GPU box (run from the repo root through gpurun): PMC step traffic + matrix-pipe occupancy, default bench line, rocprofv3
# kernel stats (b64, b8), per-launch tables, NMS / inference timings, other configurations, batch sweep, overfit curves, DP checks -> gpurun_out/<tag>m/
# usage: bash tools/collect_evidence.sh [round tag, default r06]
TAG=${1:-r06}
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}m
mkdir -p $O
cd $R
# 1. PMC traffic of the step (then the bench line reads it) and the whole-step matrix-pipe occupancy (r06)
timeout 1500 bash tools/pmc_step.sh $TAG > $O/pmc_step.txt 2>&1
cp gpurun_out/${TAG}_pmc_step_traffic.json profiles/${TAG}_pmc_step_traffic.json
timeout 1500 bash tools/pmc_step_mfma.sh $TAG > $O/pmc_step_mfma.txt 2>&1
cp gpurun_out/${TAG}_pmc_step_mfma_busy.json profiles/${TAG}_pmc_step_mfma_busy.json
# 2. default bench line
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
# 3. rocprofv3 kernel stats, serialized streams, b64 and b8
cd /tmp; export TMPDIR=/tmp
RYOLO_WGRAD_STREAM=0 RYOLO_FWD_FORK=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b64 -o run -- python $R/bench.py --no-cpu-baseline --no-loader --no-b8 --steps 8 > $O/prof_b64.json 2> $O/prof_b64.err
RYOLO_WGRAD_STREAM=0 RYOLO_FWD_FORK=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b8 -o run -- python $R/bench.py --no-cpu-baseline --no-loader --no-b8 --batch 8 --steps 20 > $O/prof_b8.json 2> $O/prof_b8.err
cd $R
# 4. per-launch tables
B=64 TOP=400 RYOLO_WGRAD_STREAM=0 RYOLO_FWD_FORK=0 timeout 600 python tools/profile_layers.py > $O/per_launch_b64.txt 2>&1
B=8 TOP=400 RYOLO_WGRAD_STREAM=0 RYOLO_FWD_FORK=0 timeout 600 python tools/profile_layers.py > $O/per_launch_b8.txt 2>&1
python tools/highres_floor_table.py $O/per_launch_b64.txt > $O/highres_layers_vs_floor.txt 2>&1
# 4a. r06: the streaming kernels of the 32 -> 64 stride-2 layer against the kernels they replace, alternating (the 8-wave weight-gradient kernels are
# unchanged since r05: profiles/r05_wgrad_isolated.txt and profiles/r05_pmc_wgrad_ring/ stand)
set +x
( for i in 1 2 3; do
    echo "new  $(python tools/bench_s2c32.py 2>&1 | tail -1) | $(MODE=fwd python tools/bench_s2c32.py 2>&1 | tail -1)"
    echo "old  $(RYOLO_S2C32_DGRAD=0 python tools/bench_s2c32.py 2>&1 | tail -1) | $(MODE=fwd RYOLO_S2C32=0 python tools/bench_s2c32.py 2>&1 | tail -1)"
  done
  echo "8 images: new $(python tools/bench_s2c32.py 8 800 50 2>&1 | tail -1) | $(MODE=fwd python tools/bench_s2c32.py 8 800 50 2>&1 | tail -1)"
  echo "8 images: old $(RYOLO_S2C32_DGRAD=0 python tools/bench_s2c32.py 8 800 50 2>&1 | tail -1) | $(MODE=fwd RYOLO_S2C32=0 python tools/bench_s2c32.py 8 800 50 2>&1 | tail -1)"
) > $O/s2c32_isolated.txt 2>&1
set -x
# 4b. BatchNorm + activation passes alone (forward, backward reduce + finalize, backward apply) and socket power / clocks while the step runs
timeout 300 python tools/bench_bnact.py 10 > $O/bnact_passes.txt 2>&1
set +x
( for i in $(seq 1 100); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power \(W\)" | sed 's/=*//g' | tr -s ' \t' ' ' | tr '\n' ' '; echo; sleep 0.25; done ) > $O/power_clocks.txt 2>/dev/null &
SMI=$!
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-b8 --no-loader 2>/dev/null | tail -1 | cut -c1-160 > $O/power_clocks_bench.txt
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
set -x
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" >> $O/power_clocks.txt
# 5. NMS, inference
timeout 600 python tools/time_nms.py > $O/nms.txt 2>&1; cp gpurun_out/nms_times.json $O/nms_times.json
timeout 600 python tools/bench_infer.py > $O/infer.txt 2>&1; cp gpurun_out/infer.json $O/infer.json
# C5 at its own size: 1024^2, 8 images per GPU, forward + decode + top-K + rotated NMS in one hipGraph, confidence threshold low enough
# that >= 50 k candidates per image reach the top-K / NMS stage
B=8 SZ=1024 CONF=0.0005 IOU=0.65 timeout 600 python tools/bench_infer.py > $O/infer_1024.txt 2>&1; cp gpurun_out/infer.json $O/infer_1024_b8.json
# per-launch tables of the eval tape (r06): the bench block's batch-64 800^2 forward and C5's batch-8 1024^2 forward
B=64 SZ=800 timeout 300 python tools/profile_infer_layers.py > $O/infer_layers_b64_800.txt 2>&1
B=8 SZ=1024 timeout 300 python tools/profile_infer_layers.py > $O/infer_layers_b8_1024.txt 2>&1
B=8 K=40 timeout 300 python tools/bench_graph_step.py > $O/graph_step_b8.txt 2>&1
# 6. other configs + batch sweep
for cfg in "yolov4 kfiou 608 2" "yolov7 csl 800 16" "yolov5 kfiou 800 16"; do set -- $cfg; timeout 300 python bench.py --ver $1 --mode $2 --size $3 --nc $4 --no-cpu-baseline --no-loader --no-b8 --steps 8 > $O/cfg_$1_$2.json 2> $O/cfg_$1_$2.err; done
for b in 96 128; do timeout 300 python bench.py --batch $b --no-cpu-baseline --no-loader --no-b8 --no-kernel-timing --steps 8 > $O/batch$b.json 2> $O/batch$b.err; done
# 7. overfit curves
timeout 600 python tools/overfit_curve.py kfiou 300 > $O/overfit_kfiou.json 2> $O/overfit_kfiou.err
timeout 600 python tools/overfit_curve.py csl 300 > $O/overfit_csl.json 2> $O/overfit_csl.err
# 8. DP check: 2 ranks gloo on the one GPU, and 1 rank through RCCL
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_check.py > $O/dp_check_gloo2.txt 2>&1
BACKEND=nccl timeout 600 python tools/dp_check.py > $O/dp_check_rccl1.txt 2>&1
# 9. loader feed rate, mAP parity (600 steps), GPU test suite
timeout 600 python tools/bench_pipeline.py > $O/pipeline.txt 2>&1; cp gpurun_out/pipeline.json $O/pipeline.json
timeout 300 python tools/bench_loader.py 64 800 10 > $O/loader_diag.txt 2>&1; cp gpurun_out/loader_diag.json $O/loader_diag.json
# mAP parity where the detector detects: HIP-trained weights at mAP@0.5 >= 0.5 evaluated on both paths + the flip analysis, the matched-candidate protocol and
# the bf16 noise floor of the metric (r06); once at C1's image size (416^2)
timeout 900 python tools/map_parity.py --reverse --seeds 3 > $O/map_parity_reverse.txt 2>&1
timeout 900 python tools/map_parity.py --reverse --seeds 1 --size 416 --chunk 500 --max-steps 4000 > $O/map_parity_reverse_416.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q --durations=10 > $O/gpu_test_suite.txt 2>&1; tail -n 3 $O/gpu_test_suite.txt
ls -la $O
tail -n 3 $O/pmc_step.txt; cat $O/bench_time.txt; tail -n 2 $O/dp_check_gloo2.txt; tail -n 2 $O/dp_check_rccl1.txt
