# r06 second pass: optimizer tests, captured-step A/B at 8 images with runtime knobs, b8 tables, mAP parity (flip analysis)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_optimizer.py -q -x 2>&1 | tail -n 5
for env in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "AMD_OPT_FLUSH=0" "ROC_SYSTEM_SCOPE_SIGNAL=0"; do
  echo "== $env"; env $env B=8 K=40 timeout 300 python tools/bench_graph_step.py 2>&1 | grep -E "ms/step|Error|error" 
done > $O/graph_step_b8.txt 2>&1
cat $O/graph_step_b8.txt
B=8 TOP=400 RYOLO_WGRAD_STREAM=0 RYOLO_FWD_FORK=0 timeout 600 python tools/profile_layers.py > $O/per_launch_b8.txt 2>&1
cd /tmp; export TMPDIR=/tmp
RYOLO_WGRAD_STREAM=0 RYOLO_FWD_FORK=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b8 -o run -- python $R/bench.py --no-cpu-baseline --no-loader --no-b8 --no-infer --batch 8 --steps 20 > $O/prof_b8.json 2> $O/prof_b8.err
cd $R
timeout 900 python tools/map_parity.py --reverse --seeds 3 > $O/map_parity_reverse.txt 2>&1; tail -n 40 $O/map_parity_reverse.txt
