set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_optimizer.py -q -x 2>&1 | tail -n 15
B=8 K=40 timeout 300 python tools/bench_graph_step.py > $O/graph_step_b8.txt 2>&1; tail -n 30 $O/graph_step_b8.txt
B=8 SZ=1024 timeout 300 python tools/profile_infer_layers.py > $O/infer_layers_b8_1024.txt 2>&1; head -n 30 $O/infer_layers_b8_1024.txt
B=64 SZ=800 timeout 300 python tools/profile_infer_layers.py > $O/infer_layers_b64_800.txt 2>&1; head -n 30 $O/infer_layers_b64_800.txt
