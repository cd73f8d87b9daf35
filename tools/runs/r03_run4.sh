O=$GRAFT_REPO_ROOT/gpurun_out/r03d; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_conv3x3.py tests/test_gpu_forced_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_trajectory.py tests/test_gpu_bnfuse.py tests/test_gpu_pool.py -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
tail -4 $O/gpu_tests.txt
for t1 in 0 1 0 1; do RYOLO_GEMM_T1=$t1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-b8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T1=$t1', d['value'], d['ms_per_step'])"; done
B=64 TOP=60 RYOLO_WGRAD_STREAM=0 RYOLO_FWD_FORK=0 timeout 600 python tools/profile_layers.py > $O/per_launch_b64.txt 2>&1; head -32 $O/per_launch_b64.txt
timeout 600 python tools/map_parity.py 400 yolov7 kfiou > $O/map_parity.txt 2>&1; tail -2 $O/map_parity.txt | cut -c1-1500
