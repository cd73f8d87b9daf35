O=$GRAFT_REPO_ROOT/gpurun_out/r03k; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wgrad3x3.py tests/test_gpu_blocks.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_forced_kernels.py -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt
for shape in "64 100 128 128 3 1" "64 50 256 256 3 1" "64 100 128 256 3 1" "64 25 512 512 3 1" "64 25 512 1024 3 1" "64 50 256 512 3 1"; do
  for v in 1 3; do RYOLO_W3_STEP64=$v CHECK=0 python tools/bench_wgrad.py $shape 20 2>/dev/null | tail -1 | sed "s/^/S64=$v /"; done
done
CHECK=1 python tools/bench_wgrad.py 4 50 128 160 3 1 2 2>/dev/null | tail -2
for v in 1 3 1 3; do RYOLO_W3_STEP64=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-b8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('S64=$v', d['value'], d['ms_per_step'])"; done
