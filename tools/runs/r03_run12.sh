O=$GRAFT_REPO_ROOT/gpurun_out/r03j; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wgrad3x3.py tests/test_gpu_blocks.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt
for shape in "64 200 64 64 3 1" "64 400 64 64 3 1" "64 100 64 64 3 1" "64 200 32 64 3 1"; do
  for v in 0 1; do RYOLO_W3_CO64_V2=$v CHECK=0 python tools/bench_wgrad.py $shape 20 2>/dev/null | tail -1 | sed "s/^/V2=$v /"; done
done
CHECK=1 python tools/bench_wgrad.py 4 100 64 64 3 1 2 2>/dev/null | tail -2
for v in 0 1 0 1; do RYOLO_W3_CO64_V2=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-b8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('V2=$v', d['value'], d['ms_per_step'])"; done
