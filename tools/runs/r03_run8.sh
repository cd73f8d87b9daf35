O=$GRAFT_REPO_ROOT/gpurun_out/r03h; mkdir -p $O; cd $GRAFT_REPO_ROOT
for shape in "8 40 64 96 1 1 2" "4 25 256 396 1 1 2" "3 13 128 160 1 1 2" "2 20 512 128 1 1 2"; do CHECK=1 python tools/bench_wgrad.py $shape 2>/dev/null | tail -2 | tr '\n' ' '; echo; done
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_head.py -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
for shape in "64 200 256 256 1 1" "64 100 512 512 1 1" "64 50 1024 1024 1 1" "64 200 128 128 1 1" "64 100 256 128 1 1" "64 25 2048 512 1 1" "64 100 256 396 1 1"; do
  for mode in 1 3; do RYOLO_WGRAD_P1=$mode CHECK=0 python tools/bench_wgrad.py $shape 20 2>/dev/null | tail -1 | sed "s/^/P1=$mode /"; done
done
for m in 1 3 1 3; do RYOLO_WGRAD_P1=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-b8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P1=$m', d['value'], d['ms_per_step'])"; done
