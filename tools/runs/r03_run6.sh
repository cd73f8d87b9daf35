O=$GRAFT_REPO_ROOT/gpurun_out/r03f; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_forced_kernels.py tests/test_gpu_stem.py tests/test_gpu_head.py -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
for shape in "64 200 256 256 1 1" "64 100 512 512 1 1" "64 50 1024 1024 1 1" "64 200 128 128 1 1" "64 100 256 128 1 1" "64 25 2048 512 1 1" "64 400 32 64 3 2" "64 200 64 128 3 2" "64 50 256 256 3 2"; do
  for lib in tools/variants/lib_w3_old.so ""; do
    if [ -z "$lib" ]; then CHECK=0 python tools/bench_wgrad.py $shape 20 2>/dev/null | tail -1 | sed 's/^/new /'; else RYOLO_LIB=$PWD/$lib CHECK=0 python tools/bench_wgrad.py $shape 20 2>/dev/null | tail -1 | sed 's/^/old /'; fi
  done
done
CHECK=1 python tools/bench_wgrad.py 8 40 64 96 3 2 2 2>/dev/null | tail -2
CHECK=1 python tools/bench_wgrad.py 8 40 64 96 1 1 2 2>/dev/null | tail -2
bash tools/ab_lib.sh tools/variants/lib_w3_old.so
