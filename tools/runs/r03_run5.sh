O=$GRAFT_REPO_ROOT/gpurun_out/r03e; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_wgrad3x3.py tests/test_gpu_blocks.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
for shape in "64 100 128 128" "64 50 256 256" "64 200 64 64" "64 400 64 64" "64 25 512 512" "64 100 128 256"; do
  for lib in tools/variants/lib_w3_old.so ""; do
    if [ -z "$lib" ]; then CHECK=0 python tools/bench_wgrad.py $shape 3 1 20 2>/dev/null | tail -1 | sed 's/^/new /'; else RYOLO_LIB=$PWD/$lib CHECK=0 python tools/bench_wgrad.py $shape 3 1 20 2>/dev/null | tail -1 | sed 's/^/old /'; fi
  done
done
bash tools/ab_lib.sh tools/variants/lib_w3_old.so
timeout 900 python tools/map_parity.py 300 yolov7 kfiou 2>/dev/null | tail -1 | cut -c1-1800
