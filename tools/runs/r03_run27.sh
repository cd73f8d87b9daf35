cd $GRAFT_REPO_ROOT
export RYOLO_LIB=$PWD/tools/variants/lib_s2d256.so
RYOLO_GEMM_S2D256=1 timeout 600 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for i in 1 2; do for v in 0 1; do
  RYOLO_GEMM_S2D256=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-b8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('S2D256=$v', d['value'], d['ms_per_step'])"
done; done
B=64 RYOLO_GEMM_S2D256=1 python tools/profile_layers.py 2>&1 | grep "taps4x1 400x400\|sum "
