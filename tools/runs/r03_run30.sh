cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gemm1x1.py -x -q 2>&1 | tail -12
RYOLO_GEMM_WS=2 timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_teacher_forced.py -x -q 2>&1 | tail -3
for i in 1 2 3; do for v in 0 1; do
  RYOLO_GEMM_WS_POOL=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-b8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('WS_POOL=$v', d['value'], d['ms_per_step'])"
done; done
