cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gemm1x1.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_teacher_forced.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for i in 1 2; do for v in 0 1; do
  RYOLO_GEMM_WS_S2D=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('WS_S2D=$v', d['value'], d['ms_per_step'], 'b8', d['b8']['value'])"
done; done
B=64 python tools/profile_layers.py 2>&1 | grep "taps4x1 400x400\|^sum "
