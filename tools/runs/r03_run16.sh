cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_forced_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_teacher_forced.py -x -q 2>&1 | tail -5
for i in 1 2; do
  for v in 0 1; do RYOLO_GEMM_N64=$v python bench.py --steps 12 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('N64=$v', d['value'], d['ms_per_step'])"; done
done
B=64 RYOLO_GEMM_N64=0 python tools/profile_layers.py > gpurun_out/r03_pl_n64_0.txt 2>&1
B=64 RYOLO_GEMM_N64=1 python tools/profile_layers.py > gpurun_out/r03_pl_n64_1.txt 2>&1
