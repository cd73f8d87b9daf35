cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_head.py tests/test_gpu_teacher_forced.py tests/test_gpu_forced_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_parity_e2e.py -x -q 2>&1 | tail -6
for i in 1 2; do
  for v in 0 1; do RYOLO_GEMM_SPLITK=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('SPLITK=$v', d['value'], d['ms_per_step'], 'b8', d.get('b8',{}).get('value'), d.get('b8',{}).get('ms_per_step'))"; done
done
B=8 TOP=400 RYOLO_WGRAD_STREAM=0 RYOLO_FWD_FORK=0 python tools/profile_layers.py > gpurun_out/r03_pl_b8_splitk.txt 2>&1
