cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_forced_kernels.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5
for i in 1 2; do
  for v in 0 1; do RYOLO_GEMM_WS=$v python bench.py --steps 12 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('WS=$v', d['value'], d['ms_per_step'], d.get('b8',{}).get('value'))"; done
done
B=64 RYOLO_GEMM_WS=1 python tools/profile_layers.py > gpurun_out/r03_pl_ws1.txt 2>&1
