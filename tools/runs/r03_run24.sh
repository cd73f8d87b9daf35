cd $GRAFT_REPO_ROOT
RYOLO_LIB=$PWD/tools/variants/lib_rp.so timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
  python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('base', d['value'], d['ms_per_step'], 'b8', d.get('b8',{}).get('value'), {k:v['ms_per_step'] for k,v in d['kernels'].items() if 'gemm' in k})"
  RYOLO_LIB=$PWD/tools/variants/lib_rp.so python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('RP  ', d['value'], d['ms_per_step'], 'b8', d.get('b8',{}).get('value'), {k:v['ms_per_step'] for k,v in d['kernels'].items() if 'gemm' in k})"
done
