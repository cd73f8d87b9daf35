O=$GRAFT_REPO_ROOT/gpurun_out/r03i; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -c "
import torch
for p in (-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print('prio', p, s.priority)
    except Exception as e: print(p, 'ERR', e)
"
timeout 900 python -m pytest tests/test_gpu_foldfinalize.py tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_bnfuse.py tests/test_gpu_stem.py -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt
for cfg in "0 1" "1 1" "0 0" "1 0" "0 1" "-1 1"; do set -- $cfg; RYOLO_SIDE_PRIO=$1 RYOLO_FUSE_FOLD=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('PRIO=$1 FOLD=$2', d['value'], d['ms_per_step'], 'b8', d['b8']['value'], d['b8']['ms_per_step'])"; done
