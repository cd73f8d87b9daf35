mkdir -p gpurun_out/s10
for cfg in "8192 8192" "1048576 8192" "8192 4096" "1048576 8192" "8192 8192"; do set -- $cfg; echo "grid2=$1 redblocks=$2 $(RYOLO_EW_GRID2=$1 RYOLO_BN_RED_BLOCKS=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["b8"]["value"])')"; done | tee gpurun_out/s10/ab.txt
python -m pytest tests/test_gpu_bnfuse.py tests/test_gpu_blocks.py tests/test_gpu_teacher_forced.py tests/test_gpu_pool.py -x -q -m gpu 2>&1 | tail -2
