mkdir -p gpurun_out/s14
python -m pytest tests/test_gpu_bnfuse.py -x -q -m gpu 2>&1 | tail -2
run() { python bench.py --batch 8 --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-loader --no-b8 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; }
for i in 1 2; do
echo "default $(run)"
echo "rows2 $(RYOLO_EW_ROWS=2 run)"
echo "rows8 $(RYOLO_EW_ROWS=8 run)"
echo "folddirect256 $(RYOLO_BN_FOLD_DIRECT=256 run)"
echo "redblocks1024 $(RYOLO_BN_RED_BLOCKS=1024 run)"
echo "wgradblocks512 $(RYOLO_WGRAD_BLOCKS=512 run)"
echo "wgradblocks1024 $(RYOLO_WGRAD_BLOCKS=1024 run)"
done | tee gpurun_out/s14/b8.txt
