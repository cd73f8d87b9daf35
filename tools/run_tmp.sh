mkdir -p gpurun_out/s15
for v in 1280 640 1024 1536 1280 640 1024 1536; do echo "redblocks=$v $(RYOLO_BN_RED_BLOCKS=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-loader 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["b8"]["value"])')"; done | tee gpurun_out/s15/ab2.txt
