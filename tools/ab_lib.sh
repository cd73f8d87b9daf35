# Same-box A/B of libryolo_hip.so builds with one ABI (RYOLO_LIB=...): bench.py per variant, the in-tree build first and last.
# usage (on the GPU box): bash tools/ab_lib.sh tools/variants/lib_X.so [tools/variants/lib_Y.so ...]
for v in "" "$@" ""; do
  if [ -z "$v" ]; then r=$(python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-infer 2>&1 | tail -1); else r=$(RYOLO_LIB=$PWD/$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-infer 2>&1 | tail -1); fi
  echo "${v:-base} $(echo "$r" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
