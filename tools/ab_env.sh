# Same-box A/B of runtime knobs: bench.py per environment setting, alternating, N rounds.
# usage (on the GPU box): bash tools/ab_env.sh N "VAR=a" "VAR=b" ...   (use "-" for the default environment)
n=$1; shift
for i in $(seq $n); do
  for e in "$@"; do
    if [ "$e" = "-" ]; then r=$(python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-loader --no-infer 2>&1 | tail -1);
    else r=$(env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-loader --no-infer 2>&1 | tail -1); fi
    echo "$e $(echo "$r" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], "b8", d["b8"]["value"])')"
  done
done
