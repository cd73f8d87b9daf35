# per-kernel time of the head / loss kernels under one knob's two settings (rocprofv3 kernel trace of bench.py, serialized streams)
# usage (GPU box): bash tools/prof_head.sh [KNOB]      KNOB default RYOLO_HEAD_SPARSE; e.g. RYOLO_HEAD_FUSED
K=${1:-RYOLO_HEAD_SPARSE}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in 0 1; do
  rm -rf /tmp/ph$v
  env $K=$v RYOLO_WGRAD_STREAM=0 RYOLO_FWD_FORK=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph$v -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-loader --no-b8 > /tmp/ph$v.log 2>&1
  f=$(find /tmp/ph$v -name '*kernel_stats.csv' | head -1)
  echo "== $K=$v"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"all kernels {tot/1e6:.2f} ms")
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("head_", "loss_", "fold_rows", "32, 3, true, true", "colsum", "chan_add")):
        print(f'{n[:86]:86s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:9.3f} avg_us {float(r["AverageNs"])/1e3:9.1f}')
PY
done
