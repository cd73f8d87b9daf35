# rocprofv3 kernel stats of the batch-64 inference forward (tools/bench_infer.py, B=64): top kernels
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pi
B=${B:-64} timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -o p -- python $R/tools/bench_infer.py > /tmp/pi.log 2>&1
f=$(find /tmp/pi -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} {float(r["TotalDurationNs"])/tot*100:5.1f} %  avg_us {float(r["AverageNs"])/1e3:9.1f}')
PY
else tail -5 /tmp/pi.log; fi
