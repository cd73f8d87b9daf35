# End-of-step tail: how long the weight-gradient stream runs alone after the main stream's last backward kernel (rocprofv3 kernel trace of bench.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/tt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-loader --no-b8 > /tmp/tt.log 2>&1
f=$(find /tmp/tt -name '*kernel_trace.csv' | head -1)
[ -z "$f" ] && { tail -3 /tmp/tt.log; exit 1; }
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sg = [i for i, r in enumerate(rows) if "sgd_nesterov" in r["Kernel_Name"]]
for a, b in zip(sg[:-1], sg[1:]):
    step = rows[a + 1:b + 1]
    t0 = int(step[0]["Start_Timestamp"])
    stem = [r for r in step if "stem3x3_bwd" in r["Kernel_Name"]]
    if not stem: continue
    e_main = int(stem[-1]["End_Timestamp"])
    side = [r for r in step if any(k in r["Kernel_Name"] for k in ("wgrad", "head_da", "head_wgrad_finish"))]
    e_side = max(int(r["End_Timestamp"]) for r in side)
    after = [r for r in side if int(r["End_Timestamp"]) > e_main]
    print(f"step {(int(step[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms: main's last backward kernel ends at {(e_main - t0) / 1e6:.2f} ms, weight-gradient stream at {(e_side - t0) / 1e6:.2f} ms "
          f"(tail {(e_side - e_main) / 1e6:.2f} ms, {len(after)} launches end after it)")
    for r in after[-12:]:
        print(f"     {r['Kernel_Name'][:60]:60s} start {(int(r['Start_Timestamp']) - t0) / 1e6:7.2f} dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us")
PY
