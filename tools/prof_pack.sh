# pack_weights / sgd kernel times of the in-tree library and of variants (rocprofv3 kernel trace of the 8-image bench).  usage: bash tools/prof_pack.sh [lib.so ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "-" "$@"; do
  rm -rf /tmp/pk
  if [ "$v" = "-" ]; then timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o p -- python $R/bench.py --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-loader --no-b8 > /tmp/pk.log 2>&1
  else RYOLO_LIB=$R/$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o p -- python $R/bench.py --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-loader --no-b8 > /tmp/pk.log 2>&1; fi
  f=$(find /tmp/pk -name '*kernel_stats.csv' | head -1)
  echo "== $v"
  if [ -n "$f" ]; then grep "pack_weights\|sgd_nesterov" "$f" | cut -d, -f1-5; else tail -3 /tmp/pk.log; fi
done
