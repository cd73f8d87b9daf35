for cfg in "64 100 128 128 3" "64 50 256 256 3" "64 25 512 512 3" "64 100 128 256 3"; do
  W3=0 timeout 120 python tools/bench_wgrad.py $cfg 2>&1 | grep -E "wgrad" | sed "s/^/GEN /"
  W3=1 timeout 120 python tools/bench_wgrad.py $cfg 2>&1 | grep -E "wgrad|rel err" | sed "s/^/W3  /"
done
