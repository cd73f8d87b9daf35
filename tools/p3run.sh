for cfg in "64 100 128 128 3" "64 50 256 256 3" "64 100 256 128 1" "64 200 64 64 3" "64 25 512 512 3" "64 50 512 256 1" "64 200 64 128 3 2"; do
  timeout 120 python tools/bench_wgrad.py $cfg 2>&1 | grep -E "wgrad|rel err"
done
