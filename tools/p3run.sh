for cfg in "64 100 256 128 1" "64 50 512 256 1" "64 200 128 64 1" "64 25 1024 512 1" "64 100 128 128 3" "64 200 64 64 3"; do
  set -- $cfg
  RYOLO_LIB=tools/variants/libryolo_old.so EPI=1 timeout 120 python tools/bench_conv.py $1 $2 $3 $4 $5 1 0x1 20 2>&1 | grep -E "kernel[01]:" | sed "s/^/OLD /"
  EPI=1 timeout 120 python tools/bench_conv.py $1 $2 $3 $4 $5 1 0x1 20 2>&1 | grep -E "kernel[01]:|img 63|stats" | sed 's/^/NEW /'
done
for cfg in "64 200 64 128 3 2" "64 100 128 128 3 2"; do
  set -- $cfg
  RYOLO_LIB=tools/variants/libryolo_old.so EPI=1 timeout 120 python tools/bench_conv.py $1 $2 $3 $4 $5 $6 0x1 20 2>&1 | grep -E "kernel[01]:" | sed "s/^/OLD /"
  EPI=1 timeout 120 python tools/bench_conv.py $1 $2 $3 $4 $5 $6 0x1 20 2>&1 | grep -E "kernel[01]:|img 63|stats" | sed 's/^/NEW /'
done
