for cfg in "64 100 256 128 1" "64 50 512 256 1" "64 200 128 64 1"; do
  GEMM_TIMING=1 RYOLO_LIB=tools/variants/lib_gemmtiming.so EPI=1 timeout 120 python tools/bench_conv.py $cfg 1 0x1 20 2>&1 | grep -E "kernel[01]:|blocks"
done
