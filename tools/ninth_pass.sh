R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
run() { echo "== $*"; env "$@" 2>&1 | grep "TF/s"; }
( for shp in "8 25 1024 1024 1 1" "8 25 1024 512 1 1" "8 25 512 512 1 1" "8 25 2048 512 1 1" "8 25 512 1024 1 1" "8 50 512 512 1 1" "8 50 1024 256 1 1" "8 50 256 1024 1 1" "8 50 512 256 1 1" "8 100 256 256 1 1" "8 100 512 128 1 1"; do
    run EPI=1 python tools/bench_conv.py $shp 0x201 50
    run EPI=1 python tools/bench_conv.py $shp 0xa01 50
    run EPI=1 RYOLO_GEMM_DEEP=0 python tools/bench_conv.py $shp 0x201 50
    run EPI=1 RYOLO_GEMM_256=2 python tools/bench_conv.py $shp 0x201 50
    run EPI=1 RYOLO_GEMM_WS=2 python tools/bench_conv.py $shp 0x201 50
  done ) > $O/small_m_1x1.txt 2>&1
grep -A1 "^== " $O/small_m_1x1.txt | grep -v "^--" | paste - - | sed 's/python tools\/bench_conv.py//; s/in [0-9]* MB out [0-9]* MB//' | cut -c1-200
