R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R
run() { echo "== $*"; env "$@" 2>&1 | grep "TF/s"; }
( for shp in "8 25 256 256 3 1" "8 25 512 512 3 1" "8 25 1024 512 3 1" "8 50 256 256 3 1" "8 50 128 128 3 1" "8 100 128 128 3 1" "8 32 512 512 3 1"; do
    for epi in 1; do
      run EPI=$epi python tools/bench_conv.py $shp 0x201 50
      run EPI=$epi RYOLO_GEMM_DEEP=0 python tools/bench_conv.py $shp 0x201 50
      run EPI=$epi RYOLO_GEMM_DEEP=4 python tools/bench_conv.py $shp 0x201 50
      run EPI=$epi python tools/bench_conv.py $shp 0xa01 50
      run EPI=$epi python tools/bench_conv.py $shp 0x301 50
      run EPI=$epi python tools/bench_conv.py $shp 0x601 50
      run EPI=$epi python tools/bench_conv.py $shp 0x200 50
    done
  done ) > $O/small_m_3x3.txt 2>&1
cat $O/small_m_3x3.txt
