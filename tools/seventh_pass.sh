R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R
run() { echo "== $*"; env "$@" 2>&1 | grep "TF/s"; }
( for shp in "1 25 256 256 3 1" "1 25 512 512 3 1" "1 50 128 128 3 1" "1 25 1024 512 3 1"; do
    run EPI=2 python tools/bench_conv.py $shp 0x201 50
    run EPI=2 RYOLO_P3_MIN_WGS=1 python tools/bench_conv.py $shp 0x201 50
  done
  for shp in "8 100 128 128 3 1" "8 100 256 256 3 1" "8 50 512 512 3 1" "64 25 512 512 3 1" "64 50 128 128 3 1" "64 50 256 256 3 1" "64 25 1024 512 3 1" "64 100 128 128 3 1"; do
    for epi in 1; do
    run EPI=$epi python tools/bench_conv.py $shp 0x201 50
    run EPI=$epi python tools/bench_conv.py $shp 0x2201 50
    done
  done ) > $O/bn64_threshold.txt 2>&1
cat $O/bn64_threshold.txt
