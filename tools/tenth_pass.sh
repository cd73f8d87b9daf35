R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R
run() { echo "== $*"; env "$@" 2>&1 | grep "TF/s"; }
( for shp in "8 50 512 512 3 2" "8 100 256 256 3 2" "8 200 128 128 3 2" "8 50 256 256 3 2" "8 100 128 128 3 2" "8 400 64 128 3 2" "1 50 512 512 3 2" "1 100 256 256 3 2" "1 200 128 128 3 2"; do
    run EPI=1 python tools/bench_conv.py $shp 0x201 50
    run EPI=1 python tools/bench_conv.py $shp 0xa01 50
    run EPI=1 python tools/bench_conv.py $shp 0x301 50
  done ) > $O/small_m_s2.txt 2>&1
grep -A1 "^== " $O/small_m_s2.txt | grep -v "^--" | paste - - | sed 's/python tools\/bench_conv.py//; s/in [0-9]* MB out [0-9]* MB//' | cut -c1-200
