R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_conv3x3.py tests/test_gpu_model.py tests/test_gpu_blocks.py tests/test_gpu_teacher_forced.py tests/test_gpu_parity_e2e.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -n 8
run() { echo "== $*"; env "$@" 2>&1 | grep "TF/s"; }
( for shp in "8 25 256 256 3 1" "8 25 512 512 3 1" "8 50 256 256 3 1" "8 50 128 128 3 1" "8 100 128 128 3 1" "8 32 512 512 3 1" "1 25 256 256 3 1" "1 50 128 128 3 1" "1 100 128 128 3 1" "1 25 512 512 3 1" "1 25 1024 512 3 1" "8 200 64 64 3 1" "1 200 64 64 3 1" "1 400 64 64 3 1"; do
    for epi in 1 2; do
      run EPI=$epi python tools/bench_conv.py $shp 0x201 50
      run EPI=$epi RYOLO_P3_SMALL_BN64=0 python tools/bench_conv.py $shp 0x201 50
      run EPI=$epi RYOLO_P3_MIN_WGS=100000 python tools/bench_conv.py $shp 0x201 50
    done
  done ) > $O/small_m_3x3_ab.txt 2>&1
cat $O/small_m_3x3_ab.txt
for v in "X=1" "RYOLO_P3_MIN_WGS=512" "X=2" "RYOLO_P3_MIN_WGS=512"; do echo "== $v"; env $v python bench.py --batch 8 --steps 40 --no-cpu-baseline --no-loader --no-b8 --no-infer --no-kernel-timing 2>/dev/null | tail -1 | cut -c1-120; done
for v in "X=1" "RYOLO_P3_MIN_WGS=512"; do echo "== $v"; env $v python bench.py --infer-only 2>&1 | tail -n 1 | cut -c1-1300; done
