cd $GRAFT_REPO_ROOT
for shape in "64 200 128 128" "64 200 128 256" "64 200 256 256" "64 200 256 128" "64 100 256 512" "64 100 256 256" "64 100 128 512" "64 50 256 1024" "64 100 256 400"; do
  for epi in 1 0; do
    for ws in 0 1; do RYOLO_GEMM_WS=$ws EPI=$epi timeout 120 python tools/bench_conv.py $shape 1 1 0x201 20 2>/dev/null | tr '\n' ' ' | sed "s/^/ws$ws epi$epi /"; echo; done
  done
done
echo "--- forced small / ragged"
for shape in "3 25 256 400" "5 31 128 136" "2 40 64 128" "1 8 96 8"; do
  RYOLO_GEMM_WS=2 EPI=1 timeout 120 python tools/bench_conv.py $shape 1 1 0x201 3 2>/dev/null | tr '\n' ' '; echo
done
