cd $GRAFT_REPO_ROOT
for shape in "64 200 64 64" "64 400 64 64" "64 100 128 128" "64 50 256 256"; do
  for pipe in 0x201 0x1201; do RYOLO_LIB=$PWD/tools/variants/lib_p3_ablate.so EPI=1 python tools/bench_conv.py $shape 3 1 $pipe 20 2>/dev/null | head -1 | sed "s/^/ablate-lib /"; done
done
