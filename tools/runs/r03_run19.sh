cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in 0 1; do RYOLO_GEMM_WS=$v B=8,64 python tools/bench_infer.py 2>/dev/null | grep -o "^[0-9]* \|'fwd_img_s': [0-9.]*\|'graph_fwd_img_s': [0-9.]*\|'fwd_pp_img_s': [0-9.]*" | tr '\n' ' ' | sed "s/^/WS=$v /"; echo; done; done
