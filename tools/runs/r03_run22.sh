cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in 0 1; do RYOLO_GEMM_SPLITK=$v B=1,8 python tools/bench_infer.py 2>/dev/null | grep -o "^[0-9]* \|'fwd_ms': [0-9.]*\|'graph_fwd_ms': [0-9.]*\|'graph_fwd_pp_captured_ms': [0-9.]*" | tr '\n' ' ' | sed "s/^/SPLITK=$v /"; echo; done; done
