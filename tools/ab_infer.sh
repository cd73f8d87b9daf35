# Same-box A/B of libryolo_hip.so builds on the INFERENCE forward (tools/bench_infer.py, batch 1 and 64): in-tree build and variants, alternating.
# usage (GPU box): bash tools/ab_infer.sh tools/variants/lib_X.so
for v in "" "$@" "" "$@"; do
  if [ -z "$v" ]; then B=1,64 python tools/bench_infer.py > /dev/null 2>&1; else RYOLO_LIB=$PWD/$v B=1,64 python tools/bench_infer.py > /dev/null 2>&1; fi
  echo "${v:-base} $(python -c "
import json; d=json.load(open('gpurun_out/infer.json')); print({k:(v['fwd_ms'], v.get('graph_fwd_ms')) for k,v in d.items() if isinstance(v,dict)})")"
done
