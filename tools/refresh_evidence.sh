# Short refresh of the evidence that depends on the sources' hash or on the inference path (after a late change): PMC step traffic, default bench
# line, inference timings, GPU suite -> gpurun_out/${TAG}m/ (then tools/copy_evidence.sh <tag>)
TAG=${1:-r06}
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}m
mkdir -p $O
cd $R
timeout 1500 bash tools/pmc_step.sh $TAG > $O/pmc_step.txt 2>&1
cp gpurun_out/${TAG}_pmc_step_traffic.json profiles/${TAG}_pmc_step_traffic.json
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
timeout 600 python tools/bench_infer.py > $O/infer.txt 2>&1; cp gpurun_out/infer.json $O/infer.json
B=8 SZ=1024 CONF=0.0005 IOU=0.65 timeout 600 python tools/bench_infer.py > $O/infer_1024.txt 2>&1; cp gpurun_out/infer.json $O/infer_1024_b8.json
timeout 1800 python -m pytest tests -m gpu -q --durations=10 > $O/gpu_test_suite.txt 2>&1; tail -n 3 $O/gpu_test_suite.txt
tail -n 1 $O/pmc_step.txt; tail -c 300 $O/bench_default.json | head -c 10; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['b8']['value'], d['hbm_whole_step'])
e=json.load(open('$O/infer.json')); print({k:(v['fwd_ms'],v['graph_fwd_ms'],v['fwd_pp_ms'],v['graph_fwd_pp_captured_ms']) for k,v in e.items()})"
