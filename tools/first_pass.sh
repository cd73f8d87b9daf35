set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
tail -c 600 $O/bench_default.err
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/gpu_test_suite.txt 2>&1; tail -n 25 $O/gpu_test_suite.txt
cd /tmp; export TMPDIR=/tmp
RYOLO_WGRAD_STREAM=0 RYOLO_FWD_FORK=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b64 -o run -- python $R/bench.py --no-cpu-baseline --no-loader --no-b8 --no-infer --steps 8 > $O/prof_b64.json 2> $O/prof_b64.err
cd $R
B=64 TOP=400 RYOLO_WGRAD_STREAM=0 RYOLO_FWD_FORK=0 timeout 600 python tools/profile_layers.py > $O/per_launch_b64.txt 2>&1
python tools/highres_floor_table.py $O/per_launch_b64.txt > $O/highres_layers_vs_floor.txt 2>&1
timeout 1500 bash tools/pmc_step_mfma.sh r06 > $O/pmc_step_mfma.txt 2>&1; tail -n 30 $O/pmc_step_mfma.txt
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['b8']['value'], d.get('hbm_whole_step'))"
cat $O/bench_time.txt; cat $O/highres_layers_vs_floor.txt
