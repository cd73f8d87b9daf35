R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x --durations=8 > $O/gpu_test_suite.txt 2>&1; tail -n 14 $O/gpu_test_suite.txt
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['b8']['value'], d['infer_c5']['img_s'], d['infer_800_b64']['forward_img_s'], d['roofline']['frac'])"
timeout 600 python tools/bench_infer.py > $O/infer.txt 2>&1; tail -n 5 $O/infer.txt
