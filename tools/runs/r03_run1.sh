# round-3 GPU call 1: full GPU suite with the new parity tests + a same-day bench baseline
O=$GRAFT_REPO_ROOT/gpurun_out/r03a; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
grep -E "TEACHER|TRAJ|DROPIN|passed|failed|rc=" $O/gpu_tests.txt | cut -c1-1500
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
