O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_teacher_forced.py tests/test_gpu_trajectory.py tests/test_gpu_bench_contract.py tests/test_gpu_model.py -m gpu -q -s --durations=8 > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
grep -E "^TEACHER|^TRAJ|^DROPIN|passed|failed|rc=|Error|assert" $O/gpu_tests.txt | cut -c1-1800 | head -60
