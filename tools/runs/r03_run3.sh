O=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_aug.py tests/test_pipeline.py tests/test_gpu_dataprep.py tests/test_gpu_teacher_forced.py tests/test_gpu_trajectory.py tests/test_gpu_postprocess.py -m gpu -q -s --durations=8 > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
grep -E "^TEACHER|^TRAJ|passed|failed|rc=|Error|assert" $O/gpu_tests.txt | cut -c1-1500 | head -60
timeout 600 python tools/bench_pipeline.py > $O/pipeline.txt 2>&1; tail -3 $O/pipeline.txt | cut -c1-1500
