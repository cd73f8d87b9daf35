cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03m
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03m/prof_b64_two -o run -- python $R/bench.py --no-cpu-baseline --no-b8 --no-kernel-timing --steps 8 > $R/gpurun_out/r03m/prof_b64_two.json 2> $R/gpurun_out/r03m/prof_b64_two.err
tail -n 1 $R/gpurun_out/r03m/prof_b64_two.json | cut -c1-200
