mkdir -p gpurun_out/s8
(RYOLO_LIB=$PWD/tools/variants/lib_base.so python tools/bench_bnact.py 10; python tools/bench_bnact.py 10) 2>&1 | grep -v amdgpu.ids > gpurun_out/s8/bnact.txt
cat gpurun_out/s8/bnact.txt
python -m pytest tests/test_gpu_bnfuse.py tests/test_gpu_blocks.py tests/test_gpu_teacher_forced.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab_lib.sh tools/variants/lib_base.so tools/variants/lib_base.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s8/ab.txt
