mkdir -p gpurun_out/s17
python -m pytest tests/test_gpu_gemm1x1.py tests/test_gpu_wgrad3x3.py tests/test_gpu_wgrad_taps.py tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_teacher_forced.py tests/test_gpu_parity_e2e.py -x -q -m gpu 2>&1 | tail -3
bash tools/ab_lib.sh tools/variants/lib_base.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s17/ab.txt
