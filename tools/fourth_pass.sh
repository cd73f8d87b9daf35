set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_conv3x3.py tests/test_gpu_conv3x3_ws.py tests/test_gpu_gemm256.py tests/test_gpu_gemm1x1.py tests/test_gpu_model.py tests/test_gpu_postprocess.py -q -x 2>&1 | tail -n 15
( for cfg in "64 400 64 64 3 1 0x201" "64 200 64 64 3 1 0x201" "64 200 256 256 1 1 0x201" "64 200 256 128 1 1 0x201" "64 200 128 128 1 1 0x201" "64 100 512 512 1 1 0x201" "64 50 1024 1024 1 1 0x201" "64 100 512 256 1 1 0x201"; do
    for e in 0 2; do EPI=$e python tools/bench_conv.py $cfg 20 2>&1 | grep "TF/s" | sed "s/^/EPI=$e /"; done
  done ) > $O/bench_conv_eval.txt 2>&1; cat $O/bench_conv_eval.txt
B=64 SZ=800 timeout 300 python tools/profile_infer_layers.py > $O/infer_layers_b64_800.txt 2>&1; head -n 16 $O/infer_layers_b64_800.txt
B=8 SZ=1024 timeout 300 python tools/profile_infer_layers.py > $O/infer_layers_b8_1024.txt 2>&1; head -n 16 $O/infer_layers_b8_1024.txt
timeout 600 python bench.py --infer-only 2>&1 | tail -n 3 | cut -c1-1500
