python -m pytest tests/test_aug.py tests/test_pipeline.py tests/test_imgds.py tests/test_gpu_dataprep.py -x -q 2>&1 | tail -3
