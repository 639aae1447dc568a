#!/bin/bash
out=gpurun_out/r03_z2; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench_batch.py tests/test_harness.py -m gpu -x -q -s > $out/pytest_model_s.log 2>&1; grep -E "max abs diff|passed|failed|compute_dtype" $out/pytest_model_s.log | cut -c1-400
