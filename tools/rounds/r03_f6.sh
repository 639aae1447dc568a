#!/bin/bash
out=gpurun_out/r03_f6; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench_batch.py tests/test_harness.py tests/test_gpu_stress.py -m gpu -q -s > $out/pytest_model.log 2>&1; grep -E "max abs diff|compute_dtype|f16 pair:|passed|failed" $out/pytest_model.log | sed 's/^\.*//' | cut -c1-400
