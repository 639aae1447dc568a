#!/bin/bash
out=gpurun_out/r02_e; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "preprocess or forward_vs or kitchen or cpp_wrappers" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
python tools/preprocess_bench.py --per-query 2>&1 | grep preprocess
python tools/preprocess_bench.py 2>&1 | grep preprocess
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python tools/preprocess_bench.py --reps 5 > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_pre.md 2>&1; rm -rf $out/prof; head -12 $out/kernel_stats_pre.md
