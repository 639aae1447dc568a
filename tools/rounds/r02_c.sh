#!/bin/bash
out=gpurun_out/r02_c; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=8 > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
grep -E "stress|passed|failed|Error|error|assert|s call" $out/pytest.log | tail -30
timeout 600 python bench.py --points 100000 --pairs 8 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_stress.json 2> $out/bench_stress.err; tail -c 1500 $out/bench_stress.json
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py --points 100000 --pairs 8 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_stress.md 2>&1; rm -rf $out/prof; head -16 $out/kernel_stats_stress.md
