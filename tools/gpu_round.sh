#!/bin/bash
# One gpurun call: GPU parity tests, bench line, kernel trace. Outputs under gpurun_out/$1
tag=${1:-run}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/prof.log 2>&1
ls $out/prof | head
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats.md 2>&1; head -30 $out/kernel_stats.md
