#!/bin/bash
# kernel-trace of a short bench run -> gpurun_out/$1/kernel_stats.md (top 30)
out=gpurun_out/$1; mkdir -p $out; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats.md 2>&1; rm -f $db; head -${2:-30} $out/kernel_stats.md; tail -1 $out/kernel_stats.md
