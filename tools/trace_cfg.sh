#!/bin/bash
# kernel statistics + last-forward trace of any bench configuration:  bash tools/trace_cfg.sh TAG NAME bench args...
tag=$1; name=$2; shift; shift; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py "$@" --no-cpu-baseline --no-roofline --parity-pairs 0 --no-strict-f32 > $out/prof_$name.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_$name.md 2>&1; python tools/trace_forward.py $db > $out/forward_trace_$name.md 2>&1; rm -rf $out/prof; head -28 $out/kernel_stats_$name.md | cut -c1-110; tail -1 $out/forward_trace_$name.md
