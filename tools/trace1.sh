#!/bin/bash
# one-pair (or N-pair) kernel trace + stats:  bash tools/trace1.sh TAG [pairs]
tag=${1:-trace1}; n=${2:-1}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py --pairs $n --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --parity-pairs 0 --no-strict-f32 > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_pairs$n.md 2>&1; python tools/trace_forward.py $db > $out/forward_trace_pairs$n.md 2>&1; rm -rf $out/prof; tail -1 $out/forward_trace_pairs$n.md
