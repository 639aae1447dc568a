#!/bin/bash
out=gpurun_out/r02_g; mkdir -p $out; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $out/prof -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/trace_forward.py $db > $out/forward_trace.md 2>&1; rm -rf $out/prof; tail -3 $out/forward_trace.md
