#!/bin/bash
out=gpurun_out/r03_g2; mkdir -p $out; export TMPDIR=/tmp
REGTR_F16_GATHER=1 timeout 300 rocprofv3 --kernel-trace -d $out/prof2 -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --parity-pairs 0 > $out/prof2.log 2>&1
db=$(find $out/prof2 -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_f16gather.md 2>&1; python tools/trace_forward.py $db > $out/forward_trace_f16gather.md 2>&1; rm -rf $out/prof2; grep -E "gather|pair_planes" $out/kernel_stats_f16gather.md; grep -E "gather_f16|gather_mfma|pair_planes" $out/forward_trace_f16gather.md | head -24
