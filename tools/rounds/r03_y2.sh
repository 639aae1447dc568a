#!/bin/bash
out=gpurun_out/r03_y2; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python tools/host_profile.py 1 > $out/host_profile_p1.txt 2>&1; head -3 $out/host_profile_p1.txt
REGTR_ONE_CALL_XENC=0 timeout 300 python tools/host_profile.py 1 > $out/host_profile_p1_opbyop.txt 2>&1; head -3 $out/host_profile_p1_opbyop.txt
timeout 300 rocprofv3 --kernel-trace -d $out/prof2 -o trace -- python bench.py --pairs 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --parity-pairs 0 > $out/prof2.log 2>&1
db=$(find $out/prof2 -name "*.db" | head -1); python tools/trace_forward.py $db > $out/forward_trace_p1.md 2>&1; python tools/rocpd_stats.py $db > $out/kernel_stats_p1.md 2>&1; rm -rf $out/prof2; tail -1 $out/forward_trace_p1.md; head -30 $out/kernel_stats_p1.md
