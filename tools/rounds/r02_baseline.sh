#!/bin/bash
# round-2 baseline: B=1 latency, worst-case (--shuffle) locality, kernel stats of each -> gpurun_out/$1
out=gpurun_out/${1:-r02_a}; mkdir -p $out; export TMPDIR=/tmp
for cfg in "--pairs 1 --steps 50 --warmup 5" "--pairs 64 --shuffle --steps 10 --warmup 2" "--pairs 64 --steps 10 --warmup 2"; do
  tag=$(echo $cfg | tr -d ' -' | cut -c1-24)
  timeout 300 python bench.py $cfg --no-cpu-baseline > $out/bench_$tag.json 2> $out/bench_$tag.err; tail -c 600 $out/bench_$tag.json
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o trace -- python bench.py $cfg --no-cpu-baseline --no-roofline > $out/prof_$tag.log 2>&1
  db=$(find $out/prof_$tag -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_$tag.md 2>&1; rm -rf $out/prof_$tag
  head -14 $out/kernel_stats_$tag.md
done
