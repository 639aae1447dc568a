#!/bin/bash
out=gpurun_out/r02_q; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for cfg in "--pairs 1 --steps 100 --warmup 10" "--pairs 2 --steps 60 --warmup 5" "--pairs 8 --steps 30 --warmup 5"; do
  tag=$(echo $cfg | tr -d ' -' | cut -c1-20)
  timeout 300 python bench.py $cfg --no-cpu-baseline --no-roofline > $out/bench_$tag.json 2> $out/bench_$tag.err; python - <<PY
import json; d=json.loads(open('$out/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value'],1), round(d['ms_per_step'],3))
PY
done
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py --pairs 1 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_pairs1.md 2>&1; rm -rf $out/prof; head -8 $out/kernel_stats_pairs1.md; tail -1 $out/kernel_stats_pairs1.md
