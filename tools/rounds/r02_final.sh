#!/bin/bash
# final round-2 evidence: gpu_round (tests, default bench with CPU baseline, kernel stats, PMC traffic + MFMA) and the other bench lines
BENCH_ARGS="" bash tools/gpu_round.sh r02_z
out=gpurun_out/r02_z; export TMPDIR=/tmp
for cfg in "--pairs 1 --steps 50 --warmup 5" "--shuffle --steps 10 --warmup 2" "--points 100000 --pairs 8 --steps 6 --warmup 2" "--config modelnet --steps 10 --warmup 2" "--config modelnet --dtype fp32 --steps 10 --warmup 2"; do
  tag=$(echo $cfg | tr -d ' -' | cut -c1-26)
  timeout 400 python bench.py $cfg --no-cpu-baseline > $out/bench_$tag.json 2> $out/bench_$tag.err
  python - <<PY
import json; d=json.loads(open('$out/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['bound'], round(d['roofline']['frac'],4))
PY
done
timeout 300 rocprofv3 --kernel-trace -d $out/prof2 -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $out/prof2.log 2>&1
db=$(find $out/prof2 -name "*.db" | head -1); python tools/trace_forward.py $db > $out/forward_trace.md 2>&1; rm -rf $out/prof2; tail -1 $out/forward_trace.md
