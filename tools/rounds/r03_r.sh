#!/bin/bash
out=gpurun_out/r03_r; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python tools/host_profile.py 1 > $out/host_profile_p1.txt 2>&1; head -24 $out/host_profile_p1.txt
for cfg in "--pairs 1 --steps 100 --warmup 10" "--pairs 2 --steps 60 --warmup 5" "--pairs 8 --steps 30 --warmup 5"; do
  tag=$(echo $cfg | tr -d ' -' | cut -c1-20)
  timeout 300 python bench.py $cfg --no-cpu-baseline --no-roofline > $out/bench_$tag.json 2> $out/bench_$tag.err; python - <<PY
import json; d=json.loads(open('$out/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value'],1), round(d['ms_per_step'],3), d['parity']['ok'])
PY
done
