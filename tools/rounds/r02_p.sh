#!/bin/bash
out=gpurun_out/r02_p; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "mha or forward_vs or batched or determinism or compute_dtype or preprocess or kitchen" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for cfg in "--pairs 1 --steps 50 --warmup 5" "--pairs 4 --steps 30 --warmup 5" "--steps 10 --warmup 2"; do
  tag=$(echo $cfg | tr -d ' -' | cut -c1-20)
  timeout 300 python bench.py $cfg --no-cpu-baseline --no-roofline > $out/bench_$tag.json 2> $out/bench_$tag.err; python - <<PY
import json; d=json.loads(open('$out/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value'],1), round(d['ms_per_step'],3))
PY
done
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py --pairs 1 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_pairs1.md 2>&1; rm -rf $out/prof; head -12 $out/kernel_stats_pairs1.md; tail -1 $out/kernel_stats_pairs1.md
