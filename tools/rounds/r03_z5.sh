#!/bin/bash
out=gpurun_out/r03_z5; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or split or x3 or unary or kpconv" > $out/pytest_gemm.log 2>&1; tail -3 $out/pytest_gemm.log
timeout 1200 python tools/x3_bench.py --arms "il0=REGTR_X3_IL:0" "il1=REGTR_X3_IL:1" > $out/x3_il.md 2>&1
cat $out/x3_il.md
for v in 0 1; do
REGTR_X3_IL=$v timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-roofline > $out/bench_il$v.json 2> $out/bench_il$v.err; python - $v <<'PY'
import json,sys; d=json.loads(open(f'gpurun_out/r03_z5/bench_il{sys.argv[1]}.json').read().strip().splitlines()[-1]); print('IL', sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'])
PY
done
