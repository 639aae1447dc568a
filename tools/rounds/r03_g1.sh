#!/bin/bash
out=gpurun_out/r03_g1; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -s -k "gather_f16" > $out/pytest_g.log 2>&1; grep -E "f16 gather|passed|failed|Error|error|assert" $out/pytest_g.log | head -20
for v in 0 1; do
REGTR_F16_GATHER=$v timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/bench_g$v.json 2> $out/bench_g$v.err; python - $v <<'PY'
import json,sys
try:
    d=json.loads(open(f'gpurun_out/r03_g1/bench_g{sys.argv[1]}.json').read().strip().splitlines()[-1]); print('F16_GATHER', sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3), 'pose', d['parity']['pose_max_abs'], d['parity']['corr_max_abs'], d['parity']['ok'], 'gather us', round(d['roofline']['detail']['avg_launch_us'],1))
except Exception as e: print('bench failed', sys.argv[1], e); print(open(f'gpurun_out/r03_g1/bench_g{sys.argv[1]}.err').read()[-600:])
PY
done
