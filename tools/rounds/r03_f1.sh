#!/bin/bash
out=gpurun_out/r03_f1; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -s -k "f16_pair" > $out/pytest_f16.log 2>&1; grep -E "max rel err|passed|failed|Error|error" $out/pytest_f16.log | head -20
timeout 1200 python tools/x3_bench.py --arms "base=REGTR_F16_PAIR:0" "f16=REGTR_F16_PAIR:1" > $out/x3_f16.md 2>&1
cat $out/x3_f16.md
for v in 0 1; do
REGTR_F16_PAIR=$v timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-roofline > $out/bench_f16_$v.json 2> $out/bench_f16_$v.err; python - $v <<'PY'
import json,sys; d=json.loads(open(f'gpurun_out/r03_f1/bench_f16_{sys.argv[1]}.json').read().strip().splitlines()[-1]); print('F16_PAIR', sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3), 'pose', d['parity']['pose_max_abs'], 'corr', d['parity']['corr_max_abs'], d['parity']['ok'])
PY
done
