#!/bin/bash
out=gpurun_out/r03_f4; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -s -k "mha" > $out/pytest_mha.log 2>&1; grep -E "precision 3|precision 0 lens \[150|passed|failed" $out/pytest_mha.log | head -12
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench_batch.py -m gpu -x -q > $out/pytest_model.log 2>&1; tail -2 $out/pytest_model.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_f4/bench.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'], d['parity']['corr_max_abs'], d['parity']['ok'], 'mha us', round(d['roofline_secondary']['detail']['avg_launch_us'],1), d['roofline_secondary']['detail']['operands'])
PY
REGTR_F16_PAIR=0 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $out/bench_off.json 2> $out/bench_off.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_f4/bench_off.json').read().strip().splitlines()[-1]); print('f16 off', round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'])
PY
