#!/bin/bash
out=gpurun_out/r03_f5; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > $out/pytest_ops.log 2>&1; tail -2 $out/pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench_batch.py -m gpu -x -q -s > $out/pytest_model.log 2>&1; grep -E "max abs diff vs the reference|f16 pair:|passed|failed" $out/pytest_model.log | cut -c1-260
timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_f5/bench.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'], d['parity']['corr_max_abs'], d['parity']['ok'])
PY
timeout 300 python bench.py --no-cpu-baseline --pairs 1 --steps 200 --warmup 20 --no-roofline > $out/bench_p1.json 2> $out/bench_p1.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_f5/bench_p1.json').read().strip().splitlines()[-1]); print('pairs 1:', round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'])
PY
