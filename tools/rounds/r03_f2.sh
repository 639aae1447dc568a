#!/bin/bash
out=gpurun_out/r03_f2; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > $out/pytest_ops.log 2>&1; tail -3 $out/pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench_batch.py -m gpu -x -q -s > $out/pytest_model.log 2>&1; grep -E "bench batch|lomatch pairs|passed|failed" $out/pytest_model.log | cut -c1-330
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_f2/bench.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'], d['parity']['corr_max_abs'], d['parity']['ok'])
PY
timeout 600 python bench.py --no-cpu-baseline --config modelnet --steps 10 --warmup 2 > $out/bench_mn.json 2> $out/bench_mn.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_f2/bench_mn.json').read().strip().splitlines()[-1]); print('modelnet bf16', round(d['value'],1), round(d['ms_per_step'],3), d.get('reduced_precision_error'))
PY
timeout 600 python bench.py --no-cpu-baseline --config lomatch --total-pairs 1781 --steps 2 --warmup 1 --no-roofline > $out/bench_lo.json 2> $out/bench_lo.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_f2/bench_lo.json').read().strip().splitlines()[-1]); print('lomatch', round(d['value'],1), d['parity']['pose_max_abs'], d['parity']['corr_max_abs'], d['parity']['ok'], d['parity']['pose_gate'][:12])
PY
