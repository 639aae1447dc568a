#!/bin/bash
out=gpurun_out/r03_last2; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench_batch.py tests/test_harness.py -m gpu -x -q > $out/pytest.log 2>&1; tail -2 $out/pytest.log
timeout 300 python bench.py --no-cpu-baseline --pairs 1 --steps 300 --warmup 30 --no-roofline > $out/bench_p1.json 2> $out/bench_p1.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_last2/bench_p1.json').read().strip().splitlines()[-1]); print('pairs 1:', round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'])
PY
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_last2/bench.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'], d['parity']['ok'])
PY
