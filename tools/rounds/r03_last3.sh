#!/bin/bash
out=gpurun_out/r03_last3; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -2 $out/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_last3/bench.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'], d['parity']['ok'])
PY
