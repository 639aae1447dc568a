#!/bin/bash
out=gpurun_out/r03_z3; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "mha or attn or attention" > $out/pytest_att.log 2>&1; tail -3 $out/pytest_att.log
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q > $out/pytest_model.log 2>&1; tail -3 $out/pytest_model.log
timeout 300 python bench.py --no-cpu-baseline --pairs 1 --steps 200 --warmup 20 --no-roofline > $out/bench_p1.json 2> $out/bench_p1.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_z3/bench_p1.json').read().strip().splitlines()[-1]); print('pairs 1:', round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'])
PY
timeout 300 python bench.py --no-cpu-baseline --pairs 2 --steps 100 --warmup 10 --no-roofline > $out/bench_p2.json 2> $out/bench_p2.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_z3/bench_p2.json').read().strip().splitlines()[-1]); print('pairs 2:', round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'])
PY
timeout 300 rocprofv3 --kernel-trace -d $out/prof3 -o trace -- python bench.py --pairs 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --parity-pairs 0 > $out/prof3.log 2>&1
db=$(find $out/prof3 -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_p1.md 2>&1; rm -rf $out/prof3; head -12 $out/kernel_stats_p1.md
