#!/bin/bash
out=gpurun_out/r03_z4; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "kpconv or gather or instnorm" > $out/pytest_kp.log 2>&1; tail -3 $out/pytest_kp.log
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q > $out/pytest_model.log 2>&1; tail -3 $out/pytest_model.log
for p in 1 2 8; do
timeout 300 python bench.py --no-cpu-baseline --pairs $p --steps 100 --warmup 20 --no-roofline > $out/bench_p$p.json 2> $out/bench_p$p.err; python - $p <<'PY'
import json,sys; d=json.loads(open(f'gpurun_out/r03_z4/bench_p{sys.argv[1]}.json').read().strip().splitlines()[-1]); print('pairs', sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'])
PY
done
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-roofline > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_z4/bench.json').read().strip().splitlines()[-1]); print('pairs 64', round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'])
PY
timeout 300 rocprofv3 --kernel-trace -d $out/prof3 -o trace -- python bench.py --pairs 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --parity-pairs 0 > $out/prof3.log 2>&1
db=$(find $out/prof3 -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_p1.md 2>&1; python tools/trace_forward.py $db > $out/forward_trace_p1.md 2>&1; rm -rf $out/prof3; head -14 $out/kernel_stats_p1.md; tail -1 $out/forward_trace_p1.md
