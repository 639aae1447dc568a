#!/bin/bash
out=gpurun_out/r03_s; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "block_tail" > $out/pytest_tail.log 2>&1; tail -5 $out/pytest_tail.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench_batch.py tests/test_gpu_stress.py -m gpu -x -q > $out/pytest_model.log 2>&1; tail -3 $out/pytest_model.log
for v in 0 1; do
REGTR_BLOCK_TAIL_RES=$v timeout 600 python bench.py --no-cpu-baseline --no-roofline > $out/bench_res$v.json 2> $out/bench_res$v.err; python - <<PY
import json; d=json.loads(open('gpurun_out/r03_s/bench_res$v.json').read().strip().splitlines()[-1]); print('res=$v', round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'], d['parity']['corr_max_abs'])
PY
done
timeout 300 rocprofv3 --kernel-trace -d $out/prof2 -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --parity-pairs 0 > $out/prof2.log 2>&1
db=$(find $out/prof2 -name "*.db" | head -1); python tools/trace_forward.py $db > $out/forward_trace.md 2>&1; rm -rf $out/prof2; tail -1 $out/forward_trace.md; sed -n 120,150p $out/forward_trace.md
