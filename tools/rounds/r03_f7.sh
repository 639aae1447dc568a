#!/bin/bash
out=gpurun_out/r03_f7; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or kpconv" > $out/pytest_ops.log 2>&1; tail -2 $out/pytest_ops.log
for v in 0 1; do
REGTR_F16_THIN=$v timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-roofline > $out/bench_thin$v.json 2> $out/bench_thin$v.err; python - $v <<'PY'
import json,sys; d=json.loads(open(f'gpurun_out/r03_f7/bench_thin{sys.argv[1]}.json').read().strip().splitlines()[-1]); print('THIN', sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3), 'pose', d['parity']['pose_max_abs'], d['parity']['corr_max_abs'], d['parity']['ok'])
PY
done
timeout 300 rocprofv3 --kernel-trace -d $out/prof2 -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --parity-pairs 0 > $out/prof2.log 2>&1
db=$(find $out/prof2 -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats.md 2>&1; rm -rf $out/prof2; grep -E "x3d<4, 2, 2, (true|false), 2, 1>|gemm_f32|instnorm_partial" $out/kernel_stats.md
