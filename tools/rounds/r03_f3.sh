#!/bin/bash
out=gpurun_out/r03_f3; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or f16" > $out/pytest_gemm.log 2>&1; tail -2 $out/pytest_gemm.log
timeout 1200 python tools/x3_bench.py --arms "cw2=REGTR_F16_CW4:0" "cw4=REGTR_F16_CW4:1" > $out/x3_f16_cw4.md 2>&1
cat $out/x3_f16_cw4.md
for v in 0 1; do
REGTR_F16_CW4=$v timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-roofline > $out/bench_cw4_$v.json 2> $out/bench_cw4_$v.err; python - $v <<'PY'
import json,sys; d=json.loads(open(f'gpurun_out/r03_f3/bench_cw4_{sys.argv[1]}.json').read().strip().splitlines()[-1]); print('F16_CW4', sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3), 'pose', d['parity']['pose_max_abs'], d['parity']['ok'])
PY
done
