#!/bin/bash
# full GPU suite + default bench line with the new GEMM; reduced-precision parity table (bf16x2 candidate for the cross-encoder linears)
out=gpurun_out/r03_n; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench exit $?"; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_n/bench.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'], d['parity']['corr_max_abs'], d['roofline']['frac'], d['roofline']['detail']['gemm_s_per_step'])
PY
timeout 600 python tools/dtype_parity.py > $out/dtype_parity.txt 2>&1; cat $out/dtype_parity.txt
timeout 300 python bench.py --no-cpu-baseline --dtype bf16x2 --steps 10 > $out/bench_bf16x2.json 2> $out/bench_bf16x2.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_n/bench_bf16x2.json').read().strip().splitlines()[-1]); print('bf16x2', round(d['value'],1), round(d['ms_per_step'],3), d['parity'], d.get('reduced_precision_error'))
PY
