#!/bin/bash
out=gpurun_out/r03_v; mkdir -p $out; export TMPDIR=/tmp
REGTR_VARIANT=mhaprof timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --parity-pairs 0 > $out/mha_prof.txt 2>&1
grep "^mha" $out/mha_prof.txt | sort -u | grep "nk 391" | head -6
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "mha or attention or attn" > $out/pytest_mha.log 2>&1; tail -2 $out/pytest_mha.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_v/bench.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'], d['roofline_secondary']['detail']['avg_launch_us'], d['roofline_secondary']['achieved'])
PY
