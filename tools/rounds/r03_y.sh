#!/bin/bash
out=gpurun_out/r03_y; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench_batch.py -m gpu -x -q > $out/pytest_model.log 2>&1; tail -3 $out/pytest_model.log
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" > $out/pytest_ops.log 2>&1; tail -2 $out/pytest_ops.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r03_y/bench.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'], d['roofline_secondary']['detail']['avg_launch_us'])
PY
for v in 1 0; do
REGTR_ONE_CALL_XENC=$v timeout 300 python bench.py --no-cpu-baseline --pairs 1 --steps 200 --warmup 20 --no-roofline > $out/bench_p1_$v.json 2> $out/bench_p1_$v.err; python - $v <<'PY'
import json,sys; d=json.loads(open(f'gpurun_out/r03_y/bench_p1_{sys.argv[1]}.json').read().strip().splitlines()[-1]); print('pairs 1, one-call', sys.argv[1], ':', round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'])
PY
done
timeout 300 python tools/host_profile.py --pairs 1 > $out/host_profile_p1.txt 2>&1; head -3 $out/host_profile_p1.txt
timeout 300 rocprofv3 --kernel-trace -d $out/prof2 -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --parity-pairs 0 > $out/prof2.log 2>&1
db=$(find $out/prof2 -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats.md 2>&1; rm -rf $out/prof2; grep -E "gemm_f32" $out/kernel_stats.md
