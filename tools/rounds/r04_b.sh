#!/bin/bash
tag=${1:-r04_b}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_range.py tests/test_gpu_stress.py tests/test_gpu_nccl.py -m gpu -q -rs --durations=8 > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -40 $out/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - <<PY
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); p=d['parity']
print('bench', round(d['value'],1), round(d['ms_per_step'],3), 'parity ok', p['ok'], p.get('reason'), 'pose', p['pose_max_abs'], 'corr', p['corr_max_abs'], 'pairs', p['pairs_checked'], 'cond', p['kabsch_cond_max'])
print([ (x['slot'], '%.1e'%x['pose'], x['cond']) for x in p['per_pair']])
PY
timeout 600 python bench.py --no-cpu-baseline --config lomatch --total-pairs 1781 --steps 2 --warmup 1 --no-strict-f32 > $out/bench_lomatch.json 2> $out/bench_lomatch.err; python - <<PY
import json
d=json.loads(open('$out/bench_lomatch.json').read().strip().splitlines()[-1]); p=d['parity']
print('lomatch', round(d['value'],1), round(d['ms_per_step'],3), 'parity ok', p['ok'], p.get('reason'), 'pose', p['pose_max_abs'], 'corr', p['corr_max_abs'], 'cond', p['kabsch_cond_max'])
PY
