#!/bin/bash
# round 4, first GPU call: all GPU tests (new: floor voxel keys, demo goldens, range fallback, threads, nccl), default bench line
# (8 parity pairs, fp32x3 side number), one-pair line (cost of the end-of-forward status read), kernel stats
tag=${1:-r04_a}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --durations=15 > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -25 $out/pytest.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; python - <<PY
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); p=d['parity']
print('bench', round(d['value'],1), round(d['ms_per_step'],3), 'parity ok', p['ok'], p.get('reason'), 'pose', p['pose_max_abs'], 'corr', p['corr_max_abs'], 'pairs', p['pairs_checked'])
print([ (x['slot'], '%.1e'%x['pose'], x['cond']) for x in p['per_pair']])
print('fp32x3', d['config'].get('fp32x3_same_workload')); print('roofline', d['roofline']['frac'], d['roofline']['detail']['avg_launch_us'])
PY
for cfg in "--pairs 1 --steps 200 --warmup 20" ; do
  t=$(echo $cfg | tr -d ' -' | cut -c1-26)
  timeout 600 python bench.py $cfg --no-cpu-baseline --no-strict-f32 > $out/bench_$t.json 2> $out/bench_$t.err
  python -c "import json; d=json.loads(open('$out/bench_$t.json').read().strip().splitlines()[-1]); print('$t', round(d['value'],1), round(d['ms_per_step'],3), d['parity']['ok'], d['parity']['pose_max_abs'])"
done
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --parity-pairs 0 --no-strict-f32 > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats.md 2>&1; rm -f $db; head -40 $out/kernel_stats.md
