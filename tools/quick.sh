#!/bin/bash
# A development gpurun call.   gpurun --timeout 1000 -- 'bash tools/quick.sh TAG "PYTEST TARGETS" "bench args 1" "bench args 2" ...'
# PYTEST TARGETS: e.g. "tests" or "tests/test_gpu_ops.py -k block_tail" ("-" = skip the tests); each further argument is one bench.py line.
tag=${1:-quick}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
if [ "${2:--}" != "-" ]; then
  timeout 900 python -m pytest $2 -m gpu -q -rs -x --durations=5 > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -12 $out/pytest.log
fi
shift; shift
for cfg in "$@"; do
  t=$(echo "$cfg" | tr -d ' -' | cut -c1-30)
  timeout 400 python bench.py $cfg --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err; echo "exit $?" >> $out/bench_$t.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/bench_$t.json').read().strip().splitlines()[-1]); p=d.get('parity',{}); r=d.get('roofline',{})
    print('$t', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms | parity', p.get('ok'), 'pose', p.get('pose_max_abs'), 'corr', p.get('corr_max_abs'), 'cond', p.get('kabsch_cond_max'), p.get('reason'),
          '| roofline', round(r.get('frac',0),3), r.get('detail',{}).get('launches_per_step'), '| fp32x3', d.get('fp32x3_pairs_per_s'), '| pre', d.get('preprocess',{}).get('pyramid_ms_alone'), '| gemm', d.get('roofline_gemm',{}).get('ms_per_step'), d.get('roofline_gemm',{}).get('frac'))
except Exception as e: print('$t FAILED', e, open('$out/bench_$t.err').read()[-800:])
PY
done
