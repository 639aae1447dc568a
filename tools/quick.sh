#!/bin/bash
# A development gpurun call: GPU tests + a few short bench lines.   gpurun --timeout 1000 -- 'bash tools/quick.sh r05_a'
tag=${1:-quick}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -rs -x --durations=5 > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -8 $out/pytest.log
run() { t=$(echo "$*" | tr -d ' -' | cut -c1-30); timeout 400 python bench.py "$@" --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err; echo "exit $?" >> $out/bench_$t.err
python - <<PY
import json
try:
    d=json.loads(open('$out/bench_$t.json').read().strip().splitlines()[-1]); p=d.get('parity',{}); r=d.get('roofline',{})
    print('$t', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms | parity', p.get('ok'), 'pose', p.get('pose_max_abs'), 'corr', p.get('corr_max_abs'), 'cond', p.get('kabsch_cond_max'), p.get('reason'),
          '| roofline', round(r.get('frac',0),3), r.get('detail',{}).get('launches_per_step'), '| fp32x3', d.get('fp32x3_pairs_per_s'), '| pre', d.get('preprocess',{}).get('pyramid_ms_alone'), '| gemm', d.get('roofline_gemm',{}).get('ms_per_step'), d.get('roofline_gemm',{}).get('frac'))
except Exception as e: print('$t FAILED', e, open('$out/bench_$t.err').read()[-800:])
PY
}
shift
if [ $# -eq 0 ]; then
run --steps 10 --warmup 3
run --real --steps 10 --warmup 3
run --config modelnet --dtype fp32 --steps 10 --warmup 2
run --config modelnet --steps 10 --warmup 2
run --pairs 1 --steps 200 --warmup 20 --no-roofline
run --pairs 8 --steps 50 --warmup 5 --no-roofline
fi
