#!/bin/bash
# The last gpurun call of a round, after the last kernel-source change: every GPU test, the counter passes (profiles/pmc_traffic.json is stamped with
# the kernel-source hash, so it must be taken on the final sources) and the default bench line.   gpurun --timeout 1800 -- 'bash tools/final_check.sh r06_z'
tag=${1:-final}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rs --durations=5 > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -5 $out/pytest.log
timeout 900 python bench.py --collect-pmc --pmc-tag $tag > $out/collect_pmc.log 2>&1; tail -1 $out/collect_pmc.log | cut -c1-300
cp profiles/pmc_traffic.json $out/pmc_traffic.json; cp profiles/${tag}_pmc_traffic_kernels.md $out/ 2>/dev/null
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?"; tail -2 $out/bench.err
python - <<PY
import json
d=json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1]); p=d['parity']; r=d['roofline']; ft=d['forward_traffic']
print('bench', round(d['value'],1), round(d['ms_per_step'],3), 'parity', p['ok'], p['pose_max_abs'], p['corr_max_abs'], 'gather', round(r['frac'],3), 'traffic', r['traffic'], r['detail'].get('traffic_note'),
      'fwd frac', d['forward_roofline_frac'], 'fwd traffic', ft['hbm_GB'], ft['compulsory_GB'], ft['ratio'], ft.get('note'), 'fp32x3', d.get('fp32x3_pairs_per_s'), d.get('fp32x3_error'), 'real', d.get('real_fragments_pairs_per_s'), d.get('real_fragments_error'), 'cpu', d['cpu_baseline']['value'])
PY
