#!/bin/bash
# round 4, call e: full GPU suite on the cleaned tree, default bench line (roofline_gemm), end-to-end harness with warm-up
tag=${1:-r04_e}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rs --durations=8 > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -18 $out/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - <<PY
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); p=d['parity']
print('bench', round(d['value'],1), round(d['ms_per_step'],3), 'parity', p['ok'], 'pose', p['pose_max_abs'], 'corr', p['corr_max_abs'], 'peak GiB', d['config']['peak_hbm_allocated_GiB'], 'gather frac', round(d['roofline']['frac'],3))
g=d['roofline_gemm']; print('gemm ms', g['ms_per_step'], 'roofline ms', g['roofline_ms_per_step'], 'frac', g['frac'], g['by_route_ms'])
for r in g['top_shapes']: print('  ', r['route'][:28].ljust(28), r['M'], r['N'], r['K'], 'fold' if r['folded_norm_operand'] else '', 'x%d'%r['launches_per_step'], r['us'], 'us', r['bound'], r['frac'], r['TFLOPs_f32_equiv'], 'TF', r['GBs'], 'GB/s')
PY
E2E="python test.py --benchmark 3DLoMatch --config regtr_amd/conf/3dmatch.yaml --logdir /tmp/e2e_logs --synthetic 1781 --overlap lomatch --materialize /tmp/e2e_data --distinct 128"
timeout 900 $E2E --num_workers 4 > $out/e2e_pth_cold.log 2>&1; grep -E "End to end|loader:|warm-up" $out/e2e_pth_cold.log | tail -3
timeout 600 $E2E --num_workers 4 > $out/e2e_pth_w4.log 2>&1; grep -E "End to end|loader:" $out/e2e_pth_w4.log | tail -2
timeout 600 $E2E --num_workers 4 --cache_dir /tmp/e2e_cache > $out/e2e_npy_build.log 2>&1; grep -E "End to end" $out/e2e_npy_build.log | tail -1
for w in 2 4; do timeout 600 $E2E --num_workers $w --cache_dir /tmp/e2e_cache > $out/e2e_npy_w$w.log 2>&1; grep -E "End to end|loader:|pairs on" $out/e2e_npy_w$w.log | tail -3; done
timeout 600 $E2E --num_workers 0 > $out/e2e_thread.log 2>&1; grep -E "End to end" $out/e2e_thread.log | tail -1
