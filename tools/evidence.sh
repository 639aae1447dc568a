#!/bin/bash
# The evidence run of a round, ONE gpurun call: every GPU test, the counter passes behind roofline.traffic / forward_traffic (bench.py --collect-pmc),
# the default bench line (with CPU baseline, fp32x3 and real-fragment side measurements), kernel statistics + per-forward trace (rocprofv3), the
# MFMA / wave-cycle PMC table and the radius kernels' PMC table, every other bench configuration (64 pairs, real fragments, one / three / eight pairs
# per forward, one forward at a time at 64 / 192 pairs, shuffle, stress, ModelNet x3, parity mode, 3DLoMatch set, strong-scaling predictions for
# 2 / 4 / 8 ranks, fp32x3), the one-pair
# kernel statistics / trace / host profile, and the end-to-end harness (192 and 64 per forward, warm and cold start).
# Outputs under gpurun_out/<tag>; tools/collect_evidence.py copies the summaries into profiles/<tag>_*.
#   gpurun --timeout 2700 -- 'bash tools/evidence.sh r06_x'
tag=${1:-evidence}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rs --durations=5 > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; tail -6 $out/pytest.log
timeout 900 python bench.py --collect-pmc --pmc-tag $tag > $out/collect_pmc.log 2>&1; tail -2 $out/collect_pmc.log | cut -c1-400
cp profiles/pmc_traffic.json $out/pmc_traffic.json; cp profiles/${tag}_pmc_traffic_kernels.md $out/ 2>/dev/null
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; python - <<PY
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); p=d['parity']; r=d['roofline']; ft=d['forward_traffic']
print('bench', round(d['value'],1), round(d['ms_per_step'],3), 'parity', p['ok'], p['pose_max_abs'], p['corr_max_abs'], 'gather frac', round(r['frac'],3), 'traffic', r['traffic'], r['detail'].get('traffic_note'), 'counter frac', d.get('counter_hbm_frac_of_peak'),
      'forward: roofline frac', d['forward_roofline_frac'], 'traffic', ft['hbm_GB'], 'compulsory', ft['compulsory_GB'], 'ratio', ft['ratio'], ft.get('note'), 'gemm', d['roofline_gemm']['ms_per_step'], d['roofline_gemm']['frac'],
      'cpu', d['cpu_baseline']['value'], 'fp32x3', d.get('fp32x3_pairs_per_s'), 'real', d.get('real_fragments_pairs_per_s'), d.get('real_fragments',{}).get('parity'), 'pre', d.get('preprocess',{}).get('pyramid_ms_alone'))
PY
prof() { n=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py "$@" --settle-s 0 --no-cpu-baseline --no-roofline --parity-pairs 0 --no-strict-f32 --no-real > $out/prof$n.log 2>&1
  db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats$n.md 2>&1; python tools/trace_forward.py $db > $out/forward_trace$n.md 2>&1; rm -rf $out/prof; head -8 $out/kernel_stats$n.md; tail -1 $out/forward_trace$n.md; }
prof "" --replicas 1 --steps 8 --warmup 2
prof _concurrent --steps 5 --warmup 2
prof _pairs192 --replicas 1 --pairs 192 --steps 4 --warmup 2
prof _pairs1 --pairs 1 --steps 50 --warmup 10
prof _real --real --replicas 1 --pairs 192 --steps 4 --warmup 2
pmc() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_$n -o p -- python bench.py --replicas 1 --pairs 192 --steps 2 --warmup 1 --settle-s 0 --no-cpu-baseline --no-roofline --parity-pairs 0 --no-strict-f32 --no-real > $out/pmc_$n.log 2>&1 || tail -3 $out/pmc_$n.log; }
pmc 3 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
python tools/pmc_summary.py --mfma $out > $out/pmc_mfma.md 2>&1; head -14 $out/pmc_mfma.md; find $out -name "*.csv" -size +4M -delete
for cfg in "--replicas 1 --steps 40 --warmup 5" "--pairs 192 --replicas 1 --steps 20 --warmup 3" "--pairs 192 --replicas 2 --steps 10 --warmup 3" "--real --steps 10 --warmup 3" "--real --replicas 1 --pairs 192 --steps 10 --warmup 3" "--pairs 1 --steps 300 --warmup 30" "--pairs 3 --steps 100 --warmup 10" "--pairs 8 --steps 50 --warmup 5" "--shuffle --steps 10 --warmup 2" "--points 100000 --pairs 8 --steps 6 --warmup 2" "--config modelnet --steps 10 --warmup 2" "--config modelnet --pairs 256 --replicas 1 --steps 10 --warmup 2" "--config modelnet --dtype fp32 --steps 10 --warmup 2" "--parity-mode --pairs 64 --steps 5 --warmup 2 --no-roofline" "--config lomatch --total-pairs 1781 --steps 3 --warmup 1" "--config lomatch --total-pairs 1781 --emulate-rank-of 2 --steps 5 --warmup 2 --no-roofline" "--config lomatch --total-pairs 1781 --emulate-rank-of 4 --steps 8 --warmup 2 --no-roofline" "--config lomatch --total-pairs 1781 --emulate-rank-of 8 --steps 12 --warmup 3 --no-roofline" "--dtype fp32x3 --steps 10 --warmup 2"; do
  t=$(echo $cfg | tr -d ' -' | cut -c1-44)
  timeout 600 python bench.py $cfg --no-cpu-baseline --no-strict-f32 --no-real > $out/bench_$t.json 2> $out/bench_$t.err; echo "exit $?" >> $out/bench_$t.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/bench_$t.json').read().strip().splitlines()[-1]); p=d.get('parity',{}); print('$t', round(d['value'],1), round(d['ms_per_step'],3), 'pose', p.get('pose_max_abs'), 'corr', p.get('corr_max_abs'), p.get('ok'), p.get('reason'), 'cond', p.get('kabsch_cond_max'), d['config'].get('peak_hbm_allocated_GiB'), round(d['roofline']['frac'],3) if 'roofline' in d else '', d.get('preprocess',{}).get('pyramid_ms_alone'), (d.get('predicted') or {}).get('predicted_pairs_per_s'))
except Exception as e: print('$t FAILED', e, open('$out/bench_$t.err').read()[-600:])
PY
done
timeout 300 python tools/host_profile.py 1 > $out/host_profile_p1.txt 2>&1; head -2 $out/host_profile_p1.txt | tail -1
E2E="python test.py --benchmark 3DLoMatch --config regtr_amd/conf/3dmatch.yaml --logdir /tmp/e2e_logs --synthetic 1781 --overlap lomatch --materialize /tmp/e2e_data --distinct 128"
timeout 900 $E2E --cache_dir /tmp/e2e_cache > $out/e2e_build.log 2>&1; grep -E "End to end" $out/e2e_build.log | tail -1
timeout 600 $E2E --cache_dir /tmp/e2e_cache > $out/e2e_npy.log 2>&1; grep -E "End to end" $out/e2e_npy.log | tail -1
timeout 600 $E2E > $out/e2e_pth.log 2>&1; grep -E "End to end" $out/e2e_pth.log | tail -1
timeout 600 $E2E --cache_dir /tmp/e2e_cache --batch 64 > $out/e2e_npy_batch64.log 2>&1; grep -E "End to end" $out/e2e_npy_batch64.log | tail -1
timeout 600 $E2E --cache_dir /tmp/e2e_cache --no_warmup > $out/e2e_npy_cold.log 2>&1; grep -E "End to end" $out/e2e_npy_cold.log | tail -1
