#!/bin/bash
# final round-3 evidence: gpu_round (all GPU tests, default bench line with CPU baseline + parity, kernel stats, PMC traffic + MFMA) and every other bench line
tag=${1:-r03_z}
BENCH_ARGS="" bash tools/gpu_round.sh $tag
out=gpurun_out/$tag; export TMPDIR=/tmp
for cfg in "--pairs 1 --steps 200 --warmup 20" "--shuffle --steps 10 --warmup 2" "--points 100000 --pairs 8 --steps 6 --warmup 2" "--config modelnet --steps 10 --warmup 2" "--config modelnet --dtype fp32 --steps 10 --warmup 2" "--parity-mode --steps 5 --warmup 2 --no-roofline" "--config lomatch --total-pairs 1781 --steps 3 --warmup 1" "--dtype fp32x3 --steps 10 --warmup 2"; do
  t=$(echo $cfg | tr -d ' -' | cut -c1-26)
  timeout 600 python bench.py $cfg --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err
  python - <<PY
import json; d=json.loads(open('$out/bench_$t.json').read().strip().splitlines()[-1]); p=d.get('parity',{}); print('$t', round(d['value'],1), round(d['ms_per_step'],3), 'pose', p.get('pose_max_abs'), 'corr', p.get('corr_max_abs'), p.get('ok'), d['config'].get('peak_hbm_allocated_GiB'))
PY
done
timeout 300 rocprofv3 --kernel-trace -d $out/prof2 -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --parity-pairs 0 > $out/prof2.log 2>&1
db=$(find $out/prof2 -name "*.db" | head -1); python tools/trace_forward.py $db > $out/forward_trace.md 2>&1; rm -rf $out/prof2; tail -1 $out/forward_trace.md
timeout 300 rocprofv3 --kernel-trace -d $out/prof3 -o trace -- python bench.py --pairs 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --parity-pairs 0 > $out/prof3.log 2>&1
db=$(find $out/prof3 -name "*.db" | head -1); python tools/trace_forward.py $db > $out/forward_trace_p1.md 2>&1; python tools/rocpd_stats.py $db > $out/kernel_stats_p1.md 2>&1; rm -rf $out/prof3; tail -1 $out/forward_trace_p1.md
timeout 300 python tools/host_profile.py 1 > $out/host_profile_p1.txt 2>&1; head -2 $out/host_profile_p1.txt | tail -1
timeout 600 python tools/dtype_parity.py > $out/dtype_parity.txt 2>&1; grep -c max $out/dtype_parity.txt
# end-to-end harness: 1781 lomatch-like pairs as .pth files -> loader thread -> H2D -> forward -> gather -> est.log (second run: page cache warm)
timeout 900 python test.py --benchmark 3DLoMatch --config regtr_amd/conf/3dmatch.yaml --logdir /tmp/e2e_logs --synthetic 1781 --overlap lomatch --materialize /tmp/e2e_data > $out/e2e.log 2>&1; grep -E "End to end" $out/e2e.log | tail -1
timeout 600 python test.py --benchmark 3DLoMatch --config regtr_amd/conf/3dmatch.yaml --logdir /tmp/e2e_logs --synthetic 1781 --overlap lomatch --materialize /tmp/e2e_data > $out/e2e_warm.log 2>&1; grep -E "End to end" $out/e2e_warm.log | tail -1
