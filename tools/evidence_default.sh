#!/bin/bash
# The default bench line's own evidence after the default went from 64 to 192 pairs per forward (round 5): counter passes behind roofline.traffic
# on the default workload, the default line, the 64-pair line on the same box, the real-fragment line, and the rocprofv3 kernel statistics +
# forward trace of the default workload.   gpurun --timeout 1500 -- 'bash tools/evidence_default.sh r05_w'
tag=${1:-r05_w}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python bench.py --collect-pmc --pmc-tag $tag > $out/collect_pmc.log 2>&1; tail -2 $out/collect_pmc.log | cut -c1-300
cp profiles/pmc_traffic.json $out/pmc_traffic.json; cp profiles/${tag}_pmc_traffic_kernels.md $out/ 2>/dev/null
line() { python - $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p = d['parity']; r = d.get('roofline', {})
    print(sys.argv[1].split('/')[-1], round(d['value'], 1), round(d['ms_per_step'], 3), 'parity', p.get('ok'), p.get('pose_max_abs'), p.get('corr_max_abs'), 'gather frac', r.get('frac'), 'traffic', r.get('traffic'),
          'counter frac', d.get('counter_hbm_frac_of_peak'), 'fp32x3', d.get('fp32x3_pairs_per_s'), 'GiB', d['config'].get('peak_hbm_allocated_GiB'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; line $out/bench.json
timeout 600 python bench.py --pairs 64 --no-cpu-baseline > $out/bench_pairs64.json 2> $out/bench_pairs64.err; line $out/bench_pairs64.json
timeout 600 python bench.py --real --steps 10 --warmup 3 --no-cpu-baseline --no-strict-f32 > $out/bench_real.json 2> $out/bench_real.err; line $out/bench_real.json
timeout 600 python bench.py --shuffle --steps 10 --warmup 2 --no-cpu-baseline --no-strict-f32 > $out/bench_shuffle.json 2> $out/bench_shuffle.err; line $out/bench_shuffle.json
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --parity-pairs 0 --no-strict-f32 > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats.md 2>&1; python tools/trace_forward.py $db > $out/forward_trace.md 2>&1; rm -rf $out/prof; head -8 $out/kernel_stats.md; tail -1 $out/forward_trace.md
