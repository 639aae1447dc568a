out=gpurun_out/r06_c; mkdir -p $out; export TMPDIR=/tmp
python tools/probe/cmp_run.py > $out/cmp_probe.txt 2>&1; cat $out/cmp_probe.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stress.py tests/test_gpu_bench_batch.py -m gpu -q -x -k "not all_pairs" 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline --no-strict-f32 --no-real > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_c/bench.json').read().strip().splitlines()[-1]); p=d['parity']
print('bench', round(d['value'],1), round(d['ms_per_step'],2), 'parity', p['ok'], p['pose_max_abs'], p['corr_max_abs'], p['kabsch_cond_max'], 'pre', d['preprocess'], 'frac', d['forward_roofline_frac'])
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py --steps 5 --warmup 2 --settle-s 0 --no-cpu-baseline --no-roofline --parity-pairs 0 --no-strict-f32 --no-real > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats.md 2>&1; rm -rf $out/prof; grep -E "radius|insert|scatter|pack_slots|table_keys|clear" $out/kernel_stats.md
timeout 600 python tools/concurrency_probe.py 192 10 > $out/concurrency.txt 2>&1; tail -6 $out/concurrency.txt
