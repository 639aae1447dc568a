out=gpurun_out/r06_e; mkdir -p $out; export TMPDIR=/tmp
python tools/stream_bench.py --r6 2>&1 | tail -6 | tee $out/stream_bench.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bench_batch.py tests/test_gpu_model.py -m gpu -q -x -k "not all_pairs" 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --no-strict-f32 --no-real > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_e/bench.json').read().strip().splitlines()[-1]); p=d['parity']
print('bench', round(d['value'],1), round(d['ms_per_step'],2), 'parity', p['ok'], p['pose_max_abs'], p['corr_max_abs'], p['kabsch_cond_max'], 'pre', d['preprocess']['pyramid_ms_alone'], 'gemm', d['roofline_gemm']['ms_per_step'], d['roofline_gemm']['frac'], d['roofline_gemm']['by_route_ms'])
for r in d['roofline_gemm']['top_shapes']:
    if r['route'].startswith('one-shot'): print(r)
PY
