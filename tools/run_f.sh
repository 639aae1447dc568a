out=gpurun_out/r06_f; mkdir -p $out; export TMPDIR=/tmp REGTR_DEV=1
for v in nowide ""; do
  export REGTR_VARIANT=$v
  for a in "192 330 460 3" "192 330 460 0" "256 560 640 1" "64 230 360 3" "64 100 250 3"; do python tools/mha_bench.py $a 2>&1 | tail -1; done
done | tee $out/mha_bench.txt
unset REGTR_VARIANT REGTR_DEV
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-strict-f32 --no-real > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
timeout 600 python bench.py --config modelnet --no-cpu-baseline --steps 10 --warmup 2 > $out/bench_modelnet.json 2> $out/bench_modelnet.err; tail -3 $out/bench_modelnet.err
python - <<'PY'
import json
for f in ('bench','bench_modelnet'):
    d=json.loads(open(f'gpurun_out/r06_f/{f}.json').read().strip().splitlines()[-1]); p=d['parity']
    a = d['roofline_secondary'] if f=='bench' else d['roofline']
    print(f, round(d['value'],1), round(d['ms_per_step'],2), 'parity', p['ok'], p['pose_max_abs'], p['corr_max_abs'], 'attention', round(a['detail']['avg_launch_us'],1), 'us', round(a['frac'],4), d.get('reduced_precision_error'))
PY
