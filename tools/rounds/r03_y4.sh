#!/bin/bash
out=gpurun_out/r03_y4; mkdir -p $out; export TMPDIR=/tmp
for v in "" prio "" prio; do
REGTR_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --parity-pairs 0 > $out/bench_v$v.json 2> $out/bench_v$v.err; python - "$v" <<'PY'
import json,sys; d=json.loads(open(f'gpurun_out/r03_y4/bench_v{sys.argv[1]}.json').read().strip().splitlines()[-1]); print('variant', sys.argv[1] or 'default', round(d['value'],1), round(d['ms_per_step'],3), 'mha us', round(d['roofline_secondary']['detail']['avg_launch_us'],1))
PY
done
for p in 96 128 192 256; do
timeout 400 python bench.py --no-cpu-baseline --no-roofline --pairs $p --steps 6 --warmup 2 > $out/bench_p$p.json 2> $out/bench_p$p.err; python - $p <<'PY'
import json,sys; d=json.loads(open(f'gpurun_out/r03_y4/bench_p{sys.argv[1]}.json').read().strip().splitlines()[-1]); print('pairs', sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3), d['parity']['pose_max_abs'], d['config']['peak_hbm_allocated_GiB'])
PY
done
