#!/bin/bash
out=gpurun_out/${1:-r02_m}; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -2 $out/pytest.log
python tools/preprocess_bench.py 2>&1 | grep preprocess
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - <<PY
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('3dmatch', d['value'], d['ms_per_step'], 'gather frac', d['roofline']['frac'], d['roofline']['detail']['gather_s_per_step'])
PY
