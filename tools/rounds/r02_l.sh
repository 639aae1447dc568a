#!/bin/bash
out=gpurun_out/${1:-r02_l}; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "kpconv or maxpool or forward_vs or kitchen or cpp_wrappers or batched" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python - <<PY
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('3dmatch', d['value'], d['ms_per_step'], 'gather frac', d['roofline']['frac'], d['roofline']['detail']['gather_s_per_step'])
PY
timeout 600 python bench.py --steps 10 --warmup 2 --shuffle --no-cpu-baseline > $out/bench_shuffle.json 2> $out/bench_shuffle.err; python - <<PY
import json; d=json.loads(open('$out/bench_shuffle.json').read().strip().splitlines()[-1]); print('3dmatch shuffle', d['value'], d['ms_per_step'], 'gather frac', d['roofline']['frac'])
PY
timeout 300 rocprofv3 --kernel-trace -d $out/prof -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/trace_forward.py $db > $out/forward_trace.md 2>&1; python tools/rocpd_stats.py $db > $out/kernel_stats.md; rm -rf $out/prof; tail -1 $out/forward_trace.md; grep -E "gather|maxpool" $out/forward_trace.md
