#!/bin/bash
# every pair of a benchmarked forward against the CPU oracle (bench.py's gate normally checks 8 of them): synthetic, real fragments, shuffled rows,
# ModelNet-size in fp32, and the small-batch regime
out=gpurun_out/${1:-r05_sweep}; mkdir -p $out
common="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-strict-f32"
run() { name=$1; shift; timeout 900 python bench.py $common "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$?"; python - $out/$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('  value', round(d['value'], 1), 'parity', json.dumps(d['parity'])[:600])
except Exception as e:
    print('  no line:', e)
PY
}
run synth64 --parity-pairs 64
run real64 --real --parity-pairs 64
run shuffle32 --shuffle --parity-pairs 32
run modelnet_fp32_64 --config modelnet --dtype fp32 --parity-pairs 64
run pairs1 --pairs 1 --parity-pairs 1
run pairs3_real --pairs 3 --real --parity-pairs 3
run pairs4_real --pairs 4 --real --parity-pairs 4
tail -3 $out/*.err | grep -v "^$" | tail -20
