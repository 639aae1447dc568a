#!/bin/bash
# where does the small-batch regime (one-call encoder, small-batch kernel forms) stop paying?  bash tools/ab_small.sh TAG
tag=${1:-ab_small}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; n=$2; shift; shift; env "$@" python bench.py --pairs $n --steps 100 --warmup 10 --no-roofline --no-cpu-baseline --parity-pairs 0 --no-strict-f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name pairs $n', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'pairs/s')"; }
{
for n in 2 3 4 8; do
run default $n X=1
run small_regime $n REGTR_DEV=1 REGTR_SMALL_ROWS=4000000
done
run default 16 X=1
run small_regime 16 REGTR_DEV=1 REGTR_SMALL_ROWS=4000000
} 2>&1 | tee $out/ab.txt
