#!/bin/bash
# same-box A/B of the one-pair forward under a dispatch switch, interleaved repeats:  bash tools/ab_pairs1.sh TAG REPS "NAME=VALUE ..."
tag=${1:-ab}; reps=${2:-3}; sw=${3:-REGTR_EARLY_L0=0}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; n=$2; shift; shift; env "$@" python bench.py --pairs $n --steps 500 --warmup 50 --no-roofline --no-cpu-baseline --parity-pairs 0 --no-strict-f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name pairs $n', round(d['ms_per_step'],3), 'ms')"; }
{
for i in $(seq $reps); do
run default 1 X=1
run "$sw" 1 REGTR_DEV=1 $sw
done
run default 3 X=1
run "$sw" 3 REGTR_DEV=1 $sw
} 2>&1 | tee $out/ab.txt
