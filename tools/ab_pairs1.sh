#!/bin/bash
# same-box A/B of the one-pair forward under the dispatch switches, interleaved repeats:  bash tools/ab_pairs1.sh TAG [reps]
tag=${1:-ab}; reps=${2:-3}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; env "$@" python bench.py --pairs 1 --steps 500 --warmup 50 --no-roofline --no-cpu-baseline --parity-pairs 0 --no-strict-f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), 'ms/pair')"; }
{
for i in $(seq $reps); do
run default X=1
run c_pyramid_no_overlap REGTR_DEV=1 REGTR_SMALL_OVERLAP=0
run python_pyramid REGTR_DEV=1 REGTR_ONE_CALL_PYR=0
done
} 2>&1 | tee $out/ab.txt
