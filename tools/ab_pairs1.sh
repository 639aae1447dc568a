#!/bin/bash
# same-box A/B of the one-pair forward under the dispatch switches:  bash tools/ab_pairs1.sh TAG
tag=${1:-ab}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; env "$@" python bench.py --pairs 1 --steps 300 --warmup 30 --no-roofline --no-cpu-baseline --parity-pairs 0 --no-strict-f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), 'ms/pair')"; }
{
run default X=1
run no_overlap REGTR_DEV=1 REGTR_SMALL_OVERLAP=0
run no_c_pyramid REGTR_DEV=1 REGTR_ONE_CALL_PYR=0
run no_c_encoder REGTR_DEV=1 REGTR_ONE_CALL_ENC=0
run default_again X=1
run no_range_check_default X=1
} 2>&1 | tee $out/ab.txt
