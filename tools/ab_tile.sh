#!/bin/bash
# small 64 x 64-tile launches on the 2-wave strip kernel vs the tiled kernel (variant build -DREGTR_DEV_ENV=1 reads REGTR_X3_SMALL_STRIP):  bash tools/ab_tile.sh TAG
tag=${1:-ab_tile}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; n=$2; shift; shift; env "$@" python bench.py --pairs $n --steps 300 --warmup 30 --no-roofline --no-cpu-baseline --parity-pairs 2 --no-strict-f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name pairs $n', round(d['ms_per_step'],3), 'ms', 'parity', d['parity']['ok'], d['parity']['pose_max_abs'])"; }
{
for n in 1 3; do
run small_strip $n REGTR_DEV=1 REGTR_VARIANT=devenv
run tiled $n REGTR_DEV=1 REGTR_VARIANT=devenv REGTR_X3_SMALL_STRIP=0
run small_strip $n REGTR_DEV=1 REGTR_VARIANT=devenv
run tiled $n REGTR_DEV=1 REGTR_VARIANT=devenv REGTR_X3_SMALL_STRIP=0
done
run small_strip_fp32x3 1 REGTR_DEV=1 REGTR_VARIANT=devenv
} 2>&1 | tee $out/ab.txt
