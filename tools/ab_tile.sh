#!/bin/bash
# the three-deep operand pipeline of small 64 x 64-tile products (k_gemm_x3q) against the tiled kernel, same box, alternating
# (variant build -DREGTR_DEV_ENV=1 reads REGTR_X3_DEEP):  bash tools/ab_tile.sh TAG
tag=${1:-ab_tile}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; n=$2; shift; shift; env "$@" python bench.py --pairs $n --steps $((n >= 8 ? 30 : 300)) --warmup $((n >= 8 ? 5 : 30)) --no-roofline --no-cpu-baseline --parity-pairs 2 --no-strict-f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name pairs $n', round(d['ms_per_step'],3), 'ms', 'parity', d['parity']['ok'], d['parity']['pose_max_abs'], d['parity']['corr_max_abs'])"; }
{
for n in 1 3; do
run deep $n REGTR_DEV=1 REGTR_VARIANT=devenv
run tiled $n REGTR_DEV=1 REGTR_VARIANT=devenv REGTR_X3_DEEP=0
run deep $n REGTR_DEV=1 REGTR_VARIANT=devenv
run tiled $n REGTR_DEV=1 REGTR_VARIANT=devenv REGTR_X3_DEEP=0
done
run deep 2 REGTR_DEV=1 REGTR_VARIANT=devenv
for n in 8 64; do
run deep $n REGTR_DEV=1 REGTR_VARIANT=devenv
run tiled $n REGTR_DEV=1 REGTR_VARIANT=devenv REGTR_X3_DEEP=0
done
} 2>&1 | tee $out/ab.txt
