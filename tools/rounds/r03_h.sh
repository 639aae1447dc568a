#!/bin/bash
out=gpurun_out/r03_h; mkdir -p $out; export TMPDIR=/tmp
arms=""
for t in 0 1 2; do for s in 1 2 3 4 6; do arms="$arms t${t}s${s}=REGTR_X3_TILE:$t,REGTR_X3_SPLITS:$s"; done; done
timeout 1400 python tools/x3_bench.py --arms "default=REGTR_X3_DMA:1" $arms > $out/x3_sweep.md 2>&1
cat $out/x3_sweep.md
