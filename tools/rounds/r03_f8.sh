#!/bin/bash
out=gpurun_out/r03_f8; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python tools/x3_bench.py --arms "f16=REGTR_F16_PAIR:1" "t0s2=REGTR_X3_TILE:0,REGTR_X3_SPLITS:2" "t1s1=REGTR_X3_TILE:1" "t1s2=REGTR_X3_TILE:1,REGTR_X3_SPLITS:2" "t1s3=REGTR_X3_TILE:1,REGTR_X3_SPLITS:3" > $out/x3_f16_split.md 2>&1
cat $out/x3_f16_split.md
