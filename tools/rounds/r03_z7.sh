#!/bin/bash
out=gpurun_out/r03_z7; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python tools/x3_bench.py --arms "base=REGTR_X3_IL:1" "after=REGTR_VARIANT:after" > $out/x3_after.md 2>&1
cat $out/x3_after.md
