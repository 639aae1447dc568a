#!/bin/bash
out=gpurun_out/r03_l; mkdir -p $out; export TMPDIR=/tmp
REGTR_VARIANT=prof timeout 300 python tools/x3_prof_run.py > $out/x3d_prof.txt 2>&1; sort -u $out/x3d_prof.txt | awk 'NR%2==1' | head -60
