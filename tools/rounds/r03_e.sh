#!/bin/bash
out=gpurun_out/r03_e; mkdir -p $out; export TMPDIR=/tmp
REGTR_VARIANT=prof timeout 300 python tools/x3_prof_run.py > $out/x3_prof.txt 2>&1; sort -u $out/x3_prof.txt | head -80
