#!/bin/bash
out=gpurun_out/r03_u; mkdir -p $out; export TMPDIR=/tmp
REGTR_VARIANT=mhaprof timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --parity-pairs 0 > $out/mha_prof.txt 2>&1
grep "^mha" $out/mha_prof.txt | sort -u | head -40
