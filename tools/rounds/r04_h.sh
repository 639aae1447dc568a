#!/bin/bash
tag=${1:-r04_h}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
REGTR_VARIANT=dev REGTR_X3_ARES=1 timeout 300 python tools/ares_bench.py > $out/ares_dev_1.txt 2>&1; tail -10 $out/ares_dev_1.txt
REGTR_VARIANT=devprof REGTR_X3_ARES=1 timeout 300 python tools/ares_bench.py > $out/ares_prof.txt 2>&1; grep "ares K" $out/ares_prof.txt | sort | uniq -c | sort -rn | head -16
