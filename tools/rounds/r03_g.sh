#!/bin/bash
out=gpurun_out/r03_g; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python tools/x3_bench.py --arms "dma=REGTR_X3_DMA:1" "noA=REGTR_VARIANT:dabl1" "Bplane0=REGTR_VARIANT:dabl2" "noB=REGTR_VARIANT:dabl4" "noAB=REGTR_VARIANT:dabl5" > $out/x3_dma_abl.md 2>&1
cat $out/x3_dma_abl.md
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCP|TCC|TA|TD|GRBM|LDS)_[A-Z0-9_]+" | sort -u | tr '\n' ' ' > $out/counters.txt; wc -c $out/counters.txt
