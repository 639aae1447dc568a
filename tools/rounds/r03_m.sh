#!/bin/bash
out=gpurun_out/r03_m; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python tools/x3_bench.py --arms "tiled=REGTR_X3_STRIP:0" "default=REGTR_X3_STRIP:1" "t1_ar2=REGTR_X3_TILE:1,REGTR_X3_ARING:2" "t1_ar3=REGTR_X3_TILE:1" "t0=REGTR_X3_TILE:0" "t3_256x128=REGTR_X3_TILE:3" > $out/x3_ring.md 2>&1
cat $out/x3_ring.md
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or split or x3 or unary or kpconv" > $out/pytest_gemm.log 2>&1; tail -3 $out/pytest_gemm.log
