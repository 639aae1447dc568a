#!/bin/bash
out=gpurun_out/r03_k; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python tools/x3_bench.py --arms "tiled=REGTR_X3_STRIP:0" "dma_c=REGTR_VARIANT:nohand" "hand=REGTR_X3_STRIP:1" "hand_t0=REGTR_X3_TILE:0" "hand_t1=REGTR_X3_TILE:1" > $out/x3_hand.md 2>&1
cat $out/x3_hand.md
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or split or x3 or unary or kpconv" > $out/pytest_gemm.log 2>&1; tail -3 $out/pytest_gemm.log
