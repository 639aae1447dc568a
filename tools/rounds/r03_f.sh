#!/bin/bash
out=gpurun_out/r03_f; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python tools/x3_bench.py --arms "tiled=REGTR_X3_STRIP:0" "pipe=REGTR_X3_DMA:0" "dma=REGTR_X3_DMA:1" "dma_t1=REGTR_X3_TILE:1" "dma_p1=X3_PLANES:1" > $out/x3_dma.md 2>&1
cat $out/x3_dma.md
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or split or x3 or unary or kpconv" > $out/pytest_gemm.log 2>&1; tail -3 $out/pytest_gemm.log
