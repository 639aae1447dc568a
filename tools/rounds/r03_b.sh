#!/bin/bash
# round 3: row-strip split GEMM vs the tiled kernel on RegTR's shapes (A/B, separate processes), then the dense-op parity tests
out=gpurun_out/r03_b; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python tools/x3_bench.py > $out/x3_bench.md 2>&1; cat $out/x3_bench.md
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or split or x3 or unary or kpconv" > $out/pytest_gemm.log 2>&1; tail -3 $out/pytest_gemm.log
