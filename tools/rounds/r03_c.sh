#!/bin/bash
# ablation of the row-strip split GEMM: where does the time go? (development variants, see X3R_ABL in csrc/gemm_x3.hip)
out=gpurun_out/r03_c; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python tools/x3_bench.py --arms "strip=REGTR_X3_STRIP:1" "strip_p2=X3_PLANES:2" "strip_p1=X3_PLANES:1" "tiled_p1=X3_PLANES:1,REGTR_X3_STRIP:0" \
  "noAload=REGTR_VARIANT:abl1" "nosplit=REGTR_VARIANT:abl2" "nobar=REGTR_VARIANT:abl4" "noBread=REGTR_VARIANT:abl8" "mfma_only=REGTR_VARIANT:abl15" > $out/x3_ablation.md 2>&1
cat $out/x3_ablation.md
