#!/bin/bash
out=gpurun_out/${1:-r05_diag}; mkdir -p $out
{
python tools/real_diag.py
python tools/real_diag.py --dtype fp32x3
REGTR_DEV=1 REGTR_BLOCK_TAIL=0 python tools/real_diag.py
REGTR_DEV=1 REGTR_STREAM_GEMM=0 python tools/real_diag.py
REGTR_DEV=1 REGTR_PRENORM=0 python tools/real_diag.py
REGTR_DEV=1 REGTR_F16_PAIR=0 python tools/real_diag.py
python tools/real_diag.py --synthetic --check 0 8
} 2>&1 | grep -v Warning | tee $out/diag.txt
