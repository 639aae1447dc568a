#!/bin/bash
# phase probe of the short-K row-strip GEMM: the same launches with the epilogue's C stores switched off (tools/x3_phase_probe.patch applied to a
# -DREGTR_DEV_ENV=1 variant build: REGTR_X3_PROBE=1) -- full minus no-store = the store time a persistent kernel could hide at best
out=gpurun_out/${1:-r05_phase}; mkdir -p $out
for pr in 0 1 0 1; do echo "== REGTR_X3_PROBE=$pr"; REGTR_DEV=1 REGTR_VARIANT=devenv REGTR_X3_PROBE=$pr python tools/k256_probe.py 2>/dev/null | tail -5 | cut -d'|' -f2-10; done | tee $out/phase.txt
