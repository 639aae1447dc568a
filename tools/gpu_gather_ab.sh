#!/bin/bash
# gather micro-benchmark for the default build + variants, GPU tests, optional PMC on the default build
out=gpurun_out/$1; mkdir -p $out; shift
timeout 600 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for v in "" "$@"; do REGTR_VARIANT=$v timeout 300 python tools/gather_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $out/gather_ab.log; done
if [ -n "$PMC" ]; then bash tools/pmc_pass.sh $out python tools/gather_bench.py --reps 2; fi
