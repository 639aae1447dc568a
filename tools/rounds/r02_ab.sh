#!/bin/bash
# A/B of an env-switchable feature on the same box: bash tools/rounds/r02_ab.sh VAR
var=$1; out=gpurun_out/r02_ab_$var; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "gemm or forward_vs or kitchen or batched" > $out/pytest.log 2>&1; tail -1 $out/pytest.log
for i in 1 2; do for v in 1 0; do env $var=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['value'],1), round(d['ms_per_step'],3))"; done; done
