#!/bin/bash
tag=${1:-r04_g}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
for arm in "dev 0" "dev 1" "devpw 1"; do set -- $arm; REGTR_VARIANT=$1 REGTR_X3_ARES=$2 timeout 300 python tools/ares_bench.py > $out/ares_$1_$2.txt 2>&1; cat $out/ares_$1_$2.txt | tail -10; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-strict-f32 > $out/bench.json 2> $out/bench.err; python -c "import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); p=d['parity']; print('bench', round(d['value'],1), round(d['ms_per_step'],3), p['ok'], p['pose_max_abs'], p['corr_max_abs'])"
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_range.py -m gpu -q -k "gemm or range or cross or mha or f16" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
E2E="python test.py --benchmark 3DLoMatch --config regtr_amd/conf/3dmatch.yaml --logdir /tmp/e2e_logs --synthetic 1781 --overlap lomatch --materialize /tmp/e2e_data --distinct 128"
timeout 900 $E2E --num_workers 4 --cache_dir /tmp/e2e_cache > $out/e2e_build.log 2>&1; grep -E "End to end" $out/e2e_build.log | tail -1
timeout 600 $E2E --num_workers 2 --cache_dir /tmp/e2e_cache > $out/e2e_npy_w2.log 2>&1; grep -E "End to end|loader:|pairs on" $out/e2e_npy_w2.log | tail -3 | cut -c1-900
timeout 600 $E2E --num_workers 4 > $out/e2e_pth_w4.log 2>&1; grep -E "End to end" $out/e2e_pth_w4.log | tail -1
