#!/bin/bash
tag=${1:-r04_i}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
E2E="python test.py --benchmark 3DLoMatch --config regtr_amd/conf/3dmatch.yaml --logdir /tmp/e2e_logs --synthetic 1781 --overlap lomatch --materialize /tmp/e2e_data --distinct 128"
timeout 900 $E2E --num_workers 4 --cache_dir /tmp/e2e_cache > $out/e2e_build.log 2>&1; grep -E "End to end" $out/e2e_build.log | tail -1
for w in 2 4; do timeout 600 $E2E --num_workers $w --cache_dir /tmp/e2e_cache > $out/e2e_npy_w$w.log 2>&1; grep -E "End to end|loader:|pairs on" $out/e2e_npy_w$w.log | tail -3 | cut -c1-700; done
timeout 600 $E2E --num_workers 2 --cache_dir /tmp/e2e_cache --reserve_gb 0 > $out/e2e_npy_w2_nores.log 2>&1; grep -E "End to end" $out/e2e_npy_w2_nores.log | tail -1
timeout 600 $E2E --num_workers 4 > $out/e2e_pth_w4.log 2>&1; grep -E "End to end|loader:" $out/e2e_pth_w4.log | tail -2 | cut -c1-400
timeout 600 $E2E --num_workers 0 > $out/e2e_thread.log 2>&1; grep -E "End to end" $out/e2e_thread.log | tail -1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-strict-f32 > $out/bench.json 2> $out/bench.err; python -c "import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); p=d['parity']; print('bench', round(d['value'],1), round(d['ms_per_step'],3), p['ok'], p['pose_max_abs'], p['corr_max_abs'])"
