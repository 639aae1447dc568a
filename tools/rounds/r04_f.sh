#!/bin/bash
tag=${1:-r04_f}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
E2E="python test.py --benchmark 3DLoMatch --config regtr_amd/conf/3dmatch.yaml --logdir /tmp/e2e_logs --synthetic 1781 --overlap lomatch --materialize /tmp/e2e_data --distinct 128"
timeout 900 $E2E --num_workers 4 --cache_dir /tmp/e2e_cache > $out/e2e_build.log 2>&1; grep -E "End to end" $out/e2e_build.log | tail -1
timeout 600 $E2E --num_workers 2 --cache_dir /tmp/e2e_cache > $out/e2e_npy_w2.log 2>&1; grep -E "End to end|loader:|pairs on" $out/e2e_npy_w2.log | tail -3
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout 600 $E2E --num_workers 2 --cache_dir /tmp/e2e_cache > $out/e2e_npy_w2_exp.log 2>&1; grep -E "End to end|loader:|pairs on" $out/e2e_npy_w2_exp.log | tail -3
timeout 600 $E2E --num_workers 2 --cache_dir /tmp/e2e_cache --warmup_points 20000 > $out/e2e_npy_w2_wp20.log 2>&1; grep -E "End to end|loader:|pairs on" $out/e2e_npy_w2_wp20.log | tail -3
