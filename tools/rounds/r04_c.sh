#!/bin/bash
# round 4, call c: unary2 pre-apply A/B, harness end to end (256-pair test, 1781-pair set cold / cached), collect-pmc
tag=${1:-r04_c}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
for m in 0 1 2; do
  REGTR_PREAPPLY_UNARY2=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --parity-pairs 2 --no-strict-f32 > $out/bench_pre$m.json 2> $out/bench_pre$m.err
  python -c "import json; d=json.loads(open('$out/bench_pre$m.json').read().strip().splitlines()[-1]); print('preapply $m', round(d['value'],1), round(d['ms_per_step'],3), d['parity']['ok'], d['parity']['pose_max_abs'])"
done
timeout 900 python -m pytest tests/test_harness.py tests/test_gpu_model.py -m gpu -q -rs -s --durations=5 > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; grep -E "pairs/s|passed|failed|exit|max abs diff" $out/pytest.log | tail -30
E2E="python test.py --benchmark 3DLoMatch --config regtr_amd/conf/3dmatch.yaml --logdir /tmp/e2e_logs --synthetic 1781 --overlap lomatch --materialize /tmp/e2e_data --distinct 128"
( time timeout 900 $E2E --num_workers 8 ) > $out/e2e_pth_cold.log 2>&1; grep -E "End to end|materialised|real" $out/e2e_pth_cold.log | tail -3
timeout 600 $E2E --num_workers 8 > $out/e2e_pth_warm.log 2>&1; grep -E "End to end" $out/e2e_pth_warm.log | tail -1
timeout 600 $E2E --num_workers 16 > $out/e2e_pth_w16.log 2>&1; grep -E "End to end" $out/e2e_pth_w16.log | tail -1
timeout 600 $E2E --num_workers 8 --cache_dir /tmp/e2e_cache > $out/e2e_npy_build.log 2>&1; grep -E "End to end" $out/e2e_npy_build.log | tail -1
timeout 600 $E2E --num_workers 8 --cache_dir /tmp/e2e_cache > $out/e2e_npy_warm.log 2>&1; grep -E "End to end" $out/e2e_npy_warm.log | tail -1
timeout 600 $E2E --num_workers 4 --cache_dir /tmp/e2e_cache > $out/e2e_npy_w4.log 2>&1; grep -E "End to end" $out/e2e_npy_w4.log | tail -1
timeout 600 $E2E --num_workers 0 > $out/e2e_thread.log 2>&1; grep -E "End to end" $out/e2e_thread.log | tail -1
nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout 600 python bench.py --collect-pmc --pmc-tag $tag > $out/collect_pmc.log 2>&1; tail -3 $out/collect_pmc.log; cp profiles/pmc_traffic.json $out/ 2>/dev/null; cp profiles/${tag}_pmc_traffic_kernels.md $out/ 2>/dev/null
