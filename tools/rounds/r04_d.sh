#!/bin/bash
# round 4, call d: harness (pre-forked LoaderPool, mmap slabs, batch parts), f16 gather v2: correctness, micro-benchmark, phase clocks
tag=${1:-r04_d}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_harness.py tests/test_gpu_ops.py -m gpu -q -rs -s -k "end_to_end or gather_f16 or cli" > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; grep -E "pairs/s|passed|failed|exit|rel|Error|error" $out/pytest.log | tail -12
timeout 300 python tools/gather_bench.py --pairs 64 --pre > $out/gather_f32.txt 2>&1; tail -8 $out/gather_f32.txt
timeout 300 python tools/gather_bench.py --pairs 64 --f16 > $out/gather_f16.txt 2>&1; tail -12 $out/gather_f16.txt
REGTR_VARIANT=gfprof timeout 300 python tools/gather_bench.py --pairs 64 --f16 --reps 1 > $out/gather_f16_prof.txt 2>&1; grep "gather_f16 Cin" $out/gather_f16_prof.txt | sort | uniq -c | sort -rn | head -12
E2E="python test.py --benchmark 3DLoMatch --config regtr_amd/conf/3dmatch.yaml --logdir /tmp/e2e_logs --synthetic 1781 --overlap lomatch --materialize /tmp/e2e_data --distinct 128"
timeout 900 $E2E --num_workers 4 > $out/e2e_pth_cold.log 2>&1; grep -E "End to end|loader:" $out/e2e_pth_cold.log | tail -2
for w in 4 8; do timeout 600 $E2E --num_workers $w > $out/e2e_pth_w$w.log 2>&1; grep -E "End to end|loader:" $out/e2e_pth_w$w.log | tail -2; done
timeout 600 $E2E --num_workers 4 --cache_dir /tmp/e2e_cache > $out/e2e_npy_build.log 2>&1; grep -E "End to end" $out/e2e_npy_build.log | tail -1
for w in 2 4 6; do timeout 600 $E2E --num_workers $w --cache_dir /tmp/e2e_cache > $out/e2e_npy_w$w.log 2>&1; grep -E "End to end|loader:" $out/e2e_npy_w$w.log | tail -2; done
