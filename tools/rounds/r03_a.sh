#!/bin/bash
# round 3, first GPU call: the new parity tests + whole GPU suite, default bench line (with the parity field), parity-mode line,
# lomatch (configs[3]) line, end-to-end harness over materialised .pth files
out=gpurun_out/r03_a; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log; grep -E "max abs diff|passed|failed|error" $out/pytest.log | tail -20
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?"; tail -c 1500 $out/bench.json
timeout 600 python bench.py --parity-mode --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $out/bench_paritymode.json 2> $out/bench_paritymode.err; echo "parity-mode exit $?"; tail -c 900 $out/bench_paritymode.json
timeout 900 python bench.py --config lomatch --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_lomatch.json 2> $out/bench_lomatch.err; echo "lomatch exit $?"; tail -c 1200 $out/bench_lomatch.json
# end-to-end harness: 1781 lomatch-like pairs as .pth files -> loader thread -> H2D -> forward -> gather -> est.log
timeout 900 python test.py --benchmark 3DLoMatch --config regtr_amd/conf/3dmatch.yaml --logdir /tmp/e2e_logs --synthetic 1781 --overlap lomatch --materialize /tmp/e2e_data --batch 64 > $out/e2e.log 2>&1; echo "e2e exit $?"; grep -E "End to end|pairs/s|materialised" $out/e2e.log | tail -5
timeout 600 python test.py --benchmark 3DLoMatch --config regtr_amd/conf/3dmatch.yaml --logdir /tmp/e2e_logs --synthetic 1781 --overlap lomatch --materialize /tmp/e2e_data --batch 64 > $out/e2e_warm.log 2>&1; echo "e2e warm exit $?"; grep -E "End to end|pairs/s" $out/e2e_warm.log | tail -3
