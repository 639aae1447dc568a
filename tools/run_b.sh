out=gpurun_out/r06_b; mkdir -p $out; export TMPDIR=/tmp
python tools/probe/cmp_run.py > $out/cmp_probe.txt 2>&1; cat $out/cmp_probe.txt
timeout 300 python tools/step_times.py 192 30 real > $out/step_real_sync.txt 2>&1; head -6 $out/step_real_sync.txt
timeout 300 python tools/step_times.py 192 30 real nosync > $out/step_real_nosync.txt 2>&1; head -6 $out/step_real_nosync.txt
timeout 300 python tools/step_times.py 192 30 nosync > $out/step_synth_nosync.txt 2>&1; head -6 $out/step_synth_nosync.txt
timeout 900 python bench.py --collect-pmc --pmc-tag r06_b > $out/collect_pmc.log 2>&1; tail -2 $out/collect_pmc.log | cut -c1-600
cp profiles/pmc_traffic.json $out/; cp profiles/r06_b_pmc_traffic_kernels.md $out/
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 3000 $out/bench.json | cut -c1-3000; tail -3 $out/bench.err
timeout 900 python -m pytest tests/test_gpu_bench_batch.py -m gpu -q -x -k all_pairs -s 2>&1 | tail -8
