#!/bin/bash
# One gpurun call: GPU parity tests, bench line, kernel trace, HBM-traffic PMC passes. Outputs under gpurun_out/$1
tag=${1:-run}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
tail -3 $out/pytest.log
timeout 900 python bench.py $BENCH_ARGS > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats.md 2>&1; head -24 $out/kernel_stats.md
rm -f $db
# HBM traffic of the dominant kernel: separate counter passes, kernel-trace only
i=0
for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pmc_$i -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $out/pmc_$i.log 2>&1 || tail -3 $out/pmc_$i.log
done
python tools/pmc_summary.py --traffic $out k_kpconv_gather > $out/pmc_traffic.json 2>&1; tail -12 $out/pmc_traffic.json
python tools/pmc_summary.py --bench-traffic $out $tag > $out/pmc_traffic_bench.json 2>&1
python tools/pmc_summary.py --mfma $out > $out/pmc_mfma.md 2>&1; cat $out/pmc_mfma.md
find $out -name "*.csv" -size +4M -delete
