#!/bin/bash
# tests + default bench + ModelNet (bf16 / fp32) bench + kernel stats + MFMA-utilisation PMC pass on the ModelNet workload
out=gpurun_out/r02_d; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s > $out/pytest.log 2>&1; echo "pytest exit $?" >> $out/pytest.log
grep -E "mha precision|gemm_x3 planes|compute_dtype|passed|failed|Error|error|assert" $out/pytest.log | tail -40
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_3dmatch.json 2> $out/bench_3dmatch.err; python - <<PY
import json; d=json.loads(open('$out/bench_3dmatch.json').read().strip().splitlines()[-1]); print('3dmatch', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_secondary']['achieved'], d['roofline_secondary']['detail']['attention_s_per_step'])
PY
for dt in bf16 fp32 bf16x2; do
timeout 600 python bench.py --config modelnet --dtype $dt --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_modelnet_$dt.json 2> $out/bench_modelnet_$dt.err; python - <<PY
import json; d=json.loads(open('$out/bench_modelnet_$dt.json').read().strip().splitlines()[-1]); print('modelnet $dt', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['detail']['attention_s_per_step'], d.get('reduced_precision_error'))
PY
done
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o trace -- python bench.py --config modelnet --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $out/prof.log 2>&1
db=$(find $out/prof -name "*.db" | head -1); python tools/rocpd_stats.py $db > $out/kernel_stats_modelnet_bf16.md 2>&1; rm -rf $out/prof; head -14 $out/kernel_stats_modelnet_bf16.md
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_1 -o p -- python bench.py --config modelnet --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $out/pmc_1.log 2>&1 || tail -3 $out/pmc_1.log
python tools/pmc_summary.py --mfma $out > $out/pmc_mfma_modelnet_bf16.md 2>&1; head -12 $out/pmc_mfma_modelnet_bf16.md
find $out -name "*.csv" -size +4M -delete
