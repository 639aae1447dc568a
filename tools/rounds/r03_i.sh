#!/bin/bash
out=gpurun_out/r03_i; mkdir -p $out; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_WAVES" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "GRBM_GUI_ACTIVE FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o p -- python tools/x3_one.py 37888 256 3840 4 > $out/p$i.log 2>&1 || tail -2 $out/p$i.log
done
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/r03_i/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_gemm_x3' in r['Kernel_Name']:
            t = tot[r['Counter_Name']]; t[0] += float(r['Counter_Value']); t[1] += 1
names = sorted(tot)
with open('gpurun_out/r03_i/pmc_x3d_L3.md', 'w') as o:
    o.write('| counter | mean per launch |\n|---|---|\n')
    for n in names: o.write(f'| {n} | {tot[n][0] / max(tot[n][1], 1):.4g} |\n')
print(open('gpurun_out/r03_i/pmc_x3d_L3.md').read())
PY
find $out -name "*.csv" -size +2M -delete
