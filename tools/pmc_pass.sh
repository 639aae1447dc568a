#!/bin/bash
# PMC counter passes (one rocprofv3 run per counter group, kernel-trace only) over a command.
# usage: tools/pmc_pass.sh <outdir> <cmd...>      -> <outdir>/pmc_<group>/... csv, then tools/pmc_summary.py <outdir>
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pmc_$i -o p -- "$@" > $out/pmc_$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -3 $out/pmc_$i.log)"
done
python tools/pmc_summary.py $out > $out/pmc_summary.md 2>&1
cat $out/pmc_summary.md | head -60
# keep only the summaries + small csvs
find $out -name "*.csv" -size +8M -delete
