#!/bin/bash
# the level-0 KPConv gather with the J dependent f32 MFMAs split into two accumulator chains (build: REGTR_VARIANT=chains2
# REGTR_VARIANT_FLAGS=-DRG_MG_CHAINS=2 python -m regtr_amd.build) against the product build: launch times + issue / MFMA counters.
tag=${1:-r05_chain}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
{
echo "== product build"; python tools/gather_bench.py --pre --pairs 64 --levels 0,1,2,4 --against chains2
echo "== chains2 build"; REGTR_DEV=1 REGTR_VARIANT=chains2 python tools/gather_bench.py --pre --pairs 64 --levels 0,1,2,4
echo "== product build again"; python tools/gather_bench.py --pre --pairs 64 --levels 0,1,2,4
} 2>&1 | grep -v Warning | tee $out/times.txt
for v in product chains2; do
  envs="X=1"; [ $v = chains2 ] && envs="REGTR_DEV=1 REGTR_VARIANT=chains2"
  env $envs timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/$v/pmc_1 -o p -- python tools/gather_bench.py --pre --pairs 64 --levels 0,1 --reps 3 > $out/pmc_$v.log 2>&1 || tail -3 $out/pmc_$v.log
  python tools/pmc_summary.py $out/$v k_kpconv_gather_mfma > $out/pmc_$v.md 2>&1; cat $out/pmc_$v.md
  find $out/$v -name "*.csv" -size +2M -delete
done
