#!/bin/bash
out=gpurun_out/r03_z9; mkdir -p $out; export TMPDIR=/tmp
for shape in "37888 256 3840" "141056 128 1920" "540672 64 960"; do
tag=$(echo $shape | tr ' ' '_')
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/${tag}_p$i -o p -- python tools/x3_one.py $shape 4 > $out/${tag}_p$i.log 2>&1 || tail -2 $out/${tag}_p$i.log
done
done
python - <<'PY'
import csv, glob, collections, os
out='gpurun_out/r03_z9'
res={}
for d in sorted(glob.glob(out+'/*_p*')):
    if not os.path.isdir(d): continue
    tag=os.path.basename(d).rsplit('_p',1)[0]
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_gemm_x3' in r['Kernel_Name']:
                t=res.setdefault(tag,{}).setdefault(r['Counter_Name'],[0.0,0]); t[0]+=float(r['Counter_Value']); t[1]+=1
with open(out+'/pmc_x3d_f16.md','w') as o:
    names=sorted({n for t in res.values() for n in t})
    o.write('| counter (mean per launch) | '+' | '.join(res)+' |\n|---|'+'---|'*len(res)+'\n')
    for n in names: o.write(f'| {n} | '+' | '.join(f"{res[t][n][0]/max(res[t][n][1],1):.4g}" if n in res[t] else '' for t in res)+' |\n')
    o.write('\n')
    for t,c in res.items():
        g=lambda n: c[n][0]/max(c[n][1],1)
        try:
            o.write(f'{t}: MFMA-busy {g("SQ_VALU_MFMA_BUSY_CYCLES")/(g("GRBM_GUI_ACTIVE")/8*1024)*100:.1f} % of SIMD cycles; waves: active {g("SQ_ACTIVE_INST_ANY")/g("SQ_WAVE_CYCLES")*100:.0f} %, issue-stall {g("SQ_WAIT_INST_ANY")/g("SQ_WAVE_CYCLES")*100:.0f} %, parked {g("SQ_WAIT_ANY")/g("SQ_WAVE_CYCLES")*100:.0f} %; mean resident waves per SIMD {g("SQ_WAVE_CYCLES")*4/(g("GRBM_GUI_ACTIVE")/8*1024):.2f}; VALU per MFMA {g("SQ_INSTS_VALU")/g("SQ_INSTS_MFMA"):.1f}\n')
        except Exception as e: o.write(f'{t}: {e}\n')
print(open(out+'/pmc_x3d_f16.md').read())
PY
find $out -name "*.csv" -size +2M -delete
