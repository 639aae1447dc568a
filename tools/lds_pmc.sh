#!/bin/bash
# LDS counters of every kernel of two forwards: bank-conflict cycles against all LDS-array cycles (is a swizzle wrong for gfx950's ds_read_b128 lane groups?)
out=gpurun_out/${1:-r05_lds}; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/$out/pmc -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --parity-pairs 0 --no-strict-f32 > $R/$out/pmc.log 2>&1 || tail -5 $R/$out/pmc.log
cd $R; python - <<'PY' $out
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/pmc/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in f:
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_BUSY_CYCLES': n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:28]
with open(out + '/lds_pmc.md', 'w') as o:
    o.write('| kernel | launches | LDS array cycles (IDX_ACTIVE) | bank-conflict cycles | conflict share | addr-conflict | LDS instr | LDS-issue stall / wave cycles |\n|---|---|---|---|---|---|---|---|\n')
    for k, c in rows:
        ia = c.get('SQ_LDS_IDX_ACTIVE', 0)
        o.write(f"| {k} | {n[k]} | {ia:.3g} | {c.get('SQ_LDS_BANK_CONFLICT', 0):.3g} | {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(ia, 1):.3f} | {c.get('SQ_LDS_ADDR_CONFLICT', 0):.3g} | {c.get('SQ_INSTS_LDS', 0):.3g} | {c.get('SQ_WAIT_INST_LDS', 0) / max(c.get('SQ_WAVE_CYCLES', 0), 1):.3f} |\n")
print(open(out + '/lds_pmc.md').read())
PY
find $out -name "*.csv" -size +4M -delete
