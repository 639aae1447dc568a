#!/bin/bash
# default bench line (20 steps, side measurements on) with the pyramid stream at high (-1, shipped) and normal (0) priority, alternating on one box
out=gpurun_out/${1:-r05_pr}; mkdir -p $out
for i in 1 2; do for prio in -1 0; do
REGTR_DEV=1 REGTR_SIDE_PRIO=$prio python bench.py --no-cpu-baseline > $out/b.json 2>/dev/null
python -c "
import json
for l in open('$out/b.json').read().strip().splitlines():
    if l.startswith('{'):
        d=json.loads(l); print('prio$prio', round(d['ms_per_step'],3), round(d['value'],1), 'gather', round(d['roofline']['frac'],3))" >> $out/ab.txt
done; done
cat $out/ab.txt
