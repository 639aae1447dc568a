#!/bin/bash
# what does an initialised RCCL process group (world 1) cost a forward, and does a high-priority pyramid stream (its own hardware-queue pool) remove it?
out=gpurun_out/${1:-r05_do}; mkdir -p $out; export TMPDIR=/tmp
line() { python -c "
import json
for l in open('$1').read().strip().splitlines():
    if l.startswith('{'):
        d=json.loads(l); print('$2', round(d['ms_per_step'],3), round(d['value'],1))" 2>/dev/null; }
B="--steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-strict-f32 --parity-pairs 0"
D="env RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1"
python bench.py $B > /dev/null 2>&1     # warm the box
for i in 1 2 3; do
for prio in 0 -1; do
REGTR_DEV=1 REGTR_SIDE_PRIO=$prio python bench.py $B > $out/p.json 2>/dev/null; line $out/p.json plain_prio$prio
REGTR_DEV=1 REGTR_SIDE_PRIO=$prio $D MASTER_PORT=2959$i python bench.py --gpus 1 $B > $out/d.json 2>/dev/null; line $out/d.json dist_prio$prio
done; done
