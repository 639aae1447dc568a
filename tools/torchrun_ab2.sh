#!/bin/bash
# plain launch vs the driver's torchrun launch (world 1), short and long timed regions, alternating on one warm box
out=gpurun_out/${1:-r05_tr3}; mkdir -p $out
line() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['ms_per_step'],2), round(d['value'],1))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
B="--warmup 3 --no-cpu-baseline --no-roofline --no-strict-f32 --parity-pairs 0"
plain() { python bench.py --steps $1 $B > $out/plain$1.json 2> $out/plain$1.err; line $out/plain$1.json plain_$1steps; }
tr() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus 1 --steps $1 $B > $out/tr$1.json 2> $out/tr$1.err; line $out/tr$1.json torchrun_$1steps; }
plain 8; plain 24; tr 24 29551; plain 24; tr 24 29552; tr 8 29553; plain 8
