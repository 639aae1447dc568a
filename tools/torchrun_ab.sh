#!/bin/bash
# does the torchrun launch (the driver's N > 1 mode) run the same forward as the plain launch?  world 1, same box, alternating
out=gpurun_out/${1:-r05_tr}; mkdir -p $out
line() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['ms_per_step'],2), round(d['value'],1))" 2>/dev/null || { echo "$2 FAILED"; tail -3 ${1%.json}.err; }; }
A="--steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-strict-f32 --parity-pairs 0"
for i in 1 2; do
python bench.py $A > $out/plain.json 2> $out/plain.err; line $out/plain.json plain
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 1 $A > $out/tr.json 2> $out/tr.err; line $out/tr.json torchrun
OMP_NUM_THREADS=16 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$i bench.py --gpus 1 $A > $out/tr16.json 2> $out/tr16.err; line $out/tr16.json torchrun_omp16
OMP_NUM_THREADS=1 python bench.py $A > $out/plain1.json 2> $out/plain1.err; line $out/plain1.json plain_omp1
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 $A --pairs 64 > $out/tr64.json 2> $out/tr64.err; line $out/tr64.json torchrun_pairs64
