"""GPU: forwards on two host threads / two HIP streams against the same forwards one after the other.
    python tools/concurrency_probe.py [pairs per forward=96] [steps=12] [replicas=2]
Two MODEL REPLICAS (same weights), one per thread, each with its own batch of `pairs` pairs: a forward has two host waits (the pyramid's level sizes,
the status word), so one thread cannot keep the chip busy across forwards; a second thread's forward fills the 5-6 ms head of the first (cell grid
+ level-0 conv table, during which nothing else of that forward can run) and its host gaps.
Round 6: 2 x 96 pairs against one 192-pair forward: +1.9 % (below the bar: halving the batch costs ~6 % by itself).
This form answers the follow-up: 2 x 192 pairs concurrently against 2 x 192 one after the other."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from regtr_amd import RegTR  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 96
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
R = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device('cuda:0')
cfg, model, prs, batch = bench.build_workload('3dmatch', R * pairs, 20000, False, 0, dev, 'fp32')
halves = [{k: v[i * pairs:(i + 1) * pairs] for k, v in batch.items()} for i in range(R)]
models = [model]
for _ in range(R - 1):
    twin = RegTR(cfg).to(dev).eval()
    twin.load_state_dict(model.state_dict())
    models.append(twin)


def timed(fn):
    fn(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def run_seq(n):
    for _ in range(n):
        for h in halves:
            models[0](dict(h))


outs = [None] * R


def run_threads(n):
    def worker(i):
        with torch.cuda.stream(streams[i]), torch.no_grad():
            for _ in range(n):
                outs[i] = models[i](dict(halves[i]))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(R)]
    for t in th: t.start()
    for t in th: t.join()


with torch.no_grad():
    ref = [models[0](dict(h))['pose'].clone() for h in halves]
    t_seq = timed(run_seq)
    print(f'{R} forwards of {pairs} pairs, one after the other, one thread:            {t_seq * 1e3:7.2f} ms per {R * pairs} pairs = {R * pairs / t_seq:7.1f} pairs/s')
    for prio in (None,):
        streams = [torch.cuda.Stream(dev) for i in range(R)]
        t_thr = timed(run_threads)
        same = all(torch.equal(outs[i]['pose'], ref[i]) for i in range(R))
        print(f'{R} threads x {pairs} pairs, {R} model replicas, {R} streams: {t_thr * 1e3:7.2f} ms per {R * pairs} pairs = {R * pairs / t_thr:7.1f} pairs/s'
              f'  ({(t_seq / t_thr - 1) * 100:+.1f} %; poses bit-identical to the sequential run: {same})')
    print('peak GiB', round(torch.cuda.max_memory_allocated() / 2**30, 2))
