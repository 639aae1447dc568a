"""GPU: two half-size forwards on two host threads / two HIP streams against one full-size forward.
    python tools/concurrency_probe.py [pairs=192] [steps=12]
Round 2's micro-batch loss was measured at 16-32 pairs per forward, where launches shrink below the chip; at 96 pairs the per-launch rates are on the
plateau (profiles/r05_z_batch_sweep.txt), so the question is only whether issue-bound gathers co-issue with waitcnt-bound GEMMs of the OTHER forward."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device('cuda:0')
cfg, model, prs, batch = bench.build_workload('3dmatch', pairs, 20000, False, 0, dev, 'fp32')
half = pairs // 2
halves = [{k: v[:half] for k, v in batch.items()}, {k: v[half:] for k, v in batch.items()}]


def run_full(n):
    for _ in range(n):
        model(dict(batch))


def timed(fn):
    fn(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def run_seq_halves(n):
    for _ in range(n):
        model(dict(halves[0])); model(dict(halves[1]))


def run_threads(n):
    def worker(i):
        st = streams[i]
        with torch.cuda.stream(st), torch.no_grad():
            for _ in range(n):
                model(dict(halves[i]))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()


with torch.no_grad():
    t_full = timed(run_full)
    t_seq = timed(run_seq_halves)
    for prio in (None, (0, 0), (-1, 0)):
        streams = [torch.cuda.Stream(dev) if prio is None else torch.cuda.Stream(dev, priority=prio[i]) for i in range(2)]
        t_thr = timed(run_threads)
        print(f'two threads x {half} pairs on two streams (priorities {prio}): {t_thr * 1e3:7.2f} ms per {pairs} pairs = {pairs / t_thr:7.1f} pairs/s')
    print(f'one forward of {pairs} pairs:                                   {t_full * 1e3:7.2f} ms = {pairs / t_full:7.1f} pairs/s')
    print(f'two forwards of {half} pairs, one after the other, one thread:    {t_seq * 1e3:7.2f} ms = {pairs / t_seq:7.1f} pairs/s')
