"""Times the preprocessing pyramid (grid subsample + cell grids + radius queries) on the bench workload:
    [REGTR_VARIANT=name] python tools/preprocess_bench.py [--pairs 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from regtr_amd import load_config  # noqa: E402
from regtr_amd.kpconv import Preprocessor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=64)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--per-query', action='store_true', help='route the conv (self) tables through the per-query kernel too (A/B)')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    if args.per_query:
        from regtr_amd import ops
        ops.self_query_kernel = False
    cfg = load_config(os.path.join(bench.ROOT, 'regtr_amd', 'conf', '3dmatch.yaml'))
    pairs = [bench.synth_pair(i, 20000) for i in range(args.pairs)]
    pts = [torch.from_numpy(s).to(dev) for s, _ in pairs] + [torch.from_numpy(t).to(dev) for _, t in pairs]
    pre = Preprocessor(cfg)
    for _ in range(2):
        pre(pts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.reps):
        meta = pre(pts)
    e1.record(); torch.cuda.synchronize()
    chk = sum(int(t.sum()) for t in meta['_neighbors_i32'])
    print(f'preprocess {e0.elapsed_time(e1) / args.reps:.3f} ms per forward of {args.pairs} pairs  chk={chk}  variant={os.environ.get("REGTR_VARIANT", "")} per_query={args.per_query}')


if __name__ == '__main__':
    main()
