"""Allocator behaviour of the two-stream forward over many steps (record_stream defers block reuse): reserved memory must plateau.
    python tools/mem_check.py [--steps 60]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from regtr_amd import RegTR, load_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--pairs', type=int, default=64)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = load_config(os.path.join(bench.ROOT, 'regtr_amd', 'conf', '3dmatch.yaml'))
    torch.manual_seed(0)
    model = RegTR(cfg).to(dev).eval()
    pairs = [bench.synth_pair(i, 20000) for i in range(args.pairs)]
    batch = {'src_xyz': [torch.from_numpy(s).to(dev) for s, _ in pairs], 'tgt_xyz': [torch.from_numpy(t).to(dev) for _, t in pairs]}
    for i in range(args.steps):
        out = model({'src_xyz': list(batch['src_xyz']), 'tgt_xyz': list(batch['tgt_xyz'])})
        del out
        if i % 10 == 9 or i < 3:
            torch.cuda.synchronize()
            print(f'step {i + 1:3d}: allocated {torch.cuda.memory_allocated() / 2**30:6.2f} GiB  reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB '
                  f'peak {torch.cuda.max_memory_allocated() / 2**30:6.2f} GiB')


if __name__ == '__main__':
    main()
