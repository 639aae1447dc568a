"""Micro-benchmark of the attention core on the cross-encoder's shape of a 64-pair forward (128 clouds of ~295 tokens, 8 heads of 32;
self and cross attention alternate as in a layer).  A/B kernel variants: REGTR_DEV=1 REGTR_VARIANT=name python tools/mha_bench.py
    [pairs=64] [lo=230] [hi=360] [precision=0]   (round 6: 192 330 460 3 = the 3DMatch default line's tokens in the f16 pair form; 256 560 640 1 = ModelNet bf16)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regtr_amd import ops  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(0)
    a = [int(x) for x in sys.argv[1:5]] + [64, 230, 360, 0][len(sys.argv[1:5]):]
    B, lo, hi, prec = a
    lens = rng.integers(lo, hi, size=2 * B)
    seg = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32, device=dev)
    N = int(lens.sum())
    kv_self = torch.arange(2 * B, dtype=torch.int32, device=dev)
    kv_cross = torch.cat((torch.arange(B, 2 * B), torch.arange(0, B))).to(torch.int32).to(dev)
    torch.manual_seed(0)
    qkv = torch.randn(N, 768, device=dev)
    mx = int(lens.max())
    outs = []
    for kv in (kv_self, kv_cross):
        run = lambda: ops.mha(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], seg, kv, mx, 8, precision=prec)
        for _ in range(5):
            o = run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            o = run()
        e1.record(); torch.cuda.synchronize()
        outs.append((e0.elapsed_time(e1) / 50 * 1e3, float(o.double().sum()), float(o.abs().max())))
    print(f'variant={os.environ.get("REGTR_VARIANT", ""):10s} tokens {N} ({2 * B} clouds of {lo}-{hi}, precision {prec}): self {outs[0][0]:7.1f} us  cross {outs[1][0]:7.1f} us   chk {outs[0][1]:.9e} {outs[1][1]:.9e}')


if __name__ == '__main__':
    main()
