"""Resnet-block tail at level-0 size: regtr_block_tail vs unary2 GEMM + shortcut GEMM + instnorm_apply.
    python tools/tail_bench.py [--clouds 128] [--rows 18900]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regtr_amd import ops  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clouds', type=int, default=128)
    ap.add_argument('--rows', type=int, default=18900)
    ap.add_argument('--reps', type=int, default=10)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(0)
    lens = [args.rows + (i * 37) % 1500 for i in range(args.clouds)]
    M, K1, K2, N = sum(lens), 32, 64, 128
    seg = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    x1 = torch.randn(M, K1, device=dev, generator=g) * 2 + 0.5
    f = torch.randn(M, K2, device=dev, generator=g)
    sw1 = ops.SplitWeight(torch.randn(N, K1, device=dev, generator=g) / 6, 'nk')
    sw2 = ops.SplitWeight(torch.randn(N, K2, device=dev, generator=g) / 8, 'nk')
    st = ops.instnorm_stats(x1, seg, max(lens))

    def separate():
        u, u_st = ops.gemm(x1, sw1, a_stats=st, a_seg_off=seg, want_stats=(seg, max(lens)))
        sc, sc_st = ops.gemm(f, sw2, want_stats=(seg, max(lens)))
        return ops.instnorm_apply(u, seg, max(lens), u_st, residual=sc, res_stats=sc_st, lrelu=True, out=u)

    def fused():
        return ops.block_tail(x1, st, f, sw1, sw2, seg, max(lens))

    err = (separate() - fused()).abs().max().item()
    t_s, t_f = timed(separate, args.reps), timed(fused, args.reps)
    print(f'M={M} clouds={args.clouds}: separate {t_s:.0f} us | block_tail {t_f:.0f} us | max |diff| {err:.2e}')


if __name__ == '__main__':
    main()
