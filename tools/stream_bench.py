"""Micro-benchmark of the short-K encoder GEMMs at level-0 / level-1 size: strip kernel vs tiled kernel.
    python tools/stream_bench.py [--rows 2415616] [--reps 5]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regtr_amd import context, ops  # noqa: E402


def timed(fn, reps):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=2415616)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--r6', action='store_true', help='the wide-output shapes of a 192-pair forward; the tiled kernel in the f16 pair format, as the forward runs it')
    args = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    shapes = ((args.rows, 64, 128, False), (args.rows, 32, 128, True), (args.rows, 64, 32, False), (args.rows, 128, 32, False),
              (args.rows // 4, 64, 256, True), (args.rows // 4, 128, 64, False), (args.rows // 4, 128, 256, False))
    if args.r6:      # the round-6 shapes of a 192-pair forward: level-1 shortcut 128 -> 256, level-2 / level-3 unary2 128 -> 512, level-1 unary2 64 -> 256
        shapes = ((1883814, 128, 256, False), (545177, 128, 512, False), (151059, 128, 512, False), (1883814, 64, 256, True), (1883814, 128, 64, False))
    for M, K, N, fold in shapes:
        if not ops._lib.lib().regtr_gemm_stream_supported(M, N, K):
            print(f'K={K} N={N}: not served'); continue
        lens = np.full(384 if args.r6 else 128, M // (384 if args.r6 else 128), np.int32); lens[-1] += M - lens.sum()
        seg = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
        a = torch.randn(M, K, generator=g).cuda()
        sw = ops.SplitWeight((torch.randn(N, K, generator=g) / K ** 0.5).cuda(), 'nk')
        res = torch.randn(M, N, generator=g).cuda()
        a_st = ops.instnorm_stats(a, seg, int(lens.max())) if fold else None
        gb = (M * K * 4 + M * N * 4) / 1e9
        t_s = timed(lambda: ops.gemm_stream(a, sw, seg, a_stats=a_st, want_stats=True), args.reps)
        ops.use_stream_gemm = False
        with context.forward(a.device, f16_pair=bool(args.r6)):
            t_t = timed(lambda: ops.gemm(a, sw if N % 64 == 0 else sw.kn, a_stats=a_st, a_seg_off=seg if fold else None,
                                         want_stats=(seg, int(lens.max()))), args.reps)
        ops.use_stream_gemm = True
        print(f'K={K} N={N} M={M} fold={int(fold)}: strip {t_s:.0f} us ({gb / t_s * 1e3:.2f} TB/s) | tiled {t_t:.0f} us ({gb / t_t * 1e3:.2f} TB/s)')


if __name__ == "__main__":
    main()
