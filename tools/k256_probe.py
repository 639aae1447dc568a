"""GPU: what a short-K split GEMM launch spends outside its k loop.  The cross-encoder / head products (K = 256) sit at 0.21-0.31 of their
roofline (bench.py roofline_gemm); this times the same [M, N] launch at K = 128 ... 2048 (f16 pair, bias epilogue as the in-projection has
it) and fits t(K) = fixed + per_ktile * K / 32: `fixed` is prologue + epilogue + launch, `per_ktile` the marginal cost of 32 more k.
    python tools/k256_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regtr_amd import ops  # noqa: E402

dev = 'cuda'
torch.manual_seed(0)
print('| M | N | K = 128 | 256 | 512 | 1024 | 2048 us | fixed us | per k-tile us (all workgroups) | fixed share at K = 256 | HBM-roofline us at K = 256 | MFMA-roofline us (3 terms) |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|')
for M, N in ((37723, 768), (37723, 256), (37723, 1024), (226338, 256), (309760, 768)):
    ts = []
    Ks = (128, 256, 512, 1024, 2048)
    for K in Ks:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; bias = torch.randn(N, device=dev)
        sw = ops.SplitWeight(w, 'nk')

        def run():
            with ops.f16_pair(True):
                return ops.gemm(a, sw, bias=bias)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        del a, w, sw
    A = np.stack([np.ones(len(Ks)), np.array(Ks) / 32.0], 1)
    fixed, per = np.linalg.lstsq(A, np.array(ts), rcond=None)[0]
    hbm = (4.0 * M * 256 + 4.0 * M * N + 4.0 * 256 * N) / 8e12 * 1e6
    mfma = 2.0 * M * N * 256 * 3 / 2.5e15 * 1e6
    print(f'| {M} | {N} | ' + ' | '.join(f'{t:.1f}' for t in ts) + f' | {fixed:.1f} | {per:.2f} | {fixed / ts[1]:.2f} | {hbm:.1f} | {mfma:.1f} |')
