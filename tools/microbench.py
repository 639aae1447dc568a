"""GPU micro-benchmarks of individual C-ABI kernels (run on the MI355X box): python tools/microbench.py"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regtr_amd import ops  # noqa: E402


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    dev = 'cuda'
    print(subprocess.run('rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -5', shell=True,
                         capture_output=True, text=True).stdout)
    print('--- gemm (M, N, K): us, TFLOP/s')
    for M, N, K in [(2432, 256, 256), (2432, 768, 256), (2432, 1024, 256), (2432, 256, 1024), (2432, 256, 3840),
                    (9728, 256, 256), (38912, 256, 256), (152000, 32, 480), (152000, 128, 32), (152000, 128, 64),
                    (34000, 64, 960), (34000, 256, 64), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev)
        us = timeit(lambda: ops.gemm(a, b), reps=20 if M * N * K > 1e11 else 50)
        print(f'  {M:7d} {N:5d} {K:5d}: {us:9.1f} us  {2 * M * N * K / us / 1e6:7.1f} TF')
    print('--- mha (tokens per cloud, clouds): us')
    for n, c in [(400, 2), (400, 8), (400, 32), (2000, 2)]:
        N = n * c
        qkv = torch.randn(N, 768, device=dev)
        seg = torch.arange(c + 1, dtype=torch.int32, device=dev) * n
        kv = torch.arange(c, dtype=torch.int32, device=dev)
        us = timeit(lambda: ops.mha(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], seg, kv, n, 8))
        fl = 4 * 32 * n * n * 8 * c
        print(f'  {n:5d} x {c:3d}: {us:8.1f} us  {fl / us / 1e6:6.1f} TF')
    print('--- instnorm stats/apply (N, C): us, GB/s')
    for n, C in [(152000, 128), (152000, 32), (34000, 256), (2432, 1024)]:
        x = torch.randn(n, C, device=dev); r = torch.randn(n, C, device=dev)
        seg = torch.tensor([0, n // 2, n], dtype=torch.int32, device=dev)
        us = timeit(lambda: ops.instnorm_stats(x, seg, n - n // 2))
        st = ops.instnorm_stats(x, seg, n - n // 2)
        us2 = timeit(lambda: ops.instnorm_apply(x, seg, n - n // 2, st, residual=r, res_stats=st, lrelu=True))
        print(f'  {n:7d} {C:5d}: stats {us:7.1f} us {n * C * 4 / us / 1e3:7.0f} GB/s | apply {us2:7.1f} us {3 * n * C * 4 / us2 / 1e3:7.0f} GB/s')
    print(subprocess.run('rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -5', shell=True,
                         capture_output=True, text=True).stdout)


if __name__ == '__main__':
    main()
