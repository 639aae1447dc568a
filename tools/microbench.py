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
    print('--- gemm (M, N, K): exact-f32 MFMA kernel | bf16x3 split kernel  (us, f32-equivalent TFLOP/s)')
    for M, N, K in [(604382, 128, 32), (604382, 128, 64), (135017, 128, 32), (135017, 64, 128), (135017, 64, 960), (135017, 256, 64),
                    (135017, 256, 128), (135017, 64, 256), (35142, 64, 960), (35142, 256, 64), (35142, 128, 256), (35142, 128, 1920),
                    (35142, 512, 128), (35142, 512, 256), (35142, 128, 512), (9381, 128, 1920), (9381, 512, 128), (9381, 256, 512),
                    (9381, 256, 3840), (9381, 1024, 256), (9381, 1024, 512), (9381, 256, 1024), (9381, 768, 256), (9381, 256, 256),
                    (9381, 512, 256), (56286, 256, 256), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev)
        sw = ops.SplitWeight(b, 'kn')
        reps = 10 if M * N * K > 1e11 else 30
        us = timeit(lambda: ops.gemm(a, b), reps=reps)
        ops.force_x3_gemm = True
        us3 = timeit(lambda: ops.gemm(a, sw), reps=reps)
        ops.force_x3_gemm = False
        print(f'  {M:7d} {N:5d} {K:5d}: {us:9.1f} us {2 * M * N * K / us / 1e6:7.1f} TF | {us3:9.1f} us {2 * M * N * K / us3 / 1e6:7.1f} TF   x{us / us3:.2f}')
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
