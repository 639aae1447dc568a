"""Achievable HBM bandwidth by access mix (torch streaming kernels as the probe): write-only, read-only, copy, 2 reads + 1 write.
Puts the write-heavy short-K GEMMs (1 read : 2 writes) and the InstanceNorm apply pass (2 reads : 1 write) in context."""
import torch


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    n = 1 << 30          # 1 Gi floats = 4 GiB per tensor (far beyond the 256 MB Infinity Cache)
    a = torch.empty(n, device='cuda'); b = torch.empty(n, device='cuda'); c = torch.empty(n, device='cuda')
    a.normal_(); b.normal_()
    gb = n * 4 / 1e9
    t = timed(lambda: c.fill_(1.0)); print(f'write only      : {gb / t / 1e3:.2f} TB/s')
    t = timed(lambda: a.sum()); print(f'read only (sum) : {gb / t / 1e3:.2f} TB/s')
    t = timed(lambda: c.copy_(a)); print(f'copy 1r:1w      : {2 * gb / t / 1e3:.2f} TB/s')
    t = timed(lambda: torch.add(a, b, out=c)); print(f'add 2r:1w       : {3 * gb / t / 1e3:.2f} TB/s')
    a3 = a[: n // 3 * 3].view(-1, 3)
    t = timed(lambda: torch.add(a, 1.0, out=c)); print(f'scale 1r:1w     : {2 * gb / t / 1e3:.2f} TB/s')


if __name__ == '__main__':
    main()
