import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'libprobe.so'))
M = 2415616
A = torch.randn(M, 64, device='cuda'); C = torch.empty(M, 128, device='cuda'); sink = torch.zeros(M, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
P = ctypes.c_void_p
for name, fn, gb in (('read  fragment pattern', lambda: lib.probe_read_frag(P(A.data_ptr()), M, P(sink.data_ptr()), P(st)), M * 256 / 1e9),
                     ('read  coalesced       ', lambda: lib.probe_read_coal(P(A.data_ptr()), M, P(sink.data_ptr()), P(st)), M * 256 / 1e9),
                     ('write fragment pattern', lambda: lib.probe_write_frag(P(C.data_ptr()), M, P(st)), M * 512 / 1e9),
                     ('write coalesced float4', lambda: lib.probe_write_coal(P(C.data_ptr()), M, P(st)), M * 512 / 1e9)):
    t = timed(fn)
    print(f'{name}: {t * 1e6:7.0f} us  {gb / t / 1e3:.2f} TB/s')
gb = M * (256 + 512) / 1e9
for waves, ldsb, spin in ((4, 0, 0), (4, 20000, 0), (4, 40000, 0), (8, 65536, 0), (4, 0, 500), (4, 0, 2000), (8, 65536, 500), (8, 65536, 2000), (4, 40000, 2000)):
    t = timed(lambda: lib.probe_rw(P(A.data_ptr()), P(C.data_ptr()), M, waves, ldsb, spin, P(st)))
    print(f'read+write one-shot, {waves} waves/WG, {ldsb:6d} B LDS/WG, {spin:4d} dependent FMAs: {t * 1e6:7.0f} us  {gb / t / 1e3:.2f} TB/s')
Wt = torch.randint(0, 2 ** 15, (3 * 128 * 64,), dtype=torch.int16, device='cuda')
stat = torch.empty((M // 32 + 8) * 128 * 2, dtype=torch.float64, device='cuda')
names = ['8 waves, x3, block by block', '8 waves, x3, interleaved blocks', '8 waves, 1 plane, interleaved', '8 waves, x3, interleaved + f64 stats',
         '4 waves, x3, interleaved', '4 waves, 1 plane, interleaved']
for vnt, nm in enumerate(names):
    t = timed(lambda: lib.probe_strip(P(A.data_ptr()), P(Wt.data_ptr()), P(C.data_ptr()), P(stat.data_ptr()), M, vnt, P(st)))
    print(f'strip GEMM 64->128 [{nm}]: {t * 1e6:7.0f} us  {gb / t / 1e3:.2f} TB/s')
