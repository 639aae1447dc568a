"""Compare-rate probe (see cmp_probe.hip): SIMD cycles per rank step, LATENCY (one wave per SIMD: 256 blocks x 256 threads on 256 CUs) and
THROUGHPUT (eight waves per SIMD: 2048 blocks)."""
import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'libcmp.so')
if not os.path.exists(so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so, os.path.join(here, 'cmp_probe.hip')])
lib = ctypes.CDLL(so)
keys = torch.randint(0, 2**62, (128,), device='cuda', dtype=torch.int64)
out = torch.empty(2048 * 256, device='cuda', dtype=torch.int32)
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
iters = 20000
for blocks, waves in ((256, 1), (2048, 8)):
    for mode, nm in enumerate(['u64 <', 'u32 <', 'f32 <', 'u64 order from two u32 compares', 'f64 <']):
        def run(): lib.cmp_probe(P(keys.data_ptr()), P(out.data_ptr()), mode, iters, blocks, P(st))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f'{waves} wave(s) per SIMD, mode {mode} {nm:34s}: {ms:8.3f} ms   {ms * 1e6 / iters / 32 / waves * 2.4:5.1f} SIMD cycles per key step and wave at 2.4 GHz')
