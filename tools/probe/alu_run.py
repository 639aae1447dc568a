"""f32-MFMA / VALU overlap probe (see alu_probe.hip).  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libalu.so alu_probe.hip"""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'libalu.so'))
out = torch.empty(256 * 512, device='cuda')
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
names = ['f32 MFMA only (4/iter/wave)', 'VALU only (32 FMA/iter/wave)', 'both, same wave', 'MFMA waves + VALU waves (same SIMD)',
         'bf16 MFMA 16x16x32 only', 'bf16 MFMA waves + VALU waves', 'f32 MFMA waves alone (1 per SIMD)', 'VALU waves alone (1 per SIMD)',
         'bf16 MFMA waves alone (1 per SIMD)']
iters = 20000
for mode, nm in enumerate(names):
    def run(): lib.alu_probe(P(out.data_ptr()), mode, iters, 256, P(st))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f'mode {mode} {nm:40s}: {ms:8.3f} ms   {ms * 1e6 / iters:7.1f} ns / iteration')
