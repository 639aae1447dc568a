import os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo') else '.')
from regtr_amd import ops
ops.force_x3_gemm = True; ops.use_stream_gemm = False
for M, N, K in [(37888, 256, 3840), (37888, 1024, 256), (141056, 128, 1920), (37888, 256, 1024)]:
    a = torch.randn(M, K, device='cuda'); w = torch.randn(K, N, device='cuda') / K ** 0.5
    sw = ops.SplitWeight(w, 'kn')
    for _ in range(2): ops.gemm(a, sw)
    torch.cuda.synchronize()
