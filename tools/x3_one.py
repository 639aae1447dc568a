"""GPU: one split-GEMM shape in a loop (for rocprofv3 --pmc passes): python tools/x3_one.py M N K [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regtr_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ops.force_x3_gemm = True; ops.use_stream_gemm = False
a = torch.randn(M, K, device='cuda'); w = torch.randn(K, N, device='cuda') / K ** 0.5
sw = ops.SplitWeight(w, 'kn')
with ops.f16_pair(ops.f16_pair_default):      # REGTR_F16_PAIR=0: the bf16 six-term kernel
    for _ in range(reps): ops.gemm(a, sw)
torch.cuda.synchronize()
