import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regtr_amd import ops
dev='cuda'
ops.force_x3_gemm=True
for M,N,K in [(140568,512,256),(37524,256,3840)]:
    a=torch.randn(M,K,device=dev); b=torch.randn(K,N,device=dev); sw=ops.SplitWeight(b,'kn')
    for _ in range(3): ops.gemm(a,sw)
    torch.cuda.synchronize()
