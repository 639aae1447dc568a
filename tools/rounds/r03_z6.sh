#!/bin/bash
python - <<'PY'
from regtr_amd import _lib
L=_lib.lib()
import torch; torch.zeros(1,device='cuda')
for cw,ar in ((4,2),(4,4),(2,2),(2,3),(2,4)):
    for st in (0,1): print('cw',cw,'ar',ar,'stats',st,'-> workgroups per CU', L.regtr_gemm_x3_strip_occupancy(cw,ar,st))
PY
