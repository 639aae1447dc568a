"""GPU: wall time of every forward of a long run (is a step's time a function of how long the chip has been busy?).
    python tools/step_times.py [pairs] [steps] [real] [nosync]   (nosync: forwards enqueued back to back, timed in groups of five)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
real, nosync = 'real' in sys.argv[3:], 'nosync' in sys.argv[3:]
dev = torch.device('cuda:0')
cfg, model, prs, batch = bench.build_workload('3dmatch', pairs, 20000, False, 0, dev, 'fp32', real=real)
ts = []
group = 5 if nosync else 1
with torch.no_grad():
    for i in range(0, steps, group):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(group):
            model({'src_xyz': batch['src_xyz'], 'tgt_xyz': batch['tgt_xyz']})
        torch.cuda.synchronize(); ts += [(time.perf_counter() - t0) * 1e3 / group] * group
print('peak GiB', round(torch.cuda.max_memory_allocated() / 2**30, 2), 'reserved GiB', round(torch.cuda.memory_reserved() / 2**30, 2))
ts = np.array(ts)
print(f'{pairs} pairs per forward, {steps} forwards, ms per forward:')
for i in range(0, steps, 10):
    print(f'  forwards {i:3d}-{min(i + 9, steps - 1):3d}: ' + ' '.join(f'{t:6.1f}' for t in ts[i:i + 10]))
try:
    import subprocess
    print(subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp'], capture_output=True, text=True, timeout=20).stdout[-1500:])
except Exception as e:
    print('rocm-smi:', e)
