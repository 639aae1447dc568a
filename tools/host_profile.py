"""GPU: where the HOST time of a one-pair forward goes (cProfile over 200 forwards) and how busy the GPU is meanwhile.
    python tools/host_profile.py [pairs]"""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda', 0)
cfg, model, pairs, batch = bench.build_workload('3dmatch', n, 20000, False, 0, dev, 'fp32')
f = lambda: model({'src_xyz': list(batch['src_xyz']), 'tgt_xyz': list(batch['tgt_xyz'])})
for _ in range(10): f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): f()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'{n} pair(s): {t_all * 10:.3f} ms per forward wall, host returned after {t_issue * 10:.3f} ms per forward (enqueue-bound if equal)')
pr = cProfile.Profile(); pr.enable()
for _ in range(200): f()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
