"""Diagnostic: the real-fragment batch (bench.py --real) against the CPU oracle, pair by pair, under the A/B switches of the dispatch.
    REGTR_DEV=1 REGTR_BLOCK_TAIL=0 python tools/real_diag.py [--pairs 9] [--check 4 8] [--dtype fp32]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import canonical, regtr_ref  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--pairs', type=int, default=9)
ap.add_argument('--check', type=int, nargs='+', default=[4, 8])
ap.add_argument('--dtype', default='fp32')
ap.add_argument('--head', default=None)
ap.add_argument('--synthetic', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda', 0)
cfg, model, pairs, batch = bench.build_workload('3dmatch', args.pairs, 20000, False, 0, dev, args.dtype, real=not args.synthetic, head_init=args.head)
out = model({'src_xyz': list(batch['src_xyz']), 'tgt_xyz': list(batch['tgt_xyz'])})
torch.cuda.synchronize()
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
torch.set_num_threads(16)
sw = {k: v for k, v in os.environ.items() if k.startswith('REGTR_') and k != 'REGTR_DEV'}
for b in args.check:
    s, t = pairs[b]
    with torch.no_grad():
        ref = regtr_ref.regtr_forward(sd, cfg, [s], [t], meta=canonical.canonical_meta([s, t], cfg))
    single = model({'src_xyz': [batch['src_xyz'][b]], 'tgt_xyz': [batch['tgt_xyz'][b]]})
    row = {}
    for k in ('src_feat_un', 'tgt_feat_un', 'src_kp_warped', 'tgt_kp_warped'):
        sc = max(1.0, float(ref[k][0].abs().max())) if 'feat' in k else 1.0
        row[k] = (float((out[k][b].cpu() - ref[k][0]).abs().max()) / sc, float((single[k][0].cpu() - ref[k][0]).abs().max()) / sc)
    print(f'{sw} dtype {args.dtype} pair {b}: (batch vs oracle, single vs oracle) ' + ' '.join(f'{k} {v[0]:.2e}/{v[1]:.2e}' for k, v in row.items()), flush=True)
