"""Error of each cfg.compute_dtype against the REAL reference module's outputs (the committed goldens, parity mode):
    python tools/dtype_parity.py            (run on the MI355X box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_model as T  # noqa: E402


def main():
    for dt in ('fp32x3', 'fp32', 'bf16'):
        for case, cfgn, overrides in T.GOLDEN_CASES:
            g = T.gold(case)
            cfg = T.load_cfg(cfgn)
            cfg.update(overrides)
            cfg.update({'kpconv_ref_row_order': True, 'compute_dtype': dt})
            out, _ = T._run_product(cfg, T.seeded_sd(cfg), [g['src']], [g['tgt']])
            worst = {k: float(np.abs(out[k][0].cpu().numpy() - g[k]).max()) for k in ('src_kp_warped', 'tgt_kp_warped', 'src_overlap', 'tgt_overlap')}
            worst['pose'] = float(np.abs(out['pose'].cpu().numpy() - g['pose']).max())
            if 'src_feat_last' in g:
                for side in ('src', 'tgt'):
                    worst[f'{side}_feat_last'] = float(np.abs(out[f'{side}_feat'][0][-1].cpu().numpy() - g[f'{side}_feat_last']).max())
            print(f'{dt:7s} {case:22s} max {max(worst.values()):.2e}  ' + ' '.join(f'{k}={v:.1e}' for k, v in worst.items()))


if __name__ == '__main__':
    main()
