"""Large-batch paths (strip GEMMs, moments tails, packed-record gather, two-stream forward) against the plain per-op paths on the same
batch and weights: max |difference| of every output.   python tools/path_parity.py [--pairs 16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from regtr_amd import RegTR, load_config, ops, regtr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=16)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = load_config(os.path.join(bench.ROOT, 'regtr_amd', 'conf', '3dmatch.yaml'))
    torch.manual_seed(0)
    model = RegTR(cfg).to(dev).eval()
    pairs = [bench.synth_pair(1000 + i, 20000) for i in range(args.pairs)]
    mk = lambda: {'src_xyz': [torch.from_numpy(s).to(dev) for s, _ in pairs], 'tgt_xyz': [torch.from_numpy(t).to(dev) for _, t in pairs]}

    def run(on):
        ops.use_block_tail = ops.use_stream_gemm = ops.prenorm_gather = regtr.overlap_preprocessing = on
        return model(mk())

    new, old = run(True), run(False)
    worst = {'pose': float((new['pose'] - old['pose']).abs().max())}
    for k in ('src_kp_warped', 'tgt_kp_warped', 'src_overlap', 'tgt_overlap', 'src_feat_un', 'src_feat'):
        worst[k] = max(float((a - b).abs().max()) for a, b in zip(new[k], old[k]))
    assert all(torch.equal(a, b) for a, b in zip(new['src_kp'], old['src_kp']))
    print(f'{args.pairs} pairs, large-batch paths vs per-op paths: ' + '  '.join(f'{k} {v:.2e}' for k, v in worst.items()))


if __name__ == '__main__':
    main()
