"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the REAL reference
(/root/reference) in this container.  Re-run with:  python -m oracle.make_golden

Cases
  modelnet_demo   : data/modelnet_demo_data/modelnet_test_2_{0,1}.ply, conf/modelnet.yaml
  3dmatch_crop    : 1.2 m radius crops of the red-kitchen pair (cloud_bin_0/5), conf/3dmatch.yaml
  3dmatch_kitchen : the full red-kitchen pair (18 977 + 19 084 pts), conf/3dmatch.yaml
  modelnet_postnorm : the ModelNet pair with pre_norm: False (forward_post), sa_val_has_pos_emb: False
  modelnet_attn_head : the ModelNet pair with direct_regress_coor: False (attention CorrespondenceDecoder)
  3dmatch_crop_b2 : TWO ragged red-kitchen crop pairs in ONE forward (B = 2): the reference's padded (N_max, B, D) tokens +
                    key_padding_mask path (regtr.py:147-172, transformers.py:197-226) against the packed-token path here
  3dmatch_hotel   : demo.py example 1, sun3d-hotel_umd-maryland_hotel3 cloud_bin_8 / 15 (demo.py:31-34)
  3dmatch_home_at : demo.py example 2, sun3d-home_at-home_at_scan1_2013_jan_1 cloud_bin_38 / 41 (demo.py:35-38): the dense one --
                    22.7 % of its level-0 balls hold more than K = 40 supports (2.5x the kitchen's truncation / tie pressure)
  modelnet_630    : demo.py example 4, modelnet_test_630_{0,1}.ply (demo.py:44-47)
  (with modelnet_demo = example 3 and 3dmatch_kitchen = example 0 these are all five pairs the reference ships)
Each file holds the float32 inputs, the reference module's outputs (reference row order) with
weights = oracle.seeded_weights.seeded_state_dict(cfg, seed=0), and the reference C++'s
per-level points / stack lengths.  `native_*` files hold raw outputs of the reference C++ ops.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import native, ref_loader, seeded_weights   # noqa: E402
from regtr_amd.kernel_points import K015_CENTER         # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
DATA = os.path.join(ref_loader.REF_ROOT, 'data')


def read_ply_xyz(path):
    raw = open(path, 'rb').read()
    head, body = raw.split(b'end_header\n', 1)
    n = int([l for l in head.split(b'\n') if l.startswith(b'element vertex')][0].split()[-1])
    props = [l.split() for l in head.split(b'\n') if l.startswith(b'property')]
    dt = np.dtype([(p[2].decode(), '<f8' if p[1] in (b'double', b'float64') else '<f4') for p in props])
    a = np.frombuffer(body, dtype=dt, count=n)
    return np.stack([a['x'], a['y'], a['z']], 1).astype(np.float32)


def load_pth(path):
    return np.asarray(torch.load(path, weights_only=False))[:, :3].astype(np.float32)


def run_case(name, cfg_name, src, tgt, with_feats=False, overrides=None):
    cfg = ref_loader.load_cfg(cfg_name)
    for k, v in (overrides or {}).items():
        cfg[k] = v
    model = ref_loader.build_model(cfg, 0)
    sd = seeded_weights.seeded_state_dict(cfg, 0, K015_CENTER)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), 'state_dict names differ from the reference'
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd, strict=True)
    batch = {'src_xyz': [torch.from_numpy(src)], 'tgt_xyz': [torch.from_numpy(tgt)]}
    with torch.no_grad():
        out = model(batch)
    meta = batch['kpconv_meta']
    g = {'src': src, 'tgt': tgt, 'pose': out['pose'].numpy()}
    for k in ('src_kp', 'tgt_kp', 'src_kp_warped', 'tgt_kp_warped', 'src_overlap', 'tgt_overlap'):
        g[k] = out[k][0].numpy()
    if with_feats:
        g['src_feat_un'] = out['src_feat_un'][0].numpy()
        g['tgt_feat_un'] = out['tgt_feat_un'][0].numpy()
        g['src_feat_last'] = out['src_feat'][0][-1].numpy()
        g['tgt_feat_last'] = out['tgt_feat'][0][-1].numpy()
    for l, (p, s) in enumerate(zip(meta['points'], meta['stack_lengths'])):
        if l > 0:
            g[f'points_{l}'] = p.numpy()
        g[f'lens_{l}'] = s.numpy().astype(np.int32)
    # neighbour tables of the coarsest level only (small); earlier ones are checked live vs oracle/_ref
    g['neighbors_last'] = meta['neighbors'][-1].numpy().astype(np.int32)
    np.savez_compressed(os.path.join(GOLD, f'{name}.npz'), **g)
    print(name, {k: v.shape for k, v in g.items() if k.startswith('lens') or k == 'pose'},
          [int(s.sum()) for s in meta['stack_lengths']])
    print('  pose[-1]:\n', g['pose'][-1, 0])


def run_batch_case(name, cfg_name, pairs):
    """B > 1: the reference module on several ragged pairs in one forward (pad_sequence + key_padding_mask, regtr.py:147-166)."""
    cfg = ref_loader.load_cfg(cfg_name)
    model = ref_loader.build_model(cfg, 0)
    model.load_state_dict(seeded_weights.seeded_state_dict(cfg, 0, K015_CENTER), strict=True)
    batch = {'src_xyz': [torch.from_numpy(s) for s, _ in pairs], 'tgt_xyz': [torch.from_numpy(t) for _, t in pairs]}
    with torch.no_grad():
        out = model(batch)
    meta = batch['kpconv_meta']
    g = {'pose': out['pose'].numpy()}
    for b, (s, t) in enumerate(pairs):
        g[f'src_{b}'], g[f'tgt_{b}'] = s, t
        for k in ('src_kp', 'tgt_kp', 'src_kp_warped', 'tgt_kp_warped', 'src_overlap', 'tgt_overlap'):
            g[f'{k}_{b}'] = out[k][b].numpy()
    for l, s in enumerate(meta['stack_lengths']):
        g[f'lens_{l}'] = s.numpy().astype(np.int32)
    g['points_last'] = meta['points'][-1].numpy()
    g['neighbors_last'] = meta['neighbors'][-1].numpy().astype(np.int32)
    np.savez_compressed(os.path.join(GOLD, f'{name}.npz'), **g)
    print(name, 'pose', g['pose'].shape, [s.tolist() for s in meta['stack_lengths']])


def overlaps_case(name, cfg_name, src, tgt):
    """The reference's own compute_overlaps (kpconv.py:540-566) on the reference Preprocessor's pyramid with seeded random
    level-0 masks -> tests/golden/overlaps_<name>.npz."""
    ref = ref_loader.load()
    cfg = ref_loader.load_cfg(cfg_name)
    pre = ref.kpconv.Preprocessor(cfg)
    meta = pre([torch.from_numpy(src), torch.from_numpy(tgt)])
    rng = np.random.default_rng(5)
    so, to = rng.random(len(src)) < 0.6, rng.random(len(tgt)) < 0.3
    batch = {'src_overlap': [torch.from_numpy(so)], 'tgt_overlap': [torch.from_numpy(to)], 'kpconv_meta': meta}
    pyr = ref.kpconv.compute_overlaps(batch)
    np.savez_compressed(os.path.join(GOLD, f'overlaps_{name}.npz'), src=src, tgt=tgt, src_overlap=so, tgt_overlap=to,
                        **{k: v.numpy() for k, v in pyr.items()})
    print('overlaps', name, {k: (tuple(v.shape), float(np.nanmean(v.numpy()))) for k, v in pyr.items()})


def native_case(name, pts, lens, dl, radius):
    rp, rl = native.ref_subsample_batch(pts, lens, dl)
    nb = native.ref_batch_query(pts, pts, lens, lens, radius)
    pool = native.ref_batch_query(rp, pts, rl, lens, radius)
    np.savez_compressed(os.path.join(GOLD, f'native_{name}.npz'), pts=pts, lens=lens, dl=np.float32(dl),
                        radius=np.float32(radius), sub_pts=rp, sub_lens=rl, neighbors=nb, pools=pool)
    print('native', name, rp.shape, rl, nb.shape, pool.shape)


def main():
    os.makedirs(GOLD, exist_ok=True)
    native.build()
    m0 = read_ply_xyz(os.path.join(DATA, 'modelnet_demo_data', 'modelnet_test_2_0.ply'))
    m1 = read_ply_xyz(os.path.join(DATA, 'modelnet_demo_data', 'modelnet_test_2_1.ply'))
    k0 = load_pth(os.path.join(DATA, 'indoor/test/7-scenes-redkitchen/cloud_bin_0.pth'))
    k5 = load_pth(os.path.join(DATA, 'indoor/test/7-scenes-redkitchen/cloud_bin_5.pth'))

    def crop(p, r):
        c = np.median(p, 0)
        return p[np.linalg.norm(p - c, axis=1) < r]
    c0, c5 = crop(k0, 0.9), crop(k5, 0.9)

    native_case('modelnet', np.concatenate([m0, m1]), np.array([len(m0), len(m1)], np.int32), 0.06, 0.0825)
    n0, n5 = crop(k0, 0.5), crop(k5, 0.5)
    native_case('3dmatch_crop', np.concatenate([n0, n5]), np.array([len(n0), len(n5)], np.int32), 0.05, 0.0625)
    overlaps_case('3dmatch_crop', '3dmatch', c0, c5)
    if os.environ.get('GOLDEN_ONLY') == 'overlaps':
        return
    if os.environ.get('GOLDEN_ONLY') in (None, 'batch'):
        run_batch_case('3dmatch_crop_b2', '3dmatch', [(c0, c5), (crop(k0, 0.6), crop(k5, 0.7))])
        if os.environ.get('GOLDEN_ONLY') == 'batch':
            return
    if os.environ.get('GOLDEN_ONLY') in (None, 'demo'):
        # the reference's remaining shipped pairs (demo.py:26-49, examples 1, 2, 4)
        ind = os.path.join(DATA, 'indoor', 'test')
        run_case('3dmatch_hotel', '3dmatch', load_pth(os.path.join(ind, 'sun3d-hotel_umd-maryland_hotel3/cloud_bin_8.pth')),
                 load_pth(os.path.join(ind, 'sun3d-hotel_umd-maryland_hotel3/cloud_bin_15.pth')))
        run_case('3dmatch_home_at', '3dmatch', load_pth(os.path.join(ind, 'sun3d-home_at-home_at_scan1_2013_jan_1/cloud_bin_38.pth')),
                 load_pth(os.path.join(ind, 'sun3d-home_at-home_at_scan1_2013_jan_1/cloud_bin_41.pth')))
        run_case('modelnet_630', 'modelnet', read_ply_xyz(os.path.join(DATA, 'modelnet_demo_data', 'modelnet_test_630_0.ply')),
                 read_ply_xyz(os.path.join(DATA, 'modelnet_demo_data', 'modelnet_test_630_1.ply')))
        if os.environ.get('GOLDEN_ONLY') == 'demo':
            return
    run_case('modelnet_demo', 'modelnet', m0, m1)
    run_case('3dmatch_crop', '3dmatch', c0, c5, with_feats=True)
    run_case('3dmatch_kitchen', '3dmatch', k0, k5)
    # config variant: post-norm encoder layers (forward_post, transformers.py:121-181), values do not carry the pos-emb
    run_case('modelnet_postnorm', 'modelnet', m0, m1, overrides={'pre_norm': False, 'sa_val_has_pos_emb': False,
                                                                  'ca_val_has_pos_emb': True})
    # config variant: attention CorrespondenceDecoder instead of the MLP head (regtr.py:299-396)
    run_case('modelnet_attn_head', 'modelnet', m0, m1, overrides={'direct_regress_coor': False})


if __name__ == '__main__':
    main()
