"""CPU: pins the oracle (oracle/) against the reference -- golden fixtures made by the real reference
(oracle/make_golden.py) and, where oracle/_ref is present, the unmodified reference C++ itself."""
import numpy as np
import pytest
import torch

from oracle import native, regtr_ref
from tests.util import canon_rows, gold, load_cfg, seeded_sd, synth_cloud


@pytest.mark.parametrize('case', ['modelnet', '3dmatch_crop'])
def test_grid_subsample_vs_reference_fixture(case):
    g = gold(f'native_{case}')
    pts, lens = g['pts'], g['lens']
    # reference row order (libstdc++ unordered_map iteration): bit exact, row for row
    op, ol = native.grid_subsample(pts, lens, float(g['dl']), ref_order=True)
    assert np.array_equal(ol, g['sub_lens'])
    assert np.array_equal(op.view(np.uint32), g['sub_pts'].view(np.uint32))
    # canonical order: same multiset of barycentres per cloud, bit exact
    cp, cl = native.grid_subsample(pts, lens, float(g['dl']))
    assert np.array_equal(cl, ol)
    o = 0
    for n in cl:
        a, b = cp[o:o + n], op[o:o + n]
        assert np.array_equal(a[np.lexsort(a.T[::-1])].view(np.uint32), b[np.lexsort(b.T[::-1])].view(np.uint32))
        o += n


@pytest.mark.parametrize('case', ['modelnet', '3dmatch_crop'])
def test_radius_vs_reference_fixture(case):
    g = gold(f'native_{case}')
    pts, lens, r = g['pts'], g['lens'], float(g['radius'])
    ref = g['neighbors']                                   # untruncated (Nq, max_count), reference tie order
    W = ref.shape[1]
    idx, cnt, tie = native.radius_neighbors(pts, pts, lens, lens, r, W)
    assert cnt.max() == W                                  # row width = max count (neighbors.cpp:290-293)
    assert np.array_equal((ref != len(pts)).sum(1), cnt)
    assert np.array_equal(canon_rows(ref, pts, pts, len(pts)), idx)
    # pooled queries (subsampled points vs full cloud)
    pool = g['pools']
    idx2, cnt2, _ = native.radius_neighbors(g['sub_pts'], pts, g['sub_lens'], lens, r, pool.shape[1])
    assert np.array_equal(canon_rows(pool, g['sub_pts'], pts, len(pts)), idx2)
    # truncation at K: the first K canonical entries, tie rows flagged
    K = max(2, W // 2)
    idxk, cntk, tiek = native.radius_neighbors(pts, pts, lens, lens, r, K)
    assert np.array_equal(idxk, idx[:, :K]) and np.array_equal(cntk, cnt)


@pytest.mark.skipif(not native.have_ref(), reason='oracle/_ref not built (needs /root/reference)')
def test_native_vs_unmodified_reference_cpp():
    rng = np.random.default_rng(3)
    clouds = [synth_cloud(rng, n, lattice=0.006) for n in (900, 1, 1500)]
    pts = np.concatenate(clouds); lens = np.array([len(c) for c in clouds], np.int32)
    for dl in (0.05, 0.11):
        rp, rl = native.ref_subsample_batch(pts, lens, dl)
        op, ol = native.grid_subsample(pts, lens, dl, ref_order=True)
        assert np.array_equal(rl, ol) and np.array_equal(rp.view(np.uint32), op.view(np.uint32))
    ref = native.ref_batch_query(pts, pts, lens, lens, 0.0625)
    idx, cnt, _ = native.radius_neighbors(pts, pts, lens, lens, 0.0625, ref.shape[1])
    assert np.array_equal(canon_rows(ref, pts, pts, len(pts)), idx)
    # the reference's own brute-force variant (neighbors.cpp:125-208) breaks ties by index: equal without canonicalising
    ordered = native.ref_batch_query(pts, pts, lens, lens, 0.0625, ordered=True)
    assert np.array_equal(ordered, idx)


POSTNORM = {'pre_norm': False, 'sa_val_has_pos_emb': False, 'ca_val_has_pos_emb': True}    # oracle/make_golden.py


@pytest.mark.parametrize('case,cfgn,overrides', [('modelnet_demo', 'modelnet', {}), ('3dmatch_crop', '3dmatch', {}),
                                                 ('modelnet_630', 'modelnet', {}), ('3dmatch_home_at', '3dmatch', {}),
                                                 ('modelnet_postnorm', 'modelnet', POSTNORM),
                                                 ('modelnet_attn_head', 'modelnet', {'direct_regress_coor': False})])
def test_float_restatement_vs_reference_golden(case, cfgn, overrides):
    """oracle/regtr_ref.py driven in the REFERENCE's row order reproduces the reference module's outputs
    (pre-norm layers of both shipped configs, the post-norm forward_post variant, the attention CorrespondenceDecoder)."""
    if not native.have_ref():
        pytest.skip('needs oracle/_ref for the reference row order')
    g = gold(case)
    cfg = load_cfg(cfgn)
    cfg.update(overrides)
    sd = seeded_sd(cfg)
    with torch.no_grad():
        out = regtr_ref.regtr_forward(sd, cfg, [g['src']], [g['tgt']], use_ref_cpp=True)
    assert np.array_equal(out['src_kp'][0].numpy(), g['src_kp'])
    for k in ('src_kp_warped', 'tgt_kp_warped', 'src_overlap', 'tgt_overlap'):
        assert np.abs(out[k][0].numpy() - g[k]).max() < 5e-5, k
    assert np.abs(out['pose'].numpy() - g['pose']).max() < 5e-5
    if 'src_feat_last' in g:
        assert np.abs(out['src_feat'][0][-1].numpy() - g['src_feat_last']).max() < 5e-5


def test_float_restatement_vs_reference_golden_batch2():
    """B = 2 ragged pairs in one forward: the reference pads the tokens to (N_max, B, D) and masks the padded keys (regtr.py:147-172,
    transformers.py:197-226); the restatement runs every pair on its own packed tokens.  Same outputs (golden: the real module)."""
    if not native.have_ref():
        pytest.skip('needs oracle/_ref for the reference row order')
    g = gold('3dmatch_crop_b2')
    cfg = load_cfg('3dmatch')
    sd = seeded_sd(cfg)
    with torch.no_grad():
        out = regtr_ref.regtr_forward(sd, cfg, [g['src_0'], g['src_1']], [g['tgt_0'], g['tgt_1']], use_ref_cpp=True)
    assert np.array_equal(out['kpconv_meta']['points'][-1].numpy(), g['points_last'])
    assert np.array_equal(out['kpconv_meta']['neighbors'][-1].numpy(), g['neighbors_last'])
    for b in range(2):
        assert np.array_equal(out['src_kp'][b].numpy(), g[f'src_kp_{b}']) and np.array_equal(out['tgt_kp'][b].numpy(), g[f'tgt_kp_{b}'])
        for k in ('src_kp_warped', 'tgt_kp_warped', 'src_overlap', 'tgt_overlap'):
            assert np.abs(out[k][b].numpy() - g[f'{k}_{b}']).max() < 5e-5, (k, b)
    assert np.abs(out['pose'].numpy() - g['pose']).max() < 5e-5


@pytest.mark.parametrize('case', ['3dmatch_kitchen', '3dmatch_hotel', '3dmatch_home_at'])
def test_reference_order_pyramid_matches_golden(case):
    """Level points of the shipped 3DMatch pairs in the reference's order are bit exact at every level."""
    g = gold(case)
    pts = np.concatenate([g['src'], g['tgt']]); lens = np.array([len(g['src']), len(g['tgt'])], np.int32)
    dl = 0.05
    for l in (1, 2, 3):
        pts, lens = native.grid_subsample(pts, lens, dl, ref_order=True)
        assert np.array_equal(pts.view(np.uint32), g[f'points_{l}'].view(np.uint32))
        assert np.array_equal(lens, g[f'lens_{l}'])
        dl *= 2


def test_compute_overlaps_restatement_vs_reference_golden():
    """oracle compute_overlaps (kpconv.py:540-566) == the reference's own function run on the reference Preprocessor's pyramid
    (tests/golden/overlaps_3dmatch_crop.npz, made by oracle/make_golden.py)."""
    if not native.have_ref():
        pytest.skip('needs oracle/_ref for the reference row order')
    g = gold('overlaps_3dmatch_crop')
    cfg = load_cfg('3dmatch')
    meta = regtr_ref.preprocess([g['src'], g['tgt']], cfg, use_ref_cpp=True)
    pyr = regtr_ref.compute_overlaps({'src_overlap': [torch.from_numpy(g['src_overlap'])], 'tgt_overlap': [torch.from_numpy(g['tgt_overlap'])],
                                      'kpconv_meta': meta})
    for p in range(4):
        assert np.allclose(pyr[f'pyr_{p}'].numpy(), g[f'pyr_{p}'], rtol=0, atol=1e-7, equal_nan=True), p


def test_ball_query_first_k_properties():
    """The restatement of the reference PreprocessorGPU's neighbour rule (pytorch3d ball_query, kpconv.py:261-288): every row holds
    the smallest K indices of the ball, ascending, same cloud only, padded with the support count."""
    from oracle import regtr_ref
    rng = np.random.default_rng(3)
    s = rng.uniform(0, 1, (400, 3)).astype(np.float32); s[200:] += 3.0
    q = s[::2].copy()
    lens, ql = np.array([200, 200]), np.array([100, 100])
    r, K = 0.35, 8
    t = regtr_ref.ball_query_first_k(q, s, ql, lens, r, K)
    assert t.shape == (200, K)
    for i in range(len(q)):
        lo, hi = (0, 200) if i < 100 else (200, 400)
        d2 = ((s[lo:hi] - q[i]) ** 2).sum(1)
        ball = np.nonzero(d2 < np.float32(r) ** 2)[0] + lo
        want = list(ball[:K]) + [len(s)] * (K - min(K, len(ball)))
        assert list(t[i]) == want
    assert (np.diff(np.where(t < len(s), t, len(s) + np.arange(K)), axis=1) > 0).all()      # ascending inside a row


def test_pair_tables_cut_from_a_batch_equal_the_pair_alone():
    """oracle/canonical.py: the rows of one pair cut out of a batched kpconv_meta and re-indexed == canonical_meta of that pair alone
    (preprocessing is per cloud) -- the helper the bench-batch GPU test relies on, checked here on the CPU restatement."""
    from oracle import canonical
    from regtr_amd.synthetic import synth_pair
    cfg = load_cfg('3dmatch')
    pairs = [synth_pair(i, 3000, overlap='lomatch' if i else None) for i in range(3)]
    meta = regtr_ref.preprocess([s for s, _ in pairs] + [t for _, t in pairs], cfg)
    for b in range(3):
        cm = canonical.canonical_meta(list(pairs[b]), cfg)
        pt = canonical.pair_tables_of_batch(meta, 3, b)
        for l in range(len(cm['points'])):
            assert np.array_equal(pt['points'][l], cm['points'][l].numpy())
            assert np.array_equal(pt['neighbors'][l], cm['neighbors'][l].numpy()), (b, l)
            if pt['pools'][l] is not None:
                assert np.array_equal(pt['pools'][l], cm['pools'][l].numpy()), (b, l)


def test_grid_subsample_floor_keys_restatement():
    """oracle_grid_subsample_keyed (PreprocessorGPU's voxel rule, kpconv.py:213-240; parity unpinned: MinkowskiEngine is absent) against
    a numpy statement of the same rule: voxel of p = floor(p / dl), unweighted mean of the members; and against the numbers SURVEY.md
    section 1 / VERDICT r03 measured on the red-kitchen pair (10 088 level-1 points, the CPU rule gives 9 977)."""
    g = gold('3dmatch_kitchen')
    pts = np.concatenate([g['src'], g['tgt']]).astype(np.float32)
    lens = np.array([len(g['src']), len(g['tgt'])], np.int32)
    sub, sl = native.grid_subsample(pts, lens, 0.05, key_mode=1)
    assert sl.tolist() == [5170, 4918] and native.grid_subsample(pts, lens, 0.05)[1].tolist() == [5091, 4886]
    off = 0
    row = 0
    for n in lens:
        p = pts[off:off + n]
        k = np.floor(p / np.float32(0.05)).astype(np.int64)
        uniq, first, inv = np.unique(k, axis=0, return_index=True, return_inverse=True)
        order = np.argsort(first)                                  # first-appearance order
        rank = np.empty(len(uniq), np.int64); rank[order] = np.arange(len(uniq))
        sums = np.zeros((len(uniq), 3), np.float64); cnt = np.zeros(len(uniq))
        np.add.at(sums, rank[inv.ravel()], p.astype(np.float64)); np.add.at(cnt, rank[inv.ravel()], 1)
        want = sums / cnt[:, None]
        got = sub[row:row + len(uniq)]
        assert np.abs(got - want).max() < 2e-6                     # float32 in-order sums vs float64 means
        assert np.array_equal(np.floor(got / np.float32(0.05)).astype(np.int64)[cnt == 1], uniq[order][cnt == 1])
        row += len(uniq); off += n


def test_compute_overlap_restatement_vs_reference_function_golden():
    """oracle compute_overlap == the REFERENCE's own function (utils/pointcloud.py:8-65, run from /root/reference by
    oracle/make_golden_overlap.py over a scipy-cKDTree stand-in for open3d's KDTreeFlann): both has-correspondence masks and the mutual
    correspondence list, on a jittered copy, a partial overlap, disjoint clouds and real fragments."""
    g = gold('overlap_pairs')
    for i in range(int(g['n_cases'])):
        hs, ht, corr = regtr_ref.compute_overlap(g[f'src_{i}'], g[f'tgt_{i}'], float(g[f'radius_{i}']))
        assert np.array_equal(hs, g[f'has_src_{i}']) and np.array_equal(ht, g[f'has_tgt_{i}']), i
        assert np.array_equal(corr, g[f'corr_{i}']), i
