"""GPU (-m gpu): BASELINE.json configs[4] -- a synthetic 100k + 100k-point pair.
  * every neighbour / pool table of the pyramid, full size, against the unmodified reference C++ (oracle/_ref: KD-tree, fast)
    through the canonical (d2, index) order; the same tables in PARITY mode equal the reference's element for element;
  * subsampling bit-exact against the (linear-time) C++ oracle at every level;
  * the forward against the CPU oracle restatement on the full stress pair (tables for the oracle come from oracle/_ref, the
    quadratic brute-force restatement being too slow at this size);
  * size-independent properties as a second line: invariants on every row, exact brute force on sampled queries, batch
    independence, orthonormal poses."""
import numpy as np
import pytest
import torch

from tests.util import canon_table, load_cfg, seeded_sd, seg_of, to_dev

pytestmark = pytest.mark.gpu


def _pair(points=100000):
    from regtr_amd.synthetic import synth_pair
    return synth_pair(77, points)


def _d2(q, s):
    d = q - s
    return ((np.float32(0) + d[:, 0] * d[:, 0]) + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]      # nanoflann.hpp:432-440 order


def test_stress_100k_preprocess_properties():
    from oracle import native
    from regtr_amd import ops
    src, tgt = _pair()
    assert len(src) > 90000 and len(tgt) > 70000
    pts = np.concatenate([src, tgt]); lens = np.array([len(src), len(tgt)], np.int32)
    r, K = np.float32(0.0625), 40
    grid = ops.CellGrid(to_dev(pts), seg_of(lens), len(pts), float(r))
    idx, cnt, _ = grid.query(to_dev(pts), seg_of(lens), len(pts), K, want_count=True)
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    n = len(pts)
    pad = idx == n
    # every row: self first, real entries in front of the padding, all inside the ball, ascending (d2, index), same cloud
    assert np.array_equal(idx[:, 0], np.arange(n))
    assert (np.diff(pad.astype(np.int8), axis=1) >= 0).all()
    assert np.array_equal((~pad).sum(1), np.minimum(cnt, K))
    sp = np.concatenate([pts, np.zeros((1, 3), np.float32)])
    d2 = ((np.float32(0) + (pts[:, None, 0] - sp[idx, 0]) ** 2) + (pts[:, None, 1] - sp[idx, 1]) ** 2) + (pts[:, None, 2] - sp[idx, 2]) ** 2
    r2 = r * r
    assert (d2[~pad] < r2).all()
    key = (d2.view(np.uint32).astype(np.int64) << 20) | idx         # (d2 bits, index) as one integer: idx < 2^20, d2 >= 0
    key[pad] = np.iinfo(np.int64).max
    assert (np.diff(key, axis=1) >= 0).all()
    cloud = (idx >= lens[0]).astype(np.int8)
    assert ((cloud == (np.arange(n) >= lens[0])[:, None]) | pad).all()
    # sampled queries: exact brute force in the reference's float32 arithmetic
    rng = np.random.default_rng(0)
    for q in rng.choice(n, 300, replace=False):
        lo, hi = (0, lens[0]) if q < lens[0] else (lens[0], n)
        dd = _d2(pts[q][None], pts[lo:hi])
        inside = np.nonzero(dd < r2)[0] + lo
        assert cnt[q] == len(inside)
        order = np.lexsort((inside, dd[inside - lo]))
        want = inside[order][:K]
        assert np.array_equal(idx[q, :len(want)], want)
    # voxel subsampling at full size: bit-exact against the C++ oracle (it is linear time)
    out, out_seg = ops.grid_subsample(to_dev(pts), seg_of(lens), n, 0.05)
    oseg = out_seg.cpu().numpy()
    ref_p, ref_l = native.grid_subsample(pts, lens, 0.05)
    assert np.array_equal(np.diff(oseg), ref_l)
    assert np.array_equal(out[:oseg[-1]].cpu().numpy().view(np.uint32), ref_p.view(np.uint32))


def test_stress_100k_forward_invariants():
    from regtr_amd import RegTR
    src, tgt = _pair()
    cfg = load_cfg('3dmatch')
    model = RegTR(cfg)
    model.load_state_dict(seeded_sd(cfg))
    model = model.cuda().eval()
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    one = model({'src_xyz': [s], 'tgt_xyz': [t]})
    two = model({'src_xyz': [s, s[:30000]], 'tgt_xyz': [t, t[:25000]]})        # the big pair next to an unrelated one
    assert one['pose'].shape == (6, 1, 3, 4) and torch.isfinite(one['pose']).all()
    R = one['pose'][:, 0, :, :3]
    assert (R @ R.transpose(1, 2) - torch.eye(3, device=R.device)).abs().max() < 1e-4
    assert torch.equal(one['src_kp'][0], two['src_kp'][0])                                  # preprocessing is per cloud
    assert (one['src_kp_warped'][0] - two['src_kp_warped'][0]).abs().max() < 1e-4          # pairs are independent
    assert (one['pose'][:, 0] - two['pose'][:, 0]).abs().max() < 1e-4
    assert 1000 < len(one['src_kp'][0]) < 8000                                              # ~2k tokens per cloud at this size


def test_two_stream_forward_equals_sequential():
    """A large batch reaches the encoder two ways -- pyramid on the current stream, or pyramid on the side stream with the level-0 blocks
    started early: the same kernels on the same data, identical outputs."""
    from regtr_amd import RegTR, regtr
    src, tgt = _pair()
    cfg = load_cfg('3dmatch')
    model = RegTR(cfg)
    model.load_state_dict(seeded_sd(cfg))
    model = model.cuda().eval()
    mk = lambda: {'src_xyz': [torch.from_numpy(src).cuda(), torch.from_numpy(src[:40000]).cuda()],
                  'tgt_xyz': [torch.from_numpy(tgt).cuda(), torch.from_numpy(tgt[:30000]).cuda()]}
    assert sum(len(x) for x in (src, tgt)) >= regtr.OVERLAP_MIN_POINTS
    prev = regtr.overlap_preprocessing
    try:
        regtr.overlap_preprocessing = False
        b0 = mk(); ref = model(b0)
        regtr.overlap_preprocessing = True
        b1 = mk(); two = model(b1)
    finally:
        regtr.overlap_preprocessing = prev
    assert torch.equal(two['pose'], ref['pose'])
    for k in ('src_kp', 'tgt_kp', 'src_kp_warped', 'tgt_overlap'):
        assert all(torch.equal(x, y) for x, y in zip(two[k], ref[k])), k
    for l in range(len(b0['kpconv_meta']['points'])):
        assert torch.equal(b1['kpconv_meta']['neighbors'][l], b0['kpconv_meta']['neighbors'][l])
        assert torch.equal(b1['kpconv_meta']['stack_lengths'][l], b0['kpconv_meta']['stack_lengths'][l])


def test_large_batch_paths_agree_with_per_op_paths():
    """Strip GEMMs, moments tails, packed-record gather and the two-stream forward (all gated on batch size) against the plain per-op
    paths, same weights and batch: they reorder float32 sums, nothing else -- poses / correspondences within 1e-4 (measured ~1e-5)."""
    from regtr_amd import RegTR, ops, regtr
    from regtr_amd.synthetic import synth_pair
    cfg = load_cfg('3dmatch')
    model = RegTR(cfg)
    model.load_state_dict(seeded_sd(cfg))
    model = model.cuda().eval()
    pairs = [synth_pair(500 + i, 20000) for i in range(8)]
    assert sum(len(s_) + len(t_) for s_, t_ in pairs) >= max(ops.STREAM_MIN_ROWS, regtr.OVERLAP_MIN_POINTS)
    mk = lambda: {'src_xyz': [torch.from_numpy(s_).cuda() for s_, _ in pairs], 'tgt_xyz': [torch.from_numpy(t_).cuda() for _, t_ in pairs]}
    prev = (ops.use_block_tail, ops.use_stream_gemm, ops.prenorm_gather, regtr.overlap_preprocessing)
    try:
        outs = []
        for on in (True, False):
            ops.use_block_tail = ops.use_stream_gemm = ops.prenorm_gather = regtr.overlap_preprocessing = on
            outs.append(model(mk()))
    finally:
        ops.use_block_tail, ops.use_stream_gemm, ops.prenorm_gather, regtr.overlap_preprocessing = prev
    new, old = outs
    assert all(torch.equal(a, b) for a, b in zip(new['src_kp'], old['src_kp']))
    assert (new['pose'] - old['pose']).abs().max() < 1e-4
    for k in ('src_kp_warped', 'tgt_kp_warped', 'src_overlap', 'tgt_overlap'):
        assert max(float((a - b).abs().max()) for a, b in zip(new[k], old[k])) < 1e-4, k
    scale = max(1.0, max(float(b.abs().max()) for b in old['src_feat_un']))
    assert max(float((a - b).abs().max()) for a, b in zip(new['src_feat_un'], old['src_feat_un'])) < 1e-4 * scale


def _ref_canonical_meta(pts_list, cfg):
    from oracle.canonical import canonical_meta
    return canonical_meta(pts_list, cfg)


def test_stress_100k_full_tables_and_forward_vs_oracle():
    """Full-size check of configs[4]: every table of the pyramid == the reference C++'s neighbour sets in canonical order,
    and the whole forward == the CPU oracle restatement (1e-4) on the 100k + 100k pair."""
    from oracle import native, regtr_ref
    from regtr_amd import RegTR
    if not native.have_ref():
        pytest.skip('oracle/_ref not present')
    src, tgt = _pair()
    cfg = load_cfg('3dmatch')
    sd = seeded_sd(cfg)
    model = RegTR(cfg)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    batch = {'src_xyz': [torch.from_numpy(src).cuda()], 'tgt_xyz': [torch.from_numpy(tgt).cuda()]}
    from regtr_amd import ops
    ops.SELF_QUERY_MIN_POINTS, keep = 0, ops.SELF_QUERY_MIN_POINTS        # conv tables through the cell-centric kernel (as in large batches)
    try:
        out = model(batch)
        torch.cuda.synchronize()
    finally:
        ops.SELF_QUERY_MIN_POINTS = keep
    meta = batch['kpconv_meta']
    rmeta = _ref_canonical_meta([src, tgt], cfg)
    for l in range(len(rmeta['points'])):
        assert torch.equal(meta['points'][l].cpu(), rmeta['points'][l]), l
        assert torch.equal(meta['neighbors'][l].cpu().long(), rmeta['neighbors'][l]), l
        if rmeta['pools'][l].numel():
            assert torch.equal(meta['pools'][l].cpu().long(), rmeta['pools'][l]), l
    with torch.no_grad():
        ref = regtr_ref.regtr_forward(sd, cfg, [src], [tgt], meta=rmeta)
    worst = {k: (out[k][0].cpu() - ref[k][0]).abs().max().item() for k in ('src_kp_warped', 'tgt_kp_warped', 'src_overlap', 'tgt_overlap')}
    worst['pose'] = (out['pose'].cpu() - ref['pose']).abs().max().item()
    print('stress 100k+100k forward, max abs diff vs oracle:', {k: f'{v:.2e}' for k, v in worst.items()})
    assert max(worst.values()) < 1e-4, worst


def test_stress_100k_parity_mode_tables_equal_reference():
    """Parity mode at stress size: level-0 subsampling and the level-0 / level-1 tables equal the unmodified reference C++
    element for element (100k-point KD-trees, libstdc++ order of 27k-voxel maps)."""
    from oracle import native
    from regtr_amd import cpp_wrappers
    if not native.have_ref():
        pytest.skip('oracle/_ref not present')
    src, tgt = _pair()
    pts = np.concatenate([src, tgt]); lens = np.array([len(src), len(tgt)], np.int32)
    with cpp_wrappers.reference_order():
        sub, sl = cpp_wrappers.grid_subsampling.subsample_batch(pts, lens, sampleDl=0.05)
        rsub, rsl = native.ref_subsample_batch(pts, lens, 0.05)
        assert np.array_equal(sl, rsl) and np.array_equal(sub.view(np.uint32), rsub.view(np.uint32))
        assert np.array_equal(cpp_wrappers.radius_neighbors.batch_query(pts, pts, lens, lens, radius=0.0625),
                              native.ref_batch_query(pts, pts, lens, lens, 0.0625))
        assert np.array_equal(cpp_wrappers.radius_neighbors.batch_query(sub, pts, sl, lens, radius=0.0625),
                              native.ref_batch_query(sub, pts, sl, lens, 0.0625))
