"""GPU (-m gpu): every HIP kernel, called through the C ABI, against the CPU oracle on the same seeded inputs."""
import math

import numpy as np
import pytest
import torch

from tests.util import gold, load_cfg, seeded_sd, seg_of, synth_cloud, to_dev

pytestmark = pytest.mark.gpu


def _ops():
    from regtr_amd import ops
    return ops


# ------------------------------------------------------------------------------------------------ GEMM / MFMA layout
def test_gemm_identity_asymmetric():
    """A = I with an asymmetric B catches any row/col swap in the MFMA fragment or accumulator layout."""
    ops = _ops()
    n = 96
    b = torch.arange(n * n, dtype=torch.float32).reshape(n, n) * 0.5 + torch.arange(n, dtype=torch.float32)[:, None] * 7
    out = ops.gemm(torch.eye(n).cuda(), b.cuda())
    assert torch.equal(out.cpu(), b)


@pytest.mark.parametrize('M,N,K', [(751, 256, 256), (38061, 32, 480), (1000, 3, 256), (777, 1, 256), (513, 64, 15),
                                   (64, 64, 16), (1, 768, 256), (2753, 512, 128), (130, 1024, 3840)])
def test_gemm_vs_fp64(M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g); b = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g); div = torch.randint(1, 40, (M,), generator=g).float(); res = torch.randn(M, N, generator=g)
    ref = (a.double() @ b.double())
    out = ops.gemm(a.cuda(), b.cuda()).cpu()
    tol = 2e-6 * K ** 0.5 * 4 + 1e-6
    assert (out.double() - ref).abs().max() < tol * max(1.0, ref.abs().max().item() / 10)
    ref2 = torch.relu(ref / div[:, None].double() + bias.double()) + res.double()
    out2 = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), row_div=div.cuda(), residual=res.cuda(), relu=True).cpu()
    assert (out2.double() - ref2).abs().max() < tol * max(1.0, ref.abs().max().item() / 10)


def test_gemm_strided_views():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    big = torch.randn(300, 768, generator=g).cuda()
    w = torch.randn(256, 512, generator=g).cuda()
    out = torch.zeros(300, 768).cuda()
    ops.gemm(big[:, 256:512], w, out=out[:, :512])
    ref = big[:, 256:512].cpu().double() @ w.cpu().double()
    assert (out[:, :512].cpu().double() - ref).abs().max() < 2e-4
    assert out[:, 512:].abs().max() == 0


# ------------------------------------------------------------------------------------------------ bf16x3 split GEMM
@pytest.fixture
def x3_forced():
    ops = _ops()
    ops.force_x3_gemm = True
    yield ops
    ops.force_x3_gemm = False


def test_gemm_x3_identity_asymmetric(x3_forced):
    """A = I with an asymmetric B through the split kernel: exact (every float32 here is an exact 3 x bf16 sum)."""
    ops = x3_forced
    n = 128
    b = torch.arange(n * n, dtype=torch.float32).reshape(n, n) * 0.5 + torch.arange(n, dtype=torch.float32)[:, None] * 7
    for layout, w in (('kn', b), ('nk', b.t().contiguous())):
        sw = ops.SplitWeight(w.cuda(), layout)
        assert sw.planes is not None
        out = ops.gemm(torch.eye(n).cuda(), sw)
        assert torch.equal(out.cpu(), b), layout


@pytest.mark.parametrize('M,N,K,layout', [(751, 256, 256, 'nk'), (2753, 512, 128, 'nk'), (130, 1024, 3840, 'nk'),
                                          (9381, 256, 3840, 'kn'), (40000, 64, 960, 'kn'), (1000, 128, 32, 'nk'),
                                          (513, 64, 16, 'nk'), (300, 768, 256, 'nk'), (70000, 128, 64, 'nk'), (1, 256, 1024, 'nk'),
                                          (70001, 32, 480, 'kn'), (300, 32, 64, 'nk')])
def test_gemm_x3_vs_fp64(M, N, K, layout, x3_forced):
    """The bf16x3 split kernel is float32-grade: its error against float64 is within the bound used for the exact-f32
    MFMA kernel and within 2x of that kernel's own error."""
    ops = x3_forced
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g)); b = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g); div = torch.randint(1, 40, (M,), generator=g).float(); res = torch.randn(M, N, generator=g)
    ref = (a.double() @ b.double())
    sw = ops.SplitWeight((b if layout == 'kn' else b.t().contiguous()).cuda(), layout)
    assert sw.planes is not None
    out = ops.gemm(a.cuda(), sw).cpu()
    out_f32 = ops.gemm(a.cuda(), b.cuda()).cpu()
    tol = 2e-6 * K ** 0.5 * 4 + 1e-6
    scale = max(1.0, ref.abs().max().item() / 10)
    err, err_f32 = (out.double() - ref).abs().max().item(), (out_f32.double() - ref).abs().max().item()
    assert err < tol * scale, (err, err_f32)
    assert err < 2 * err_f32 + 1e-7 * scale, (err, err_f32)
    ref2 = torch.relu(ref / div[:, None].double() + bias.double()) + res.double()
    out2 = ops.gemm(a.cuda(), sw, bias=bias.cuda(), row_div=div.cuda(), residual=res.cuda(), relu=True).cpu()
    assert (out2.double() - ref2).abs().max() < tol * scale


def test_gemm_x3_strided_and_fused_norm(x3_forced):
    ops = x3_forced
    g = torch.Generator().manual_seed(11)
    big = torch.randn(300, 768, generator=g).cuda()
    w = torch.randn(512, 256, generator=g).cuda()                     # (N, K) like nn.Linear
    out = torch.zeros(300, 768).cuda()
    ops.gemm(big[:, 256:512], ops.SplitWeight(w, 'nk'), out=out[:, :512])
    ref = big[:, 256:512].cpu().double() @ w.cpu().double().t()
    assert (out[:, :512].cpu().double() - ref).abs().max() < 2e-4
    assert out[:, 512:].abs().max() == 0
    # InstanceNorm + LeakyReLU folded into the A load: same values as the exact-f32 kernel's fused path
    lens = np.array([700, 1300], np.int32)
    y = (torch.randn(2000, 64, generator=g) * 3 + 1).cuda()
    st = ops.instnorm_stats(y, seg_of(lens), int(lens.max()))
    w2 = torch.randn(64, 128, generator=g).cuda() / 8
    o_x3 = ops.gemm(y, ops.SplitWeight(w2, 'kn'), a_stats=st, a_seg_off=seg_of(lens))
    o_f32 = ops.gemm(y, w2, a_stats=st, a_seg_off=seg_of(lens))
    assert (o_x3 - o_f32).abs().max() < 2e-5 * max(1.0, o_f32.abs().max().item())


@pytest.mark.parametrize('lens,N,K', [([700, 1300], 128, 64), ([33, 1, 64, 7, 700, 0, 300, 2, 2, 2, 61], 64, 32),
                                      ([5000, 30000, 1], 256, 960), ([64, 64, 128], 64, 128), ([40000, 1, 30001, 77], 32, 480)])
def test_gemm_x3_epilogue_instnorm_stats(lens, N, K, x3_forced):
    """InstanceNorm statistics emitted by the GEMM epilogue (per row-tile / cloud partial sums, tiles straddling tiny
    clouds included) == the stand-alone statistics pass over the result."""
    ops = x3_forced
    g = torch.Generator().manual_seed(sum(lens) + N)
    M = sum(lens)
    a = (torch.randn(M, K, generator=g) * 2 + 0.5).cuda()
    w = torch.randn(K, N, generator=g).cuda() / K ** 0.5
    div = torch.randint(1, 40, (M,), generator=g).float().cuda()
    seg = seg_of(np.array(lens, np.int32))
    out, st = ops.gemm(a, ops.SplitWeight(w, 'kn'), row_div=div, want_stats=(seg, max(lens)))
    ref = ops.instnorm_stats(out, seg, max(lens))
    assert torch.equal(out, ops.gemm(a, ops.SplitWeight(w, 'kn'), row_div=div))
    assert (st[..., 0] - ref[..., 0]).abs().max() <= 1e-6 * max(1.0, ref[..., 0].abs().max().item())
    assert ((st[..., 1] - ref[..., 1]).abs() <= 2e-6 * ref[..., 1].abs() + 1e-30).all()


# ------------------------------------------------------------------------------------------------ preprocessing
def _check_subsample(pts, lens, dl):
    from oracle import native
    ops = _ops()
    n = len(pts)
    out, out_seg = ops.grid_subsample(to_dev(pts), seg_of(lens), n, dl)
    oseg = out_seg.cpu().numpy()
    ref_p, ref_l = native.grid_subsample(pts, np.asarray(lens, np.int32), dl)
    assert np.array_equal(np.diff(oseg), ref_l)
    got = out[:oseg[-1]].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref_p.view(np.uint32)), 'barycentres / order not bit exact'
    return got, np.diff(oseg).astype(np.int32)


def _check_radius(q, ql, s, sl, r, K):
    from oracle import native
    ops = _ops()
    grid = ops.CellGrid(to_dev(s), seg_of(sl), len(s), r)
    idx, cnt, mx = grid.query(to_dev(q), seg_of(ql), len(q), K, want_count=True)
    ref_idx, ref_cnt, _ = native.radius_neighbors(q, s, np.asarray(ql, np.int32), np.asarray(sl, np.int32), r, K)
    assert np.array_equal(cnt[:len(q)].cpu().numpy(), ref_cnt)
    assert int(mx.item()) == (ref_cnt.max() if len(q) else 0)
    assert np.array_equal(idx[:len(q)].cpu().numpy(), ref_idx), 'neighbour indices not bit exact'
    if q is s and len(q):
        # queries == supports (the conv tables): the cell-centric self-query kernel must give the very same table
        s_dev, s_seg = grid.s_xyz, grid.s_seg_off
        ops.SELF_QUERY_MIN_POINTS, keep = 0, ops.SELF_QUERY_MIN_POINTS
        try:
            idx2, cnt2, mx2 = grid.query(s_dev, s_seg, len(s), K, want_count=True)
        finally:
            ops.SELF_QUERY_MIN_POINTS = keep
        assert np.array_equal(idx2[:len(q)].cpu().numpy(), ref_idx), 'self-query kernel: neighbour indices not bit exact'
        assert np.array_equal(cnt2[:len(q)].cpu().numpy(), ref_cnt) and int(mx2.item()) == ref_cnt.max()
    return idx[:len(q)]


@pytest.mark.parametrize('case', ['modelnet', '3dmatch_crop'])
def test_preprocess_native_fixture(case):
    g = gold(f'native_{case}')
    pts, lens, dl, r = g['pts'], g['lens'], float(g['dl']), float(g['radius'])
    sub, sub_l = _check_subsample(pts, lens, dl)
    _check_radius(pts, lens, pts, lens, r, 40)
    _check_radius(sub, sub_l, pts, lens, r, 40)
    _check_radius(pts, lens, pts, lens, r, 7)        # heavy truncation
    # product barycentres == the unmodified reference's, as multisets (row order is the documented difference)
    o = 0
    for n in sub_l:
        a, b = sub[o:o + n], g['sub_pts'][o:o + n]
        assert np.array_equal(a[np.lexsort(a.T[::-1])].view(np.uint32), b[np.lexsort(b.T[::-1])].view(np.uint32))
        o += n


@pytest.mark.parametrize('case', ['3dmatch_kitchen', '3dmatch_home_at', '3dmatch_hotel'])
def test_preprocess_full_pyramid_kitchen(case):
    """3DMatch-sized real pairs (the three the reference ships; home_at: 22.7 % of level-0 balls overflow K = 40): every level's points
    and both neighbour tables bit exact vs the oracle."""
    g = gold(case)
    pts = np.concatenate([g['src'], g['tgt']]); lens = np.array([len(g['src']), len(g['tgt'])], np.int32)
    dl, r = 0.05, 0.0625
    for level in range(4):
        _check_radius(pts, lens, pts, lens, r, 40)
        if level == 3:
            break
        sub, sub_l = _check_subsample(pts, lens, dl)
        assert np.array_equal(sub_l, g[f'lens_{level + 1}'])       # same voxel counts as the reference
        _check_radius(sub, sub_l, pts, lens, r, 40)
        pts, lens, dl, r = sub, sub_l, dl * 2, r * 2


def test_preprocess_edge_cases():
    rng = np.random.default_rng(11)
    # ragged batch incl. an EMPTY cloud, a single-point cloud, exact-tie lattice data, negative coordinates
    clouds = [synth_cloud(rng, 700, lattice=0.01) - 3.0, np.zeros((0, 3), np.float32), synth_cloud(rng, 1, 0.1),
              synth_cloud(rng, 1300, lattice=0.006) + 5.0]
    pts = np.concatenate(clouds); lens = np.array([len(c) for c in clouds], np.int32)
    sub, sub_l = _check_subsample(pts, lens, 0.05)
    _check_radius(pts, lens, pts, lens, 0.0625, 40)
    _check_radius(sub, sub_l, pts, lens, 0.0625, 40)
    # all points in one voxel / one ball: long member lists, > 256 candidates (LDS list shrink path), K > count
    dense = (rng.uniform(0, 0.03, (900, 3))).astype(np.float32)
    _check_subsample(dense, [900], 0.05)
    _check_radius(dense, [900], dense, [900], 0.06, 40)
    _check_radius(dense, [900], dense, [900], 0.06, 300)
    _check_radius(dense[:5], [5], dense, [900], 1e-4, 16)     # mostly empty rows -> all padding
    # duplicates: identical points tie at d2 = 0 -> index order
    dup = np.repeat(synth_cloud(rng, 50), 3, axis=0)
    _check_radius(dup, [150], dup, [150], 0.1, 40)


def test_preprocess_large_random_sizes():
    rng = np.random.default_rng(5)
    clouds = [synth_cloud(rng, n, extent=3.0) for n in (20000, 23000)]
    pts = np.concatenate(clouds); lens = np.array([len(c) for c in clouds], np.int32)
    # size-independent properties at full size (the oracle's brute force is checked on a slice)
    ops = _ops()
    out, out_seg = ops.grid_subsample(to_dev(pts), seg_of(lens), len(pts), 0.05)
    oseg = out_seg.cpu().numpy()
    sub = out[:oseg[-1]].cpu().numpy()
    # idempotence-like: every barycentre lies in its own voxel -> subsampling the result at the same dl keeps counts
    grid = ops.CellGrid(to_dev(pts), seg_of(lens), len(pts), 0.0625)
    idx = grid.query(to_dev(pts), seg_of(lens), len(pts), 40).cpu().numpy()
    assert np.array_equal(idx[:, 0], np.arange(len(pts)))               # self is the nearest neighbour
    valid = idx < len(pts)
    nb = np.where(valid, idx, 0)
    d = np.linalg.norm(pts[nb] - pts[:, None], axis=2)
    assert (d[valid] < 0.0625 + 1e-6).all()
    dd = np.where(valid, d, np.inf)
    assert (np.diff(dd, axis=1)[valid[:, 1:]] >= -1e-7).all()           # sorted by distance
    cl = (idx[:, 0] >= lens[0]).astype(int)
    assert ((np.where(valid, idx, -1) >= lens[0]) == (cl[:, None] == 1))[valid].all()   # never crosses clouds
    sl = slice(0, 3000)
    from oracle import native
    ref, _, _ = native.radius_neighbors(pts[sl], pts, np.array([3000, 0], np.int32), lens, 0.0625, 40)
    assert np.array_equal(idx[sl], ref)
    assert len(sub) == native.grid_subsample(pts, lens, 0.05)[0].shape[0]


def test_cpp_wrappers_dropin():
    """numpy-level drop-in of cpp_subsampling.subsample_batch / cpp_neighbors.batch_query (kpconv.py:176,254)."""
    from oracle import native
    from regtr_amd.cpp_wrappers import grid_subsampling, radius_neighbors
    g = gold('native_modelnet')
    pts, lens = g['pts'], g['lens']
    s_pts, s_len = grid_subsampling.subsample_batch(pts, lens, sampleDl=float(g['dl']), max_p=0, verbose=0)
    assert s_pts.dtype == np.float32 and s_len.dtype == np.int32 and np.array_equal(s_len, g['sub_lens'])
    nb = radius_neighbors.batch_query(pts, pts, lens, lens, radius=float(g['radius']))
    assert nb.dtype == np.int32 and nb.shape == g['neighbors'].shape      # same untruncated width as the reference
    ref, _, _ = native.radius_neighbors(pts, pts, lens, lens, float(g['radius']), nb.shape[1])
    assert np.array_equal(nb, ref)
    with pytest.raises(RuntimeError):
        radius_neighbors.batch_query(pts[:, :2], pts, lens, lens, radius=0.1)


def test_cpp_wrappers_more_than_448_supports_in_a_ball():
    """batch_query has no row-width limit in the reference (neighbors.cpp:290-293); above the wavefront kernel's 448 the drop-in
    answers through the KD-tree kernel and re-sorts to the canonical order."""
    from oracle import native
    from regtr_amd.cpp_wrappers import radius_neighbors
    rng = np.random.default_rng(4)
    dense = rng.uniform(0, 0.05, (700, 3)).astype(np.float32)
    far = (rng.uniform(0, 1, (300, 3)) + 3).astype(np.float32)
    pts = np.concatenate([dense, far]); lens = np.array([1000], np.int32)
    nb = radius_neighbors.batch_query(pts, pts, lens, lens, radius=0.2)
    assert nb.shape[1] > 448
    ref, _, _ = native.radius_neighbors(pts, pts, lens, lens, 0.2, nb.shape[1])
    assert np.array_equal(nb, ref)


@pytest.mark.parametrize('case', ['modelnet', '3dmatch_crop'])
def test_cpp_wrappers_reference_order(case):
    """cpp_wrappers.reference_order(True): both drop-in ops return what the unmodified reference C++ returned (committed
    fixtures made by oracle/make_golden.py) ELEMENT FOR ELEMENT -- subsampled rows in libstdc++ unordered_map iteration order,
    neighbour rows in nanoflann visiting order + std::sort tie order, untruncated width."""
    from regtr_amd import cpp_wrappers
    g = gold(f'native_{case}')
    pts, lens = g['pts'], g['lens']
    with cpp_wrappers.reference_order():
        s_pts, s_len = cpp_wrappers.grid_subsampling.subsample_batch(pts, lens, sampleDl=float(g['dl']), max_p=0, verbose=0)
        assert np.array_equal(s_len, g['sub_lens']) and np.array_equal(s_pts.view(np.uint32), g['sub_pts'].view(np.uint32))
        nb = cpp_wrappers.radius_neighbors.batch_query(pts, pts, lens, lens, radius=float(g['radius']))
        assert np.array_equal(nb, g['neighbors'])
        pool = cpp_wrappers.radius_neighbors.batch_query(s_pts, pts, s_len, lens, radius=float(g['radius']))
        assert np.array_equal(pool, g['pools'])


def test_radius_first_k_by_index_vs_restatement():
    """order = 1 (the reference PreprocessorGPU's neighbour sets: first K supports of the ball by index) against the restatement of
    pytorch3d ball_query, on clouds dense enough that most balls hold more than K supports; per-query and cell-centric kernels."""
    from oracle import regtr_ref
    ops = _ops()
    rng = np.random.default_rng(5)
    clouds = [synth_cloud(rng, 3000, lattice=0.01), synth_cloud(rng, 2500) + 5.0]
    s = np.concatenate(clouds).astype(np.float32); lens = np.array([3000, 2500], np.int32)
    q = np.concatenate([c[::3] for c in clouds]).astype(np.float32); ql = np.array([1000, 834], np.int32)
    r, K = 0.2, 16
    seg, qseg = seg_of(lens), seg_of(ql)
    sd, qd = to_dev(s), to_dev(q)
    grid = ops.CellGrid(sd, seg, len(s), r)
    ref_pool = regtr_ref.ball_query_first_k(q, s, ql, lens, r, K)
    ref_self = regtr_ref.ball_query_first_k(s, s, lens, lens, r, K)
    assert ((ref_self < len(s)).sum(1) == K).mean() > 0.5                      # truncation really happens
    got_pool = grid.query(qd, qseg, len(q), K, order=1).cpu().numpy()
    assert np.array_equal(got_pool, ref_pool)
    prev = ops.SELF_QUERY_MIN_POINTS
    try:
        for min_pts in (1, 1 << 30):                                           # cell-centric kernel, then the per-query kernel
            ops.SELF_QUERY_MIN_POINTS = min_pts
            assert np.array_equal(grid.query(sd, seg, len(s), K, order=1).cpu().numpy(), ref_self)
    finally:
        ops.SELF_QUERY_MIN_POINTS = prev
    near = grid.query(qd, qseg, len(q), K, order=0).cpu().numpy()              # the default keeps the K nearest: other sets
    assert (np.sort(near, 1) != np.sort(got_pool, 1)).any()


@pytest.mark.parametrize('mode', [1, 2])
def test_grid_subsample_floor_keys_vs_restatement(mode):
    """key_mode 1 / 2 (the reference PreprocessorGPU's voxel rule floor(p / dl), kpconv.py:213-240) against the restatement: lattice
    clouds with points exactly on voxel faces, negative coordinates, an empty and a one-point cloud; bit-exact barycentres in
    first-appearance order, and a voxel set that differs from the CPU rule's (key_mode 0)."""
    from oracle import native
    ops = _ops()
    rng = np.random.default_rng(11)
    clouds = [synth_cloud(rng, 6000, lattice=0.00625) - 1.3, np.zeros((0, 3), np.float32), synth_cloud(rng, 1, lattice=0.05),
              (rng.integers(-40, 40, (5000, 3)) * 0.0125).astype(np.float32)]
    pts = np.concatenate(clouds).astype(np.float32); lens = np.array([len(c) for c in clouds], np.int32)
    dl = 0.05
    out, out_seg = ops.grid_subsample(to_dev(pts), seg_of(lens), len(pts), dl, key_mode=mode)
    oseg = out_seg.cpu().numpy()
    ref_p, ref_l = native.grid_subsample(pts, lens, dl, key_mode=mode)
    assert np.array_equal(np.diff(oseg), ref_l)
    assert np.array_equal(out[:oseg[-1]].cpu().numpy(), ref_p)
    cpu_rule = native.grid_subsample(pts, lens, dl, key_mode=0)[1]
    assert not np.array_equal(cpu_rule, ref_l)                              # face points fall the other way without the origin shift
    # membership: every input point's floor(p / dl) cell holds exactly one output barycentre of its cloud
    k = np.floor(pts / np.float32(dl) if mode == 1 else pts * (np.float32(1) / np.float32(dl))).astype(np.int64)
    cid = np.repeat(np.arange(len(lens)), lens)
    assert len(np.unique(np.concatenate([cid[:, None], k], 1), axis=0)) == int(ref_l.sum())


# ------------------------------------------------------------------------------------------------ encoder kernels
@pytest.mark.parametrize('Cin,Cout,H', [(1, 64, 40), (32, 32, 40), (64, 64, 40), (128, 128, 50), (256, 256, 40),
                                        (16, 64, 40), (48, 32, 40), (20, 12, 33)])      # last three: general LDS-tile gather + flag pass
def test_kpconv_vs_oracle(Cin, Cout, H):
    from oracle import native, regtr_ref
    from regtr_amd.kernel_points import K015_CENTER
    ops = _ops()
    rng = np.random.default_rng(Cin)
    s = synth_cloud(rng, 1500); q = s[::3].copy()
    r = 0.12
    idx, _, _ = native.radius_neighbors(q, s, np.array([len(q)], np.int32), np.array([len(s)], np.int32), r, H)
    x = rng.standard_normal((len(s), Cin)).astype(np.float32)
    x[rng.random(len(s)) < 0.2] *= -1.0            # some supports with negative feature sums (normaliser)
    if Cin == 1:
        x[:] = 1.0
    w = (rng.standard_normal((15, Cin, Cout)) / math.sqrt(Cin * 15)).astype(np.float32)
    kp = (K015_CENTER * r).astype(np.float32)
    extent = r * 2.0 / 2.5
    ref = regtr_ref.kpconv(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(idx.astype(np.int64)),
                           torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(kp), extent)
    out = ops.kpconv(to_dev(q), to_dev(s), to_dev(idx), to_dev(x), to_dev(w.reshape(15 * Cin, Cout)), to_dev(kp), extent)
    assert (out.cpu() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
    if Cin == 1:      # (x, y, z, feature) records, as the encoder's first block passes them
        x[:] = rng.standard_normal((len(s), 1)).astype(np.float32)
        ref = regtr_ref.kpconv(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(idx.astype(np.int64)),
                               torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(kp), extent)
        out = ops.kpconv(to_dev(q), to_dev(s), to_dev(idx), to_dev(x), to_dev(w.reshape(15 * Cin, Cout)), to_dev(kp), extent,
                         xyzf=to_dev(np.concatenate([s, x], 1)))
        assert (out.cpu() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())


def test_fused_instnorm_paths_vs_oracle():
    """InstanceNorm+LeakyReLU folded into the KPConv gather (unary1 tail) and into the GEMM A load (conv tail)."""
    from oracle import native, regtr_ref
    from regtr_amd.kernel_points import K015_CENTER
    ops = _ops()
    rng = np.random.default_rng(9)
    clouds = [synth_cloud(rng, 900), synth_cloud(rng, 1400) + 7.0]
    s = np.concatenate(clouds); lens = np.array([900, 1400], np.int32)
    q, ql = native.grid_subsample(s, lens, 0.08)
    r, Cin, Cout = 0.15, 64, 48
    idx, _, _ = native.radius_neighbors(q, s, ql, lens, r, 40)
    y = (rng.standard_normal((len(s), Cin)) * rng.uniform(0.2, 3, Cin) + rng.uniform(-2, 2, Cin)).astype(np.float32)
    w = (rng.standard_normal((15, Cin, Cout)) / 30).astype(np.float32)
    kp = (K015_CENTER * r).astype(np.float32)
    L = torch.from_numpy(lens.astype(np.int64))
    x_ref = torch.nn.functional.leaky_relu(regtr_ref.instance_norm(torch.from_numpy(y), L), 0.1)
    ref = regtr_ref.kpconv(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(idx.astype(np.int64)), x_ref,
                           torch.from_numpy(w), torch.from_numpy(kp), 0.12)
    st = ops.instnorm_stats(to_dev(y), seg_of(lens), int(lens.max()))
    out = ops.kpconv(to_dev(q), to_dev(s), to_dev(idx), to_dev(y), to_dev(w.reshape(15 * Cin, Cout)), to_dev(kp), 0.12,
                     x_stats=st, s_seg_off=seg_of(lens), q_seg_off=seg_of(ql))
    assert (out.cpu() - ref).abs().max() < 3e-5 * max(1.0, ref.abs().max().item())
    # the form the encoder uses: IN + LReLU applied in place with the row flags as a by-product, gather without fold or row sums
    for C2 in (32, 64, 128, 256):
        y2 = (rng.standard_normal((len(s), C2)) * rng.uniform(0.2, 3, C2) + rng.uniform(-2, 2, C2)).astype(np.float32)
        w3 = (rng.standard_normal((15, C2, 64)) / 30).astype(np.float32)
        x2_ref = torch.nn.functional.leaky_relu(regtr_ref.instance_norm(torch.from_numpy(y2), L), 0.1)
        ref3 = regtr_ref.kpconv(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(idx.astype(np.int64)), x2_ref,
                                torch.from_numpy(w3), torch.from_numpy(kp), 0.12)
        yd = to_dev(y2)
        st3 = ops.instnorm_stats(yd, seg_of(lens), int(lens.max()))
        flag = torch.full((len(s),), -1.0, device='cuda')
        xyzf = torch.full((len(s), 4), -1.0, device='cuda')
        ops.instnorm_apply(yd.clone(), seg_of(lens), int(lens.max()), st3, lrelu=True, row_positive=flag)
        ops.instnorm_apply(yd, seg_of(lens), int(lens.max()), st3, lrelu=True, out=yd, row_xyz=to_dev(s), row_positive=xyzf)
        assert torch.equal(xyzf[:, :3].cpu(), torch.from_numpy(s)) and torch.equal(xyzf[:, 3], flag)
        assert (yd.cpu() - x2_ref).abs().max() < 2e-5
        rs = x2_ref.double().sum(1)
        sure = rs.abs() > 1e-4                                   # rows whose sign no summation order can flip
        assert torch.equal(flag.cpu()[sure], (rs[sure] > 0).float()) and set(flag.cpu().unique().tolist()) <= {0.0, 1.0}
        out3 = ops.kpconv(to_dev(q), to_dev(s), to_dev(idx), yd, to_dev(w3.reshape(15 * C2, 64)), to_dev(kp), 0.12, xyzf=xyzf)
        assert (out3.cpu() - ref3).abs().max() < 3e-5 * max(1.0, ref3.abs().max().item())
    # GEMM with the normalisation folded into the A operand
    w2 = (rng.standard_normal((Cin, 96)) / 8).astype(np.float32)
    ref2 = x_ref @ torch.from_numpy(w2)
    out2 = ops.gemm(to_dev(y), to_dev(w2), a_stats=st, a_seg_off=seg_of(lens))
    assert (out2.cpu() - ref2).abs().max() < 3e-5 * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize('nq_take', [1, 3])
def test_kpconv_fused_vs_two_kernel_path(nq_take):
    """regtr_kpconv_fused (level-0 shape: 32 -> 32 channels, weighted features kept in LDS) against gather + contraction, same inputs:
    float32 rounding only.  Query counts that are not multiples of the workgroup's 256 (surplus waves run zero tiles).
    The kernel is NOT in the product library (measured slower, docs/NEGATIVES.md): libregtr_hip.experimental.so, regtr_amd/experimental.py."""
    from oracle import native
    from regtr_amd import experimental
    from regtr_amd.kernel_points import K015_CENTER
    if not experimental.available():
        pytest.skip('libregtr_hip.experimental.so not built (python -m regtr_amd.build --experimental)')
    ops = _ops()
    rng = np.random.default_rng(21)
    clouds = [synth_cloud(rng, 3000), synth_cloud(rng, 2101) + 6.0]
    s = np.concatenate(clouds).astype(np.float32); lens = np.array([3000, 2101], np.int32)
    q = s[::nq_take].copy(); ql = np.array([len(range(0, 3000, nq_take)), len(s[::nq_take]) - len(range(0, 3000, nq_take))], np.int32)
    if nq_take > 1:      # strided-style: queries are a subset; keep clouds separate
        q = np.concatenate([clouds[0][::nq_take], clouds[1][::nq_take]]).astype(np.float32)
        ql = np.array([len(clouds[0][::nq_take]), len(clouds[1][::nq_take])], np.int32)
    r = 0.12
    idx, _, _ = native.radius_neighbors(q, s, ql, lens, r, 40)
    x = rng.standard_normal((len(s), 32)).astype(np.float32)
    x[rng.random(len(s)) < 0.3] *= -1.0
    w = (rng.standard_normal((15 * 32, 32)) / 20).astype(np.float32)
    kp = (K015_CENTER * r).astype(np.float32)
    xd = to_dev(x)
    xyzf = torch.cat((to_dev(s), (xd.sum(1, keepdim=True) > 0).float()), 1).contiguous()
    sw = ops.SplitWeight(to_dev(w), 'kn')
    assert sw.planes is not None
    args = (to_dev(q), to_dev(s), to_dev(idx), xd, sw, to_dev(kp), r * 0.8)
    fused = experimental.kpconv_fused(args[0], args[2], xd, xyzf, sw, args[5], r * 0.8)       # the experiment library
    plain = ops.kpconv(*args, xyzf=xyzf)                                                         # the product path
    assert (fused - plain).abs().max().item() < 3e-6 * max(1.0, plain.abs().max().item())


def test_maxpool_instnorm_vs_oracle():
    from oracle import native, regtr_ref
    ops = _ops()
    rng = np.random.default_rng(2)
    s = synth_cloud(rng, 2000); q = s[::4].copy()
    idx, _, _ = native.radius_neighbors(q, s, np.array([len(q)], np.int32), np.array([len(s)], np.int32), 0.1, 40)
    x = rng.standard_normal((len(s), 128)).astype(np.float32)
    ref = regtr_ref.max_pool(torch.from_numpy(x), torch.from_numpy(idx.astype(np.int64)))
    assert torch.equal(ops.maxpool(to_dev(x), to_dev(idx)).cpu(), ref)
    for C, lens in [(64, [700, 0, 1300]), (1024, [301, 450]), (32, [20000, 9]), (256, [1, 2])]:
        n = sum(lens)
        y = (rng.standard_normal((n, C)) * rng.uniform(0.1, 5, C) + rng.uniform(-3, 3, C)).astype(np.float32)
        res = rng.standard_normal((n, C)).astype(np.float32)
        seg = seg_of(lens)
        st = ops.instnorm_stats(to_dev(y), seg, max(lens))
        out = ops.instnorm_apply(to_dev(y), seg, max(lens), st, lrelu=True).cpu()
        L = torch.tensor(lens)
        ref = torch.nn.functional.leaky_relu(regtr_ref.instance_norm(torch.from_numpy(y), L), 0.1)
        assert (out - ref).abs().max() < 2e-5
        st2 = ops.instnorm_stats(to_dev(res), seg, max(lens))
        out2 = ops.instnorm_apply(to_dev(y), seg, max(lens), st, residual=to_dev(res), res_stats=st2, lrelu=True).cpu()
        ref2 = torch.nn.functional.leaky_relu(regtr_ref.instance_norm(torch.from_numpy(y), L) +
                                              regtr_ref.instance_norm(torch.from_numpy(res), L), 0.1)
        assert (out2 - ref2).abs().max() < 4e-5


def test_maxpool_rows_beyond_2_gib():
    """Level 0 of a 192-pair forward (bench.py's default since round 5) hands regtr_maxpool_gather 3.7 GB of feature rows: the branch-free kernel's
    buffer offsets are 32-bit UNSIGNED and its out-of-range offset sits above 4 GiB - 256, so rows between 2 and 4 GiB must be gathered like
    any other (max over the listed rows, a zero row for the shadow index ns; kpconv_blocks.py:127-143) -- checked on 4096 queries whose
    neighbours are spread over the whole table, the shadow index included."""
    ops = _ops()
    ns, C, H, nq = 6_000_000, 128, 40, 4096          # 3.07 GB of rows
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn((ns, C), device='cuda', generator=g)
    idx = torch.randint(0, ns + 1, (nq, H), device='cuda', generator=g, dtype=torch.int64)
    idx[:, -3:] = ns                                   # shadow neighbours
    idx[0, :] = ns - 1                                 # the last row of the table: offset 3.07 GB
    idx[1, :] = ns                                     # only shadows: zeros
    out = ops.maxpool(x, idx.to(torch.int32))
    xp = torch.cat([x, torch.zeros((1, C), device='cuda')])
    ref = torch.stack([xp[idx[i0:i0 + 256]].amax(1) for i0 in range(0, nq, 256)]).reshape(nq, C)
    assert torch.equal(out, ref)
    assert float(out[1].abs().max()) == 0.0 and torch.equal(out[0], x[ns - 1])


# ------------------------------------------------------------------------------------------------ transformer kernels
def test_layernorm_posemb_vs_oracle():
    from oracle import regtr_ref
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(751, 256, generator=g) * 3 + 1
    gm, bt, add = torch.randn(256, generator=g), torch.randn(256, generator=g), torch.randn(751, 256, generator=g)
    ref = torch.nn.functional.layer_norm(x, (256,), gm, bt)
    out, plain = ops.layernorm(x.cuda(), gm.cuda(), bt.cuda(), add=add.cuda(), want_plain=True)
    assert (plain.cpu() - ref).abs().max() < 1e-5 and (out.cpu() - (ref + add)).abs().max() < 1e-5
    xyz = (torch.rand(613, 3, generator=g) - 0.5) * 8
    pe = ops.posemb_sine(xyz.cuda(), 256, 1.0).cpu()
    assert (pe - regtr_ref.pos_embed_sine(xyz, 256, 1.0)).abs().max() < 2e-5


@pytest.mark.parametrize('precision,tol', [(0, 2e-5), (2, 2e-5), (1, 6e-2), (3, 2e-5)])
@pytest.mark.parametrize('lens', [[412, 339], [601, 612], [33, 1, 64, 7], [2100, 1900], [130, 0, 5, 129],
                                  [150, 97, 260, 33] * 10 + [170, 0, 129, 64] * 10])      # the last one: enough workgroups for the 4-wave kernel
def test_mha_vs_oracle(lens, precision, tol):
    """self- and cross-attention cores on packed ragged clouds vs the plain softmax(QK^T)V restatement in float64:
    precision 0 (bf16x3 split MFMA), 3 (f16 pair split, compute_dtype 'fp32') and 2 (exact-f32 MFMA) at float32 accuracy, 1 (plain bf16
    operands) at bf16 accuracy; an empty partner cloud gives zeros."""
    ops = _ops()
    g = torch.Generator().manual_seed(sum(lens))
    N, E, H = sum(lens), 256, 8
    qkv = torch.randn(N, 3 * E, generator=g) * 1.5
    seg = np.concatenate([[0], np.cumsum(lens)])
    B = len(lens) // 2
    worst = 0.0
    for kv in (list(range(2 * B)), list(range(B, 2 * B)) + list(range(B))):
        out = ops.mha(qkv.cuda()[:, :E], qkv.cuda()[:, E:2 * E], qkv.cuda()[:, 2 * E:], seg_of(lens),
                      torch.tensor(kv, dtype=torch.int32).cuda(), max(lens), H, precision).cpu()
        for c in range(2 * B):
            if seg[c + 1] == seg[c]:
                continue
            if seg[kv[c] + 1] == seg[kv[c]]:
                assert out[seg[c]:seg[c + 1]].abs().max() == 0
                continue
            q = qkv[seg[c]:seg[c + 1], :E].view(-1, H, 32).transpose(0, 1) / math.sqrt(32)
            k = qkv[seg[kv[c]]:seg[kv[c] + 1], E:2 * E].view(-1, H, 32).transpose(0, 1)
            v = qkv[seg[kv[c]]:seg[kv[c] + 1], 2 * E:].view(-1, H, 32).transpose(0, 1)
            ref = (torch.softmax(q.double() @ k.double().transpose(1, 2), -1) @ v.double()).transpose(0, 1).reshape(-1, E)
            worst = max(worst, (out[seg[c]:seg[c + 1]].double() - ref).abs().max().item())
    print(f'mha precision {precision} lens {lens}: max abs err {worst:.2e}')
    assert worst < tol


STREAM_LENS = [[5000, 30000, 1, 0, 7, 33, 2100, 64, 64, 9000], [40000, 25000], [31, 1, 0, 2, 70000, 3], [100, 156, 3, 250]]


@pytest.mark.parametrize('K,N', [(32, 128), (64, 128), (64, 256), (128, 32), (128, 64), (64, 32), (128, 128), (32, 32)])
@pytest.mark.parametrize('lens', STREAM_LENS)
def test_gemm_stream_vs_exact_f32(lens, K, N):
    """regtr_gemm_stream (32-row strips from global memory straight into MFMA fragments, weights in LDS per 256-row workgroup,
    per-tile float64 statistics): product vs the exact-f32 kernel, InstanceNorm statistics vs the stand-alone pass over the
    result -- ragged clouds with empty / one-row clouds inside a tile, tile boundaries inside clouds; with and without the producer's
    InstanceNorm+LeakyReLU folded into the A load (K <= 64)."""
    ops = _ops()
    g = torch.Generator().manual_seed(sum(lens) + N + K)
    M = sum(lens)
    seg = seg_of(np.array(lens, np.int32))
    a = (torch.randn(M, K, generator=g) * 2 + 0.5).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    sw = ops.SplitWeight(w, 'nk')
    assert sw.planes is not None
    for fold in ([False, True] if K <= 64 else [False]):
        a_st = ops.instnorm_stats(a, seg, max(lens)) if fold else None
        out, st = ops.gemm_stream(a, sw, seg, a_stats=a_st, want_stats=True)
        ref = ops.gemm(a, w.t().contiguous(), a_stats=a_st, a_seg_off=seg if fold else None)       # exact-f32 MFMA kernel
        assert (out - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), (K, N, fold)
        assert torch.equal(out, ops.gemm_stream(a, sw, seg, a_stats=a_st))                          # without the statistics
        rst = ops.instnorm_stats(out, seg, max(lens))
        assert (st[..., 0] - rst[..., 0]).abs().max() <= 1e-6 * max(1.0, rst[..., 0].abs().max().item())
        assert ((st[..., 1] - rst[..., 1]).abs() <= 2e-6 * rst[..., 1].abs() + 1e-30).all()


@pytest.mark.parametrize('lens', [[70000], [20000, 0, 33001, 12999], [300, 66000, 5], [1000] * 70])
@pytest.mark.parametrize('linear_shortcut', [False, True])
def test_block_tail_res_vs_separate_ops(lens, linear_shortcut):
    """regtr_block_tail_res (level-1 resnet tails: unary2's statistics from the 64 x 64 second moments of its input, the product never
    written, the finished second summand -- identity / max-pooled shortcut, or a Linear shortcut's product with its statistics --
    added in the epilogue) against unary2 GEMM + instnorm_apply with the residual.  Experiment library only (measured slower,
    docs/NEGATIVES.md): the product neither contains nor routes to it."""
    from regtr_amd import experimental
    if not experimental.available():
        pytest.skip('libregtr_hip.experimental.so not built (python -m regtr_amd.build --experimental)')
    ops = _ops()
    rng = np.random.default_rng(len(lens) + 7)
    M, K1, N = sum(lens), 64, 256
    seg = seg_of(lens)
    x1 = (rng.standard_normal((M, K1)) * rng.uniform(0.3, 3, K1) + rng.uniform(-2, 2, K1)).astype(np.float32)
    x1[:, 9] = 0.5 * x1[:, 8] - 1.0                      # correlated channels: the covariance terms matter
    res = (rng.standard_normal((M, N)) * rng.uniform(0.3, 2, N) + rng.uniform(-1, 1, N)).astype(np.float32)
    w1 = (rng.standard_normal((N, K1)) / math.sqrt(K1)).astype(np.float32)
    w1[11] *= 1e-3                                       # a nearly dead output column (eps dominates its rstd)
    x1d, rd = to_dev(x1), to_dev(res)
    sw1 = ops.SplitWeight(to_dev(w1), 'nk')
    x1_st = ops.instnorm_stats(x1d, seg, max(lens))
    r_st = ops.instnorm_stats(rd, seg, max(lens)) if linear_shortcut else None
    y = experimental.block_tail_res(x1d, x1_st, sw1, rd, r_st, seg, max(lens))
    u, u_st = ops.gemm(x1d, sw1, a_stats=x1_st, a_seg_off=seg, want_stats=(seg, max(lens)))
    ref = ops.instnorm_apply(u, seg, max(lens), u_st, residual=rd, res_stats=r_st, lrelu=True)
    assert (y - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    # float64 reference of the whole expression on one cloud
    from oracle import regtr_ref
    L = torch.tensor(lens)
    xn = torch.nn.functional.leaky_relu(regtr_ref.instance_norm(torch.from_numpy(x1).double(), L), 0.1)
    un = regtr_ref.instance_norm(xn @ torch.from_numpy(w1).double().t(), L)
    rn = regtr_ref.instance_norm(torch.from_numpy(res).double(), L) if linear_shortcut else torch.from_numpy(res).double()
    want = torch.nn.functional.leaky_relu(un + rn, 0.1)
    assert (y.cpu().double() - want).abs().max().item() < 5e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize('lens,offset', [([70000], 0.0), ([20000, 0, 33001, 12999], 0.0), ([300, 66000, 5], 0.0), ([1000] * 70, 0.0),
                                         ([40000, 30000], 50.0), ([40000, 30000], -8.0)])
def test_block_tail_vs_separate_ops(lens, offset):
    """regtr_block_tail (statistics from input moments, neither product written) against unary2 GEMM + shortcut GEMM +
    instnorm_apply, and its reported product statistics against float64.  offset: one input channel nearly constant at a large value
    (variance 1e-6 next to mean^2 = 2500) and another one riding on the same offset -- uncentred float32 second moments lose such
    variances to rounding (~1e-6 of sum x^2); the moments are taken about a pivot near the mean (trimmed mean of 16 rows spread over the
    cloud).  offset < 0: every cloud's FIRST row is an outlier 8 standard deviations out in every channel -- an isolated point of a real
    scan; round 4's pivot (the first row itself) lost a factor ~60 of the statistics' accuracy to it (3e-5 on a real 3DMatch cloud)."""
    ops = _ops()
    rng = np.random.default_rng(len(lens))
    M, K1, K2, N = sum(lens), 32, 64, 128
    seg = seg_of(lens)
    x1 = (rng.standard_normal((M, K1)) * rng.uniform(0.3, 3, K1) + rng.uniform(-2, 2, K1)).astype(np.float32)
    f = (rng.standard_normal((M, K2)) * rng.uniform(0.3, 2, K2) + rng.uniform(-1, 1, K2)).astype(np.float32)
    f[:, 5] = 0.25 * f[:, 4] + 3.0                       # correlated and offset channels: the covariance terms matter
    if offset > 0:
        f[:, 6] = offset + 1e-3 * f[:, 6]
        f[:, 7] += offset
    if offset < 0:
        first = np.concatenate([[0], np.cumsum(lens)[:-1]])
        x1[first] = x1.mean(0) + abs(offset) * x1.std(0)
        f[first] = f.mean(0) - abs(offset) * f.std(0)
    w1 = (rng.standard_normal((N, K1)) / math.sqrt(K1)).astype(np.float32)
    w2 = (rng.standard_normal((N, K2)) / math.sqrt(K2)).astype(np.float32)
    w2[7] *= 1e-3                                        # a nearly dead output column (eps dominates its rstd)
    x1d, fd = to_dev(x1), to_dev(f)
    sw1, sw2 = ops.SplitWeight(to_dev(w1), 'nk'), ops.SplitWeight(to_dev(w2), 'nk')
    x1_st = ops.instnorm_stats(x1d, seg, max(lens))
    assert ops.block_tail_ok(x1d, x1_st, fd, sw1, sw2) == (M >= ops.STREAM_MIN_ROWS)      # (the size gate is a speed choice)
    y, st = ops.block_tail(x1d, x1_st, fd, sw1, sw2, seg, max(lens), want_stats=True)
    prev = ops.use_stream_gemm
    try:
        for strip in (True, False):
            ops.use_stream_gemm = strip
            u, u_st = ops.gemm(x1d, sw1, a_stats=x1_st, a_seg_off=seg, want_stats=(seg, max(lens)))
            sc, sc_st = ops.gemm(fd, sw2, want_stats=(seg, max(lens)))
            ref = ops.instnorm_apply(u, seg, max(lens), u_st, residual=sc, res_stats=sc_st, lrelu=True)
            assert (y - ref).abs().max().item() < 3e-5
    finally:
        ops.use_stream_gemm = prev
    # statistics against float64 of the exact products
    L = torch.tensor(lens)
    from oracle import regtr_ref
    xn = torch.nn.functional.leaky_relu(regtr_ref.instance_norm(torch.from_numpy(x1).double(), L), 0.1)
    for k, prod in enumerate((xn @ torch.from_numpy(w1).double().t(), torch.from_numpy(f).double() @ torch.from_numpy(w2).double().t())):
        o = 0
        for c, n in enumerate(lens):
            if n == 0:
                continue
            blk = prod[o:o + n]; o += n
            mean, var = blk.mean(0), blk.var(0, unbiased=False)
            got = st[k, c].cpu().double()
            assert (got[:, 0] - mean).abs().max() < 2e-6 * max(1.0, mean.abs().max().item())
            assert ((got[:, 1] - 1 / torch.sqrt(var + 1e-5)) / (1 / torch.sqrt(var + 1e-5))).abs().max() < 1e-5


def test_first_block_fused_vs_separate_ops():
    """kpconv_norm_lrelu (gather -> 16-float rows -> contraction + InstanceNorm + LeakyReLU from the rows' second moments) against
    kpconv + instnorm_apply, on a two-cloud batch tall enough for the fused path."""
    from regtr_amd.kernel_points import K015_CENTER
    ops = _ops()
    rng = np.random.default_rng(11)
    clouds = [synth_cloud(rng, 40000), synth_cloud(rng, 30000) + 9.0]
    s = np.concatenate(clouds).astype(np.float32); lens = np.array([40000, 30000], np.int32)
    r = 0.06
    x = np.ones((len(s), 1), np.float32)
    w = (rng.standard_normal((15, 1, 64)) / 4).astype(np.float32)
    kp = (K015_CENTER * r).astype(np.float32)
    seg = seg_of(lens)
    sd, xd, kpd = to_dev(s), to_dev(x), to_dev(kp)
    idxd = ops.CellGrid(sd, seg, len(s), r).query(sd, seg, len(s), 40)        # (the search has its own tests)
    assert ops.first_block_ok(len(s), 1, 15, 64) == (len(s) >= ops.STREAM_MIN_ROWS)
    y_ref, st = ops.kpconv(sd, sd, idxd, xd, to_dev(w.reshape(15, 64)), kpd, r * 0.8, want_stats=(seg, int(lens.max())))
    y_ref = ops.instnorm_apply(y_ref, seg, int(lens.max()), st, lrelu=True)
    w16 = to_dev(np.concatenate([w.reshape(15, 64), np.zeros((1, 64), np.float32)]))
    for xyzf in (None, torch.cat((sd, xd), 1).contiguous()):
        y, st2 = ops.kpconv_norm_lrelu(sd, sd, idxd, xd, w16, kpd, r * 0.8, seg, int(lens.max()), xyzf=xyzf, want_stats=True)
        assert (y - y_ref).abs().max().item() < 3e-5
        assert (st2[0] - st).abs().max().item() < 2e-5 * max(1.0, st.abs().max().item())


@pytest.mark.parametrize('planes,tol', [(3, 3e-6), (2, 2e-4), (1, 2e-2)])
def test_gemm_x3_plane_count(planes, tol, x3_forced):
    """regtr_gemm_x3 with 3 (float32-grade), 2 (three-term) and 1 (plain bf16) planes per operand vs float64, relative to the
    result's scale -- the measured accuracy classes behind cfg.compute_dtype."""
    ops = x3_forced
    g = torch.Generator().manual_seed(planes)
    a = torch.randn(3000, 256, generator=g).cuda()
    w = (torch.randn(1024, 256, generator=g) / 16).cuda()
    bias = torch.randn(1024, generator=g).cuda()
    out = ops.gemm(a, ops.SplitWeight(w, 'nk'), bias=bias, relu=True, planes=planes).cpu().double()
    ref = torch.relu(a.cpu().double() @ w.cpu().double().t() + bias.cpu().double())
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print(f'gemm_x3 planes {planes}: max rel err {err:.2e}')
    assert err < tol


@pytest.mark.parametrize('M,N,K,scale', [(150000, 128, 256, 1.0), (70000, 64, 960, 1.0), (66000, 256, 512, 1e-4), (66000, 128, 1920, 20.0)])
def test_gemm_f16_pair_vs_fp64(M, N, K, scale):
    """The f16 pair operand format (regtr_gemm_x3 n_planes = 4: x = h0 + h1 / 2048, three f16 MFMA terms, scaled second accumulator) vs
    float64, next to the six-term bf16 split on the same operands: float32-grade, including operands far below f16's normal range
    (scale 1e-4: most |a| < 6.1e-5, h0 subnormal or zero, the scaled h1 carries them) and large ones (scale 20: up to ~1e4; the
    format's documented limit is f16's 65504), with heavy-tailed magnitudes; bias / ReLU / residual / row_div epilogue and the InstanceNorm statistics of the result included."""
    ops = _ops()
    g = torch.Generator().manual_seed(N + K)
    a = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g)) * scale).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias, res = torch.randn(N, generator=g).cuda() * scale, (torch.randn(M, N, generator=g) * scale).cuda()
    div = (torch.randint(1, 40, (M,), generator=g).float()).cuda()
    sw = ops.SplitWeight(w, 'nk')
    assert ops.f16_pair_ok(M, N, K), 'shape not served by the row-strip kernel'
    lens = [M // 3, M - M // 3 - 7, 7]
    seg = seg_of(lens)
    ref = a.double() @ w.double().t() / div.double()[:, None] + bias.double() + res.double()
    errs = {}
    for mode in (False, True):
        with ops.f16_pair(mode):
            out, st = ops.gemm(a, sw, bias=bias, row_div=div, residual=res, want_stats=(seg, max(lens)))
        errs[mode] = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        if mode:
            o = 0
            for c, n in enumerate(lens):
                blk = ref[o:o + n]; o += n
                assert ((st[c, :, 0].double() - blk.mean(0)).abs().max() / ref.abs().max()).item() < 2e-6
    print(f'gemm M {M} N {N} K {K} scale {scale:g}: max rel err bf16x3 {errs[False]:.2e}, f16 pair {errs[True]:.2e}')
    assert errs[True] < 3e-6 and errs[True] < 8 * max(errs[False], 2e-7)


def test_procrustes_vs_oracle():
    from oracle import regtr_ref
    from regtr_amd.se3 import compute_rigid_transform
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    B, L = 3, 6
    lens = [412, 339, 100, 50, 77, 601]           # [src_0, src_1, src_2, tgt_0, tgt_1, tgt_2]
    N = sum(lens)
    kp = (torch.rand(N, 3, generator=g) - 0.5) * 4
    corr = kp.unsqueeze(0) + torch.randn(L, N, 3, generator=g) * 0.3
    logit = torch.randn(L, N, generator=g) * 2
    pose = ops.weighted_procrustes(kp.cuda(), corr.cuda(), logit.cuda(), seg_of(lens), B).cpu()
    seg = np.concatenate([[0], np.cumsum(lens)])
    for b in range(B):
        s, t = slice(seg[b], seg[b + 1]), slice(seg[B + b], seg[B + b + 1])
        a = torch.cat([kp[s].expand(L, -1, -1), corr[:, t]], 1)
        bb = torch.cat([corr[:, s], kp[t].expand(L, -1, -1)], 1)
        w = torch.sigmoid(torch.cat([logit[:, s], logit[:, t]], 1))
        ref = regtr_ref.compute_rigid_transform(a, bb, w)
        assert (pose[:, b] - ref).abs().max() < 1e-4
        R = pose[:, b, :, :3]
        assert (R @ R.transpose(1, 2) - torch.eye(3)).abs().max() < 1e-5 and (torch.det(R) - 1).abs().max() < 1e-5
    # generic drop-in of utils/se3_torch.py:compute_rigid_transform incl. a reflection-prone (planar) case
    a = torch.rand(5, 200, 3, generator=g); a[..., 2] *= 1e-3
    Rt = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    Rt = Rt * torch.det(Rt).sign()
    b = a @ Rt.T + torch.tensor([0.3, -1.0, 2.0])
    w = torch.rand(5, 200, generator=g)
    T = compute_rigid_transform(a.cuda(), b.cuda(), w.cuda()).cpu()
    assert (T - regtr_ref.compute_rigid_transform(a, b, w)).abs().max() < 1e-4
    assert (T[:, :, :3] - Rt).abs().max() < 1e-3
