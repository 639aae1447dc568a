"""GPU (-m gpu): BASELINE.json configs[4] -- a synthetic 100k + 100k-point pair -- checked through size-independent
properties (the brute-force oracle is quadratic and would take minutes at this size): neighbour rows against an exact
float32 brute-force on sampled queries, sortedness / radius / padding invariants on every row, subsampling bit-exact
against the (linear-time) C++ oracle, and forward invariants (batch independence, finite orthonormal poses)."""
import numpy as np
import pytest
import torch

from tests.util import load_cfg, seeded_sd, seg_of, to_dev

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
