"""Deterministic synthetic scan pairs of 3DMatch size and statistics -- the workload of
bench.py and of `test.py --synthetic N`.  numpy only."""
import numpy as np


def _morton(ijk):
    ijk = ijk - ijk.min(0)
    key = np.zeros(len(ijk), np.int64)
    for b in range(16):
        for a in range(3):
            key |= ((ijk[:, a] >> b) & 1) << (3 * b + a)
    return key


# Multi-scale occlusion (round 6).  Full planes thin out faster per pyramid level than real scans and are denser inside a ball: the round-5 rooms
# gave 38 197 / 8 507 / 2 201 / 581 points per level and 37 neighbours in r = 0.0625 (21.6 % of level-0 rows over K = 40, 39-47 % deeper) where the
# shipped red-kitchen pair (/root/reference/src/demo.py:26-49 example 0) has 38 061 / 9 977 / 2 741 / 749 and 30.5 neighbours
# (9.2 % / 5.9 % / 6.2 % of rows over K) -- real fragments are patchy (occlusion shadows, holes, thin structures: fractal dimension ~1.9, not 2).
# Cells of each size below are dropped with the given probability before voxelisation (the room grows to keep N_0): four-pair means
# 37 393 / 9 718 / 2 778 / 762 = 0.98 / 0.97 / 1.01 / 1.02 of the kitchen pair, 31.2 / 30.8 neighbours at levels 0 / 1, 4.6 % / 7.7 % of rows over K.
OCCLUSION = ((0.035, 0.15), (0.07, 0.20), (0.14, 0.10), (0.28, 0.03))      # (cell size [m], drop probability)


def _occlusion_mask(rng, p):
    keep = np.ones(len(p), bool)
    for size, q in OCCLUSION:
        c = np.floor((p + rng.uniform(0, size, 3)) / size).astype(np.int64)
        c -= c.min(0)
        span = c.max(0) + 1
        _, inv = np.unique((c[:, 0] * span[1] + c[:, 1]) * span[2] + c[:, 2], return_inverse=True)
        keep &= ~(rng.random(inv.max() + 1) < q)[inv]
    return keep


def _scene_once(rng, side, voxel):
    surf = []

    def rect(o, u, v, n):
        ab = rng.random((n, 2))
        return o + ab[:, :1] * u + ab[:, 1:] * v

    dens = 10.0 / (voxel * voxel)
    X, Y, Z = 1.6 * side, 1.2 * side, 0.9 * side
    surf.append(rect(np.zeros(3), np.array([X, 0, 0]), np.array([0, Y, 0]), int(X * Y * dens)))          # floor
    surf.append(rect(np.zeros(3), np.array([X, 0, 0]), np.array([0, 0, Z]), int(X * Z * dens)))          # wall
    surf.append(rect(np.zeros(3), np.array([0, Y, 0]), np.array([0, 0, Z]), int(Y * Z * dens)))          # wall
    for _ in range(3):                                                                                   # furniture
        o = np.array([rng.uniform(0.1 * X, 0.7 * X), rng.uniform(0.1 * Y, 0.7 * Y), 0.0])
        w, d, h = rng.uniform(0.15, 0.3, 3) * side
        surf.append(rect(o + [0, 0, h], np.array([w, 0, 0]), np.array([0, d, 0]), int(w * d * dens)))
        surf.append(rect(o, np.array([w, 0, 0]), np.array([0, 0, h]), int(w * h * dens)))
        surf.append(rect(o, np.array([0, d, 0]), np.array([0, 0, h]), int(d * h * dens)))
    p = np.concatenate(surf)
    p = p + rng.normal(scale=0.002, size=p.shape)
    p = p[_occlusion_mask(rng, p)]
    key = np.floor(p / voxel).astype(np.int64)
    key -= key.min(0)
    span = key.max(0) + 1
    lin = (key[:, 0] * span[1] + key[:, 1]) * span[2] + key[:, 2]      # same lexicographic voxel order as unique(axis=0)
    _, inv, cnt = np.unique(lin, return_inverse=True, return_counts=True)
    out = np.stack([np.bincount(inv, weights=p[:, a], minlength=len(cnt)) for a in range(3)], 1)
    out /= cnt[:, None]
    out = out[np.argsort(_morton(np.floor(out / (4 * voxel)).astype(np.int64)), kind='stable')]
    # rows are in Morton order of 10 cm blocks: real 3DMatch fragments are spatially coherent too (median |i - j| between
    # neighbours is ~80 rows on the shipped red-kitchen clouds); --shuffle gives the adversarial random order
    return out, X


def synth_scene(rng, target_pts, voxel=0.025):
    """Planes and boxes of a room corner sampled densely, thinned by the multi-scale occlusion mask, then voxel-averaged at 2.5 cm like the 3DMatch fragments.
    The room size is calibrated (deterministically, from the seed) so that the scene holds ~target_pts points."""
    side = np.sqrt(target_pts * voxel * voxel / 7.0)
    for _ in range(4):
        out, X = _scene_once(rng, side, voxel)
        if abs(len(out) - target_pts) < 0.03 * target_pts:
            break
        side *= np.sqrt(target_pts / len(out))
    return out, X


def random_se3(rng, rot_deg=45.0, trans=0.5):
    axis = rng.standard_normal(3); axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(0, rot_deg))
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    t = rng.standard_normal(3); t = t / np.linalg.norm(t) * rng.uniform(0, trans)
    return R, t


def synth_pair(pair_id, pts_per_cloud=20000, shuffle=False, return_pose=False, overlap=None):
    """Two overlapping views (72 % of the scene each) of one synthetic room, the target moved by a random SE(3) (rotation
    <= 45 deg, |t| <= 0.5 m) and both jittered by N(0, 5 mm) (3dmatch.yaml:7).  Deterministic in pair_id.
    return_pose: also return the ground-truth (3, 4) transform taking src into the tgt frame.
    overlap: fraction of each view's extent shared with the other (default None = the 3DMatch-like 0.61); 'lomatch' draws it
    uniformly from [0.10, 0.30] per pair, the overlap range of the 3DLoMatch test list; the
    room grows so that each view still holds ~pts_per_cloud points."""
    rng = np.random.default_rng(1000 + pair_id)
    if overlap == 'lomatch':
        overlap = float(np.random.default_rng(5000 + pair_id).uniform(0.10, 0.30))
    f = 0.72 if overlap is None else 1.0 / (2.0 - float(overlap))         # each view spans [0, f] / [1 - f, 1] of the room's length
    scene, X = synth_scene(rng, int(pts_per_cloud / f))
    src = scene[scene[:, 0] < f * X]
    tgt = scene[scene[:, 0] > (1.0 - f) * X]
    R, t = random_se3(rng)
    tgt = tgt @ R.T + t + rng.normal(scale=0.005, size=tgt.shape)        # augment_noise 0.005 (3dmatch.yaml:7)
    src = src + rng.normal(scale=0.005, size=src.shape)
    if shuffle:
        src, tgt = src[rng.permutation(len(src))], tgt[rng.permutation(len(tgt))]
    if return_pose:
        return src.astype(np.float32), tgt.astype(np.float32), np.concatenate([R, t[:, None]], 1).astype(np.float32)
    return src.astype(np.float32), tgt.astype(np.float32)


def synth_modelnet_pair(pair_id, num_points=1024, keep=0.7, return_pose=False):
    """ModelNet40-benchmark-sized pair (conf/modelnet.yaml: num_points 1024, partial [0.7, 0.7],
    rot_mag 45, trans_mag 0.5): points on a random union of 3-6 ellipsoid / box surfaces inside the unit cube, two
    independent 70 % half-space crops (~717 points each), the target moved by a random SE(3) and both jittered by
    N(0, 0.01) clipped at 0.05.  Deterministic in pair_id."""
    rng = np.random.default_rng(7000 + pair_id)
    parts = []
    n_prim = int(rng.integers(3, 7))
    per = 4 * num_points // n_prim + 1
    for _ in range(n_prim):
        c = rng.uniform(-0.45, 0.45, 3)
        r = rng.uniform(0.15, 0.5, 3)
        if rng.random() < 0.5:                                  # ellipsoid surface
            v = rng.standard_normal((per, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
            parts.append(c + v * r)
        else:                                                   # box surface
            p = rng.uniform(-1, 1, (per, 3))
            ax = rng.integers(0, 3, per)
            p[np.arange(per), ax] = np.sign(rng.standard_normal(per))
            parts.append(c + p * r)
    obj = np.concatenate(parts)
    obj = obj / np.abs(obj).max()                               # unit scale
    obj = obj[rng.permutation(len(obj))]

    def crop(points):
        d = rng.standard_normal(3); d /= np.linalg.norm(d)
        proj = points @ d
        return points[proj > np.quantile(proj, 1.0 - keep)]
    src = crop(obj[rng.choice(len(obj), num_points, replace=False)])
    tgt = crop(obj[rng.choice(len(obj), num_points, replace=False)])
    R, t = random_se3(rng)
    jit = lambda p: p + np.clip(rng.normal(scale=0.01, size=p.shape), -0.05, 0.05)
    src, tgt = jit(src), jit(tgt @ R.T + t)
    if return_pose:
        return src.astype(np.float32), tgt.astype(np.float32), np.concatenate([R, t[:, None]], 1).astype(np.float32)
    return src.astype(np.float32), tgt.astype(np.float32)
