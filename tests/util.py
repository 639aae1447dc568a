"""Shared helpers for the parity tests (test infrastructure; may use oracle/)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def load_cfg(name):
    from regtr_amd.config import load_config
    return load_config(os.path.join(ROOT, 'regtr_amd', 'conf', f'{name}.yaml'))


def seeded_sd(cfg, seed=0):
    from oracle import seeded_weights
    from regtr_amd.kernel_points import K015_CENTER
    return seeded_weights.seeded_state_dict(cfg, seed, K015_CENTER)


def gold(name):
    return np.load(os.path.join(GOLD, f'{name}.npz'))


def synth_cloud(rng, n, extent=2.0, lattice=0.0):
    """Points on a few random planes inside a box (room-like), optionally snapped to a lattice (exact ties)."""
    pts = []
    per = n // 4 + 1
    for _ in range(4):
        o = rng.uniform(-extent / 2, extent / 2, 3)
        u, v = rng.standard_normal(3), rng.standard_normal(3)
        u /= np.linalg.norm(u); v -= u * (u @ v); v /= np.linalg.norm(v)
        ab = rng.uniform(-extent / 2, extent / 2, (per, 2))
        pts.append(o + ab[:, :1] * u + ab[:, 1:] * v)
    p = np.concatenate(pts)[:n].astype(np.float32)
    if lattice > 0:
        p = (np.round(p / lattice) * lattice).astype(np.float32)
    return p


def canon_rows(idx, q, s, pad):
    """Canonicalise a reference neighbour table: sort every row by (d2 in the reference's float32 arithmetic, index)."""
    idx = np.asarray(idx)
    out = np.full_like(idx, pad)
    s_pad = np.concatenate([s, np.full((1, 3), 1e18, np.float32)])
    for i in range(idx.shape[0]):
        row = idx[i][idx[i] != pad]
        d = q[i] - s_pad[row]
        d2 = ((np.float32(0) + d[:, 0] * d[:, 0]) + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        o = np.lexsort((row, d2))
        out[i, :len(row)] = row[o]
    return out


def canon_table(idx, q, s, pad, K=None):
    from oracle.canonical import canon_table as f
    return f(idx, q, s, pad, K)


def to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def seg_of(lens):
    return torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
