"""TEST INFRASTRUCTURE ONLY -- the product's CANONICAL kpconv_meta for clouds of any size, built from the reference's own neighbour
SETS: first-appearance subsampling from the linear-time C++ restatement (oracle/regtr_oracle.cpp, bit-exact vs the unmodified
reference C++ as multisets, tests/test_oracle.py) and, where oracle/_ref is present, the unmodified reference C++'s KD-tree radius
search (neighbors.cpp:211-332) with every row re-ordered to ascending (d2, index) and cut at K = neighborhood_limits[l] -- the order
DESIGN.md section 4 documents.  Without oracle/_ref the quadratic brute-force restatement produces the same tables (slower).
Used by the parity tests at bench / stress sizes and by bench.py's post-run parity check; never by the product."""
import numpy as np
import torch

from . import native


def canon_table(idx, q, s, pad, K=None):
    """Every row sorted by (d2 in the reference's float32 arithmetic -- nanoflann.hpp:432-440 order --, index), padding last; optionally
    cut / padded to K columns.  Vectorised (100k-row tables)."""
    idx = np.asarray(idx).astype(np.int64)
    s_pad = np.concatenate([np.asarray(s, np.float32), np.zeros((1, 3), np.float32)])
    isp = idx == pad
    d = q[:, None, :].astype(np.float32) - s_pad[np.minimum(idx, len(s_pad) - 1)]
    d2 = ((np.float32(0) + d[..., 0] * d[..., 0]) + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    key = (d2.view(np.uint32).astype(np.int64) << 32) | idx
    key[isp] = np.iinfo(np.int64).max
    out = np.take_along_axis(idx, np.argsort(key, axis=1, kind='stable'), axis=1)
    if K is not None:
        if out.shape[1] < K:
            out = np.concatenate([out, np.full((out.shape[0], K - out.shape[1]), pad, np.int64)], 1)
        out = out[:, :K]
    return out.astype(np.int32)


def canonical_meta(pts_list, cfg):
    """kpconv_meta (kpconv.py:406-412) of the clouds in `pts_list` in the product's canonical orders, as torch CPU tensors."""
    pts = np.concatenate([np.asarray(p, np.float32) for p in pts_list]).astype(np.float32)
    lens = np.array([len(p) for p in pts_list], np.int32)
    limits = cfg['neighborhood_limits']
    r = cfg['first_subsampling_dl'] * cfg['conv_radius']                                   # kpconv.py:315
    meta = {k: [] for k in ('points', 'neighbors', 'pools', 'upsamples', 'stack_lengths')}
    n_levels = 1 + sum(('strided' in b or 'pool' in b) for b in cfg['architecture'])
    use_ref = native.have_ref()

    def table(q, s, ql, sl, K):
        if use_ref:
            return canon_table(native.ref_batch_query(q, s, ql, sl, r), q, s, len(s), K)
        return native.radius_neighbors(q, s, ql, sl, r, K)[0]

    for l in range(n_levels):
        K = limits[l]
        conv = table(pts, pts, lens, lens, K)                                              # :349-351
        if l + 1 < n_levels:
            sub, sl = native.grid_subsample(pts, lens, 2 * r / cfg['conv_radius'],             # :363-366
                                            key_mode=native.VOXEL_KEY_MODES[cfg.get('kpconv_voxel_key', 'origin')])
            pool = table(sub, pts, sl, lens, K)                                            # :376
        else:
            sub, sl, pool = np.zeros((0, 3), np.float32), np.zeros(0, np.int32), np.zeros((0, 1), np.int32)
        meta['points'].append(torch.from_numpy(pts)); meta['neighbors'].append(torch.from_numpy(conv.astype(np.int64)))
        meta['pools'].append(torch.from_numpy(pool.astype(np.int64))); meta['upsamples'].append(torch.zeros((0, 1), dtype=torch.int64))
        meta['stack_lengths'].append(torch.from_numpy(lens.astype(np.int64)))
        pts, lens, r = sub, sl, r * 2
    return meta


def pair_tables_of_batch(meta, B, b):
    """The rows of pair `b` cut out of a BATCHED kpconv_meta (clouds stacked [src_0..src_{B-1}, tgt_0..tgt_{B-1}], regtr.py:117) and
    re-indexed as if the pair had been preprocessed alone ([src_b, tgt_b]; pad = that pair's support count): what canonical_meta of
    the single pair must equal.  -> {'points': [...], 'neighbors': [...], 'pools': [...]} numpy arrays per level."""
    out = {'points': [], 'neighbors': [], 'pools': []}
    L = len(meta['points'])
    lens = [np.asarray(meta['stack_lengths'][l].cpu()).astype(np.int64) for l in range(L)]
    off = [np.concatenate([[0], np.cumsum(x)]) for x in lens]

    def cut(tab, l_rows, l_sup):
        tab = np.asarray(tab.cpu()).astype(np.int64)
        n_sup_total = int(off[l_sup][-1])
        rows = []
        base = 0
        for c in (b, B + b):
            t = tab[off[l_rows][c]:off[l_rows][c + 1]]
            lo, hi = off[l_sup][c], off[l_sup][c + 1]
            real = t < n_sup_total
            assert ((t[real] >= lo) & (t[real] < hi)).all(), 'a neighbour outside the query\'s own cloud'
            rows.append(np.where(real, t - lo + base, -1))
            base += hi - lo
        t = np.concatenate(rows)
        t[t < 0] = base                                                                    # pad = supports of the pair
        return t.astype(np.int32)

    for l in range(L):
        pts = np.asarray(meta['points'][l].cpu())
        out['points'].append(np.concatenate([pts[off[l][c]:off[l][c + 1]] for c in (b, B + b)]))
        out['neighbors'].append(cut(meta['neighbors'][l], l, l))
        out['pools'].append(cut(meta['pools'][l], l + 1, l) if l + 1 < L and meta['pools'][l].numel() > 1 else None)
    return out
