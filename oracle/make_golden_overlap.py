"""TEST INFRASTRUCTURE ONLY -- tests/golden/overlap_pairs.npz: the REFERENCE's own compute_overlap (utils/pointcloud.py:8-65), executed
from /root/reference, over a stand-in for the one thing it needs from open3d (absent here): `o3d.geometry.KDTreeFlann(pcd).
search_radius_vector_3d(query, radius)` -> (count, indices sorted by distance, squared distances), served by scipy.spatial.cKDTree -- an
independent KD-tree library, float64 like open3d's -- with hits ordered by (distance, index).  Everything else that runs is the
reference's code: both search directions, `knn_indices[0]`, the mutual check with its `src_corr > 0` (index 0 can never be mutual), the
output layout.  What stays unpinned is open3d's own choice among EXACTLY equidistant hits.  Re-run: python -m oracle.make_golden_overlap"""
import importlib
import os
import sys
import types

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF_SRC = '/root/reference/src'
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _open3d_stand_in():
    o3d = types.ModuleType('open3d')
    geometry, utility = types.ModuleType('open3d.geometry'), types.ModuleType('open3d.utility')

    class PointCloud:
        def __init__(self):
            self.points = None

    class KDTreeFlann:
        def __init__(self, pcd):
            self.xyz = np.asarray(pcd.points, dtype=np.float64)
            self.tree = cKDTree(self.xyz)

        def search_radius_vector_3d(self, query, radius):
            q = np.asarray(query, dtype=np.float64)
            idx = np.asarray(self.tree.query_ball_point(q, radius), dtype=np.int64)
            d2 = ((self.xyz[idx] - q) ** 2).sum(1) if len(idx) else np.zeros(0)
            keep = d2 < float(radius) ** 2                       # nanoflann's strict comparison (open3d's KDTreeFlann is nanoflann)
            idx, d2 = idx[keep], d2[keep]
            order = np.lexsort((idx, d2))
            return len(idx), idx[order], d2[order]

    geometry.PointCloud, geometry.KDTreeFlann = PointCloud, KDTreeFlann
    utility.Vector3dVector = lambda a: np.asarray(a, dtype=np.float64)
    o3d.geometry, o3d.utility = geometry, utility
    return {'open3d': o3d, 'open3d.geometry': geometry, 'open3d.utility': utility}


def main():
    sys.modules.update(_open3d_stand_in())
    sys.path.insert(0, REF_SRC)
    pc = importlib.import_module('utils.pointcloud')              # the reference's own module
    from tests.util import synth_cloud
    rng = np.random.default_rng(17)
    out = {}
    cases = []
    a = synth_cloud(rng, 4000, lattice=0.0)
    b = (a[rng.permutation(4000)[:3300]] + rng.normal(scale=0.004, size=(3300, 3))).astype(np.float32)
    cases.append((a, b, 0.0375))                                  # a jittered, permuted copy: dense mutual matches
    c = synth_cloud(rng, 2500)
    d = np.concatenate([c[::2] + np.float32(0.01), synth_cloud(rng, 900) + 6.0]).astype(np.float32)
    cases.append((c, d, 0.05))                                    # partial overlap: a far-away part with no correspondences
    e = synth_cloud(rng, 300)
    cases.append((e, e + 10.0, 0.05))                             # disjoint
    g = np.load(os.path.join(GOLD, '3dmatch_crop.npz'))
    cases.append((g['src'][::3].copy(), g['tgt'][::3].copy(), 0.0375))      # real fragments (unregistered: few hits)
    for i, (s, t, r) in enumerate(cases):
        hs, ht, corr = pc.compute_overlap(s.astype(np.float32), t.astype(np.float32), r)
        out[f'src_{i}'], out[f'tgt_{i}'], out[f'radius_{i}'] = s.astype(np.float32), t.astype(np.float32), np.float64(r)
        out[f'has_src_{i}'], out[f'has_tgt_{i}'], out[f'corr_{i}'] = hs, ht, corr.astype(np.int64)
        print(i, s.shape, t.shape, r, int(hs.sum()), int(ht.sum()), corr.shape)
    out['n_cases'] = np.int64(len(cases))
    np.savez_compressed(os.path.join(GOLD, 'overlap_pairs.npz'), **out)


if __name__ == '__main__':
    main()
