"""TEST INFRASTRUCTURE ONLY -- golden vectors for regtr_amd/evaluation.py: a small synthetic Redwood-format benchmark (2 scenes)
evaluated by the REFERENCE's own /root/reference/src/benchmark/benchmark_predator.py (run here, in this container; numpy 2
removed np.float / np.int and nibabel is absent, so those three names are provided: np.float = float, np.int = int,
nibabel.quaternions.mat2quat via scipy with nibabel's w >= 0 convention).  Writes tests/golden/predator_eval.npz.
    python oracle/make_golden_eval.py"""
import os
import sys
import tempfile
import types

import numpy as np
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pose(rng, rot_deg, trans):
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(rng.standard_normal(3) * np.deg2rad(rot_deg) / 3).as_matrix()
    T[:3, 3] = rng.standard_normal(3) * trans
    return T


def make_files(folder, rng, scenes=('scene-a', 'scene-b'), n_frag=(9, 7)):
    texts = {}
    for scene, nf in zip(scenes, n_frag):
        gt_log, gt_info, est_log = '', '', ''
        for i in range(nf):
            for j in range(i + 1, nf):
                if rng.random() < 0.55:
                    continue
                T = _pose(rng, 60, 1.0)
                A = rng.standard_normal((6, 6)); info = A @ A.T * 50 + np.eye(6) * 200
                gt_log += f'{i}\t{j}\t{nf}\n' + ''.join('\t'.join(f'{v:.8e}' for v in row) + '\n' for row in T)
                gt_info += f'{i}\t{j}\t{nf}\n' + ''.join('\t'.join(f'{v:.8e}' for v in row) + '\n' for row in info)
                if rng.random() < 0.9:       # most pairs are estimated: small / large perturbations of the ground truth
                    E = T @ _pose(rng, *((2, 0.02) if rng.random() < 0.6 else (25, 0.4)))
                    est_log += f'{i}\t{j}\t-1\n' + ''.join('\t'.join(f'{v:.12f}' for v in row) + '\n' for row in E)
        texts[scene] = (gt_log, gt_info, est_log)
        os.makedirs(os.path.join(folder, 'gt', scene)); os.makedirs(os.path.join(folder, 'est', scene))
        open(os.path.join(folder, 'gt', scene, 'gt.log'), 'w').write(gt_log)
        open(os.path.join(folder, 'gt', scene, 'gt.info'), 'w').write(gt_info)
        open(os.path.join(folder, 'est', scene, 'est.log'), 'w').write(est_log)
    return texts


def main():
    np.float, np.int = float, int            # removed from numpy 2; the reference still uses them
    nq = types.ModuleType('nibabel.quaternions')

    def mat2quat(M):
        x, y, z, w = Rotation.from_matrix(np.asarray(M)).as_quat()
        q = np.array([w, x, y, z])
        return -q if q[0] < 0 else q
    nq.mat2quat = mat2quat
    nib = types.ModuleType('nibabel'); nib.quaternions = nq
    sys.modules['nibabel'] = nib; sys.modules['nibabel.quaternions'] = nq
    sys.path.insert(0, '/root/reference/src')
    from benchmark import benchmark_predator as ref
    rng = np.random.default_rng(2024)
    with tempfile.TemporaryDirectory() as d:
        texts = make_files(d, rng)
        out = {}
        for scene in texts:
            gt_pairs, gt_traj = ref.read_trajectory(os.path.join(d, 'gt', scene, 'gt.log'))
            n_frag, gt_info = ref.read_trajectory_info(os.path.join(d, 'gt', scene, 'gt.info'))
            est_pairs, est_traj = ref.read_trajectory(os.path.join(d, 'est', scene, 'est.log'))
            p, r, flags, errs = ref.evaluate_registration(n_frag, est_traj, est_pairs, gt_pairs, gt_traj, gt_info)
            out[scene] = (p, r, np.asarray(flags), errs)
        table, mean_recall = ref.benchmark(os.path.join(d, 'est'), os.path.join(d, 'gt'))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'predator_eval.npz'),
                        scenes=np.array(list(texts)), gt_log=np.array([texts[s][0] for s in texts]),
                        gt_info=np.array([texts[s][1] for s in texts]), est_log=np.array([texts[s][2] for s in texts]),
                        precision=np.array([out[s][0] for s in texts]), recall=np.array([out[s][1] for s in texts]),
                        flags=np.array([out[s][2] for s in texts], dtype=object), errors=np.array([out[s][3] for s in texts], dtype=object),
                        table=np.array(table), mean_recall=mean_recall)
    print(table, mean_recall)
    modelnet_golden()


def modelnet_golden():
    """benchmark/benchmark_modelnet.py compute_metrics + summarize_metrics of the reference on seeded random inputs."""
    import torch
    sys.path.insert(0, ROOT)
    from oracle import ref_loader
    ref_loader.load()                                   # mocks for the absent optional dependencies (cvhelpers, ...)
    import cvhelpers.torch_helpers as th
    th.to_numpy = lambda x: x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)
    import importlib
    bm = importlib.import_module('benchmark.benchmark_modelnet')
    bm.to_numpy = th.to_numpy
    rng = np.random.default_rng(7)
    B, N = 4, 300
    raw = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    gt = np.stack([_pose(rng, 45, 0.5)[:3] for _ in range(B)]).astype(np.float32)
    pred = np.stack([(np.vstack([g, [0, 0, 0, 1]]) @ _pose(rng, 4, 0.03))[:3] for g in gt]).astype(np.float32)
    ref = raw[:, :220] + rng.normal(scale=0.01, size=(B, 220, 3)).astype(np.float32)
    Rg, tg = gt[:, :, :3], gt[:, :, 3]
    src = (np.einsum('bij,bnj->bni', np.transpose(Rg, (0, 2, 1)), raw[:, 80:] - tg[:, None]) +
           rng.normal(scale=0.01, size=(B, N - 80, 3))).astype(np.float32)       # inverse gt applied: src -> ref under gt
    data = {'transform_gt': torch.from_numpy(gt), 'points_src': torch.from_numpy(src), 'points_ref': torch.from_numpy(ref),
            'points_raw': torch.from_numpy(raw)}
    m = bm.compute_metrics(data, torch.from_numpy(pred))
    sm = bm.summarize_metrics(m)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'modelnet_metrics.npz'), gt=gt, pred=pred, src=src, ref=ref, raw=raw,
                        **{'m_' + k: np.asarray(v) for k, v in m.items()}, **{'s_' + k: np.asarray(v) for k, v in sm.items()})
    print({k: float(v) for k, v in sm.items()})


if __name__ == '__main__':
    main()
