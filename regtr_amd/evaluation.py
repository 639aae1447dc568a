"""3DMatch / 3DLoMatch registration recall from `est.log` files, in process and with numpy only.

Restates the Redwood / Predator evaluation protocol the reference runs after its test loop
(/root/reference/src/benchmark/benchmark_predator.py: read_trajectory :84-118, read_trajectory_info :120-149,
computeTransformationErr :60-81, evaluate_registration :223-282, benchmark :285-374; called from
models/generic_reg_model.py:180-186) without pandas / nibabel / torch: `test.py` prints the same table right after the
poses are gathered.  Quaternions follow nibabel.quaternions.mat2quat's convention (w >= 0), of which only the vector
part enters the error.
"""
import os

import numpy as np

SHORT_NAMES = ['Kitchen', 'Home 1', 'Home 2', 'Hotel 1', 'Hotel 2', 'Hotel 3', 'Study', 'MIT Lab']


def read_trajectory(filename, dim=4):
    """Redwood trajectory file -> (keys (n, 3) str array, traj (n, dim, dim))   (benchmark_predator.py:84-118)."""
    with open(filename) as f:
        lines = f.readlines()
    keys = [[s.strip() for s in ln.split('\t')[0:3]] for ln in lines[0::dim + 1]]
    traj = [ln.split('\t')[0:dim] for i, ln in enumerate(lines) if i % (dim + 1) != 0]
    return np.asarray(keys), np.asarray(traj, dtype=np.float64).reshape(-1, dim, dim)


def read_trajectory_info(filename, dim=6):
    """Redwood information file -> (number of fragments, (n, dim, dim) information matrices)   (:120-149)."""
    with open(filename) as fid:
        contents = fid.readlines()
    n_pairs = len(contents) // 7
    assert len(contents) == 7 * n_pairs
    info, n_frame = [], 0
    for i in range(n_pairs):
        _, _, n_frame = [int(item) for item in contents[i * 7].strip().split()]
        info.append(np.stack([np.array(item.split(), dtype=np.float64) for item in contents[i * 7 + 1:i * 7 + 7]]))
    return n_frame, np.asarray(info, dtype=np.float64).reshape(-1, dim, dim)


def mat2quat(M):
    """Rotation matrices (..., 3, 3) -> quaternions (..., 4) as (w, x, y, z) with w >= 0: eigenvector of the 4x4 K matrix for its largest
    eigenvalue (Bar-Itzhack), the method and sign convention of nibabel.quaternions.mat2quat -- for a whole stack at once (one batched
    symmetric eigen-decomposition; K is handed over as its lower triangle, which is all eigh reads)."""
    M = np.asarray(M, dtype=np.float64)
    lead = M.shape[:-2]
    R = M.reshape(-1, 3, 3)
    K = np.zeros((R.shape[0], 4, 4))
    K[:, 0, 0] = R[:, 0, 0] - R[:, 1, 1] - R[:, 2, 2]
    K[:, 1, 0] = R[:, 1, 0] + R[:, 0, 1]; K[:, 1, 1] = R[:, 1, 1] - R[:, 0, 0] - R[:, 2, 2]
    K[:, 2, 0] = R[:, 2, 0] + R[:, 0, 2]; K[:, 2, 1] = R[:, 2, 1] + R[:, 1, 2]; K[:, 2, 2] = R[:, 2, 2] - R[:, 0, 0] - R[:, 1, 1]
    K[:, 3, 0] = R[:, 2, 1] - R[:, 1, 2]; K[:, 3, 1] = R[:, 0, 2] - R[:, 2, 0]; K[:, 3, 2] = R[:, 1, 0] - R[:, 0, 1]
    K[:, 3, 3] = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    vals, vecs = np.linalg.eigh(K / 3.0)
    q = np.take_along_axis(vecs, np.argmax(vals, axis=1)[:, None, None], axis=2)[:, [3, 0, 1, 2], 0]
    q = np.where(q[:, :1] < 0, -q, q)
    return q.reshape(lead + (4,))


def transformation_error(trans, info):
    """RMSE proxy of the Redwood protocol, e^T I e / I[0,0] with e = (t, q_xyz)   (:60-81), for stacks (n, 4, 4) / (n, 6, 6) or single matrices."""
    trans, info = np.asarray(trans, np.float64), np.asarray(info, np.float64)
    single = trans.ndim == 2
    T, I = (trans[None], info[None]) if single else (trans, info)
    e = np.concatenate([T[:, :3, 3], mat2quat(T[:, :3, :3])[:, 1:]], axis=1)
    p = np.einsum('ni,nij,nj->n', e, I, e) / I[:, 0, 0]
    return float(p[0]) if single else p


def evaluate_registration(num_fragment, result, result_pairs, gt_pairs, gt, gt_info, err2=0.2):
    """(:223-282) -> precision, recall, flags (0 good / 1 bad / 2 not in gt), per-pair errors.  Whole scene at once: a fragment x
    fragment lookup table of the tested ground-truth pairs (non-consecutive only; like the reference's, it stores the ground-truth INDEX,
    so index 0 can never be matched), one batched inverse / quaternion / quadratic form for every estimated pair found in it."""
    thr = err2 ** 2
    gi, gj = np.asarray(gt_pairs[:, 0], dtype=np.int64), np.asarray(gt_pairs[:, 1], dtype=np.int64)
    table = np.zeros((num_fragment, num_fragment), dtype=np.int64)
    tested = np.nonzero(gj - gi > 1)[0]
    table[gi[tested], gj[tested]] = tested                      # (a repeated pair keeps its last index, as the reference's loop does)
    n_gt = int(np.count_nonzero(table))
    ri, rj = np.asarray(result_pairs[:, 0], dtype=np.int64), np.asarray(result_pairs[:, 1], dtype=np.int64)
    k = table[ri, rj]
    hit = k > 0
    errors = np.full(len(ri), np.nan)
    if hit.any():
        errors[hit] = transformation_error(np.linalg.inv(gt[k[hit]]) @ np.asarray(result, np.float64)[hit], gt_info[k[hit]])
    good = errors <= thr                                        # NaN (not in gt) compares False
    flags = np.where(hit, np.where(good, 0, 1), 2).tolist()
    n_good, n_res = int(good.sum()), int(hit.sum())
    return n_good / (n_res if n_res else 1e6), n_good / n_gt, flags, errors


def rotation_error_deg(R_gt, R_est):
    tr = np.clip((np.einsum('nji,nji->n', R_gt, R_est) - 1) / 2, -1, 1)       # trace(R_gt^T R_est)
    return np.degrees(np.arccos(tr))


def benchmark(est_folder, gt_folder):
    """(:285-374) -> (table string, mean recall over scenes).  est_folder/<scene>/est.log vs gt_folder/<scene>/gt.{log,info}."""
    scenes = sorted(os.listdir(gt_folder))
    missing = [sc for sc in scenes if not os.path.exists(os.path.join(est_folder, sc, 'est.log'))]
    if missing:       # a partial run (--max_pairs, an interrupted job): the reference would die on the first missing file (:301)
        raise RuntimeError(f'benchmark: no est.log for scene(s) {missing} under {est_folder}; the registration recall is defined '
                           'over the complete test set -- run every pair (no --max_pairs) before evaluating')
    out = 'Scene\t¦ prec.\t¦ rec.\t¦ re\t¦ te\t¦ samples\t¦\n'
    precision, recall, n_valids, med_re, med_te = [], [], [], [], []
    for idx, scene in enumerate(scenes):
        gt_pairs, gt_traj = read_trajectory(os.path.join(gt_folder, scene, 'gt.log'))
        gt_ij = gt_pairs[:, :2].astype(np.int64)
        n_valid = int(np.count_nonzero(np.abs(gt_ij[:, 0] - gt_ij[:, 1]) > 1))
        n_fragments, gt_info = read_trajectory_info(os.path.join(gt_folder, scene, 'gt.info'))
        est_pairs, est_traj = read_trajectory(os.path.join(est_folder, scene, 'est.log'))
        p, r, flags, _ = evaluate_registration(n_fragments, est_traj, est_pairs, gt_pairs, gt_traj, gt_info)
        # ground truth of every estimated pair (extract_corresponding_trajectors :152-171)
        where = np.full((n_fragments, n_fragments), -1, dtype=np.int64)
        where[gt_ij[:, 0], gt_ij[:, 1]] = np.arange(len(gt_ij))
        est_ij = est_pairs[:, :2].astype(np.int64)
        found = where[est_ij[:, 0], est_ij[:, 1]]
        ext = np.where((found >= 0)[:, None, None], gt_traj[np.maximum(found, 0)], 0.0)
        ok = np.asarray(flags) == 0
        re = rotation_error_deg(ext[:, :3, :3], est_traj[:, :3, :3])[ok]
        te = np.linalg.norm(ext[:, :3, 3] - est_traj[:, :3, 3], axis=1)[ok]
        precision.append(p); recall.append(r); n_valids.append(n_valid)
        med_re.append(np.median(re) if len(re) else np.nan); med_te.append(np.median(te) if len(te) else np.nan)
        name = SHORT_NAMES[idx] if idx < len(SHORT_NAMES) else scene
        out += '{}\t¦ {:.3f}\t¦ {:.3f}\t¦ {:.3f}\t¦ {:.3f}\t¦ {:3d}¦\n'.format(name, p, r, med_re[-1], med_te[-1], n_valid)
    out += 'Mean precision: {:.3f}: +- {:.3f}\n'.format(np.mean(precision), np.std(precision))
    out += 'Weighted precision: {:.3f}\n'.format((np.array(n_valids) * np.array(precision)).sum() / np.sum(n_valids))
    out += 'Mean median RRE: {:.3f}: +- {:.3f}\n'.format(np.nanmean(med_re), np.nanstd(med_re))
    out += 'Mean median RTE: {:.3F}: +- {:.3f}\n'.format(np.nanmean(med_te), np.nanstd(med_te))
    return out, float(np.mean(recall))


# ------------------------------------------------------------------------------------------------------ ModelNet metrics
def modelnet_metrics(pred, gt, points_src, points_ref, points_raw):
    """benchmark/benchmark_modelnet.py:33-90 (compute_metrics, RPMNet / DCP conventions) in numpy.
    pred, gt (B, 3, 4); points_* (B, N, >=3).  Returns the per-instance metric arrays of the reference."""
    from scipy.spatial import cKDTree
    from scipy.spatial.transform import Rotation
    pred, gt = np.asarray(pred, np.float64), np.asarray(gt, np.float64)
    src, ref, raw = (np.asarray(p, np.float64)[..., :3] for p in (points_src, points_ref, points_raw))
    e_gt = Rotation.from_matrix(gt[:, :3, :3]).as_euler('xyz', degrees=True)
    e_pr = Rotation.from_matrix(pred[:, :3, :3]).as_euler('xyz', degrees=True)
    t_gt, t_pr = gt[:, :3, 3], pred[:, :3, 3]

    def cat(a, b):        # a o b  (se3_torch.se3_cat)
        return np.concatenate([a[:, :, :3] @ b[:, :, :3], a[:, :, :3] @ b[:, :, 3:] + a[:, :, 3:]], axis=2)

    def inv(a):
        Rt = np.swapaxes(a[:, :, :3], 1, 2)
        return np.concatenate([Rt, -Rt @ a[:, :, 3:]], axis=2)

    def apply(a, p):
        return p @ np.swapaxes(a[:, :, :3], 1, 2) + a[:, None, :, 3]

    c = cat(inv(gt), pred)
    err_r = np.degrees(np.arccos(np.clip(0.5 * (np.trace(c[:, :, :3], axis1=1, axis2=2) - 1), -1, 1)))
    err_t = np.linalg.norm(c[:, :, 3], axis=1)
    src_t = apply(pred, src)
    src_clean = apply(cat(pred, inv(gt)), raw)
    chamfer = np.array([np.mean(cKDTree(raw[b]).query(src_t[b])[0] ** 2) + np.mean(cKDTree(src_clean[b]).query(ref[b])[0] ** 2)
                        for b in range(len(pred))])
    return {'r_mse': np.mean((e_gt - e_pr) ** 2, axis=1), 'r_mae': np.mean(np.abs(e_gt - e_pr), axis=1),
            't_mse': np.mean((t_gt - t_pr) ** 2, axis=1), 't_mae': np.mean(np.abs(t_gt - t_pr), axis=1),
            'err_r_deg': err_r, 'err_t': err_t, 'chamfer_dist': chamfer}


def summarize_metrics(metrics):
    """benchmark_modelnet.py:93-105."""
    out = {}
    for k, v in metrics.items():
        if k.endswith('mse'):
            out[k[:-3] + 'rmse'] = np.sqrt(np.mean(v))
        elif k.startswith('err'):
            out[k + '_mean'] = np.mean(v)
            out[k + '_rmse'] = np.sqrt(np.mean(v ** 2))
        else:
            out[k] = np.mean(v)
    return out
