"""regtr_amd/evaluation.py against golden outputs of the reference's own benchmark_predator.py (tests/golden/predator_eval.npz,
written by oracle/make_golden_eval.py): per-scene precision / recall, per-pair flags and errors, and the printed table."""
import os

import numpy as np


def test_predator_benchmark_matches_reference(tmp_path):
    from regtr_amd import evaluation as ev
    from tests.util import GOLD
    g = np.load(os.path.join(GOLD, 'predator_eval.npz'), allow_pickle=True)
    for s, gl, gi, el in zip(g['scenes'], g['gt_log'], g['gt_info'], g['est_log']):
        os.makedirs(tmp_path / 'gt' / str(s)); os.makedirs(tmp_path / 'est' / str(s))
        open(tmp_path / 'gt' / str(s) / 'gt.log', 'w').write(str(gl))
        open(tmp_path / 'gt' / str(s) / 'gt.info', 'w').write(str(gi))
        open(tmp_path / 'est' / str(s) / 'est.log', 'w').write(str(el))
    for k, s in enumerate(g['scenes']):
        gt_pairs, gt_traj = ev.read_trajectory(str(tmp_path / 'gt' / str(s) / 'gt.log'))
        n_frag, gt_info = ev.read_trajectory_info(str(tmp_path / 'gt' / str(s) / 'gt.info'))
        est_pairs, est_traj = ev.read_trajectory(str(tmp_path / 'est' / str(s) / 'est.log'))
        p, r, flags, errs = ev.evaluate_registration(n_frag, est_traj, est_pairs, gt_pairs, gt_traj, gt_info)
        assert abs(p - g['precision'][k]) < 1e-12 and abs(r - g['recall'][k]) < 1e-12
        assert np.array_equal(np.asarray(flags), g['flags'][k])
        assert np.allclose(errs, g['errors'][k].astype(np.float64), rtol=1e-9, atol=1e-12, equal_nan=True)
    table, mean_recall = ev.benchmark(str(tmp_path / 'est'), str(tmp_path / 'gt'))
    assert abs(mean_recall - float(g['mean_recall'])) < 1e-12
    assert table == str(g['table'])


def test_est_log_roundtrip_through_evaluation(tmp_path):
    """write_est_log (test.py's writer) -> read_trajectory: poses survive with 12 decimals, keys are (tgt, src, -1)."""
    from regtr_amd import evaluation as ev
    from regtr_amd.harness import write_est_log
    rng = np.random.default_rng(3)
    poses = rng.standard_normal((3, 3, 4))
    write_est_log(str(tmp_path), '3DMatch', [
        {'src_path': f'test/sc/cloud_bin_{2 * i + 1}.pth', 'tgt_path': f'test/sc/cloud_bin_{2 * i}.pth', 'pose': poses[i]} for i in range(3)])
    keys, traj = ev.read_trajectory(str(tmp_path / '3DMatch' / 'sc' / 'est.log'))
    assert keys.tolist() == [['0', '1', '-1'], ['2', '3', '-1'], ['4', '5', '-1']]
    assert np.allclose(traj[:, :3, :], poses, atol=1e-11) and np.allclose(traj[:, 3], [0, 0, 0, 1])


def test_modelnet_metrics_match_reference():
    """benchmark_modelnet.compute_metrics / summarize_metrics of the reference (golden, float32 torch) vs the numpy restatement."""
    from regtr_amd import evaluation as ev
    from tests.util import GOLD
    g = np.load(os.path.join(GOLD, 'modelnet_metrics.npz'))
    m = ev.modelnet_metrics(g['pred'], g['gt'], g['src'], g['ref'], g['raw'])
    for k, v in m.items():
        assert np.allclose(v, g['m_' + k], rtol=2e-4, atol=2e-6), (k, v, g['m_' + k])
    for k, v in ev.summarize_metrics(m).items():
        assert np.allclose(v, g['s_' + k], rtol=2e-4, atol=2e-6), k
