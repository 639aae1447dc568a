"""CPU: bench.py's multi-GPU entry logic -- self-launch under torch.distributed.run, world-size check, pose gather, the
reported n_gpus -- on gloo with a stand-in forward (`--stub-backend gloo`; the real forward needs MI355X kernels)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, env=env,
                          timeout=300)


def test_plain_invocation_self_launches_two_ranks():
    r = _run(['--gpus', '2', '--steps', '3', '--warmup', '1', '--pairs', '5', '--stub-backend', 'gloo'])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(line) == 1, r.stdout                       # rank 0 only
    d = json.loads(line[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['value'] > 0


def test_world_size_mismatch_is_an_error():
    r = _run(['--gpus', '2', '--stub-backend', 'gloo'], {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'refusing' in (r.stderr + r.stdout)
    r = _run(['--gpus', '1', '--stub-backend', 'gloo', '--steps', '1', '--pairs', '2'])
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])['n_gpus'] == 1


def test_lomatch_set_is_sharded_and_gathered_every_pass():
    """--config lomatch (BASELINE configs[3]): 23 pairs over 2 ranks (12 + 11: ragged shards), at most 5 per forward (rank 0: 4 + 4 + 4, equal
    forwards), every pose back on every rank in pair-id order after each pass; strong scaling is what the line says."""
    r = _run(['--gpus', '2', '--config', 'lomatch', '--total-pairs', '23', '--pairs', '5', '--steps', '2', '--warmup', '1', '--stub-backend', 'gloo'])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['n_gpus'] == 2 and d['pairs_per_step'] == 23 and d['forwards_per_step_rank0'] == 3 and d['scaling'] == 'strong'


def test_lomatch_1781_pairs_on_eight_ranks():
    """The world-8 rehearsal of configs[3] that needs no 8-GPU node: `bench.py --gpus 8 --config lomatch --total-pairs 1781` on gloo with the
    stand-in forward -- 223 / 222-row shards (ragged), at most 64 per forward cut into EQUAL forwards (56 + 56 + 56 + 55 on rank 0), every pose
    on every rank in pair order after each pass (asserted inside run_stub), eight ranks seen through the gather."""
    r = _run(['--gpus', '8', '--config', 'lomatch', '--total-pairs', '1781', '--steps', '2', '--warmup', '1', '--stub-backend', 'gloo'])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['n_gpus'] == 8 and d['pairs_per_step'] == 1781 and d['forwards_per_step_rank0'] == 4 and d['scaling'] == 'strong'
    assert len(d['per_rank_ms_per_step']) == 8


def test_parity_check_logic_on_cpu():
    """bench.parity_check (the "pose err vs ref" field of the bench line) on a stand-in product: the oracle's own batched forward must
    pass with zero error, a 1e-3 shift of one correspondence row must fail the gate, a broken key point must fail bit-exactness."""
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import regtr_ref
    from regtr_amd.synthetic import synth_pair
    from tests.util import load_cfg, seeded_sd
    cfg = load_cfg('3dmatch')
    sd = seeded_sd(cfg)

    class Model:
        def state_dict(self):
            return sd
    pairs = [synth_pair(i, 2500) for i in range(2)]
    with torch.no_grad():
        out = regtr_ref.regtr_forward(sd, cfg, [s for s, _ in pairs], [t for _, t in pairs])
    par = bench.parity_check(cfg, Model(), pairs, out, [0, 1])
    assert par['ok'] and par['keypoints_bit_exact'] and par['corr_max_abs'] < 1e-5 and par['pose_max_abs'] < 1e-4 and par['pairs_checked'] == 2, par
    assert par['kabsch_cond_max'] > 1.0 and par['pose_vs_f64_kabsch_of_own_outputs'] < 1e-4
    bad = dict(out)
    bad['src_kp_warped'] = [c.clone() for c in out['src_kp_warped']]
    bad['src_kp_warped'][1][:, 0] += 1e-3
    r = bench.parity_check(cfg, Model(), pairs, bad, [0, 1])
    assert not r['ok'] and r['reason'] == 'mismatch'
    # a pose beyond 1e-4 with correspondences inside it: never ok -- diagnosed ('conditioning' only if the pose is the float64 Kabsch of
    # the product's own correspondences on an ill-conditioned pair; a shifted pose is not)
    bad = dict(out)
    bad['pose'] = out['pose'].clone(); bad['pose'][:, 0, 0, 3] += 5e-4
    r = bench.parity_check(cfg, Model(), pairs, bad, [0, 1])
    assert not r['ok'] and r['reason'] == 'mismatch' and r['corr_max_abs'] < 1e-5
    assert len(r['per_pair']) == 2 and r['per_pair'][0]['slot'] == 0
    assert bench.parity_slots([5, 9, 1, 7, 3, 3, 8, 2, 6, 4], 8) == sorted(set(bench.parity_slots([5, 9, 1, 7, 3, 3, 8, 2, 6, 4], 8)))
    sl = bench.parity_slots([5, 9, 1, 7, 3, 3, 8, 2, 6, 4], 8)
    assert len(sl) == 8 and {0, 9, 1, 2} <= set(sl)                 # first, last, largest, smallest
    assert bench.parity_slots([5, 9], 8) == [0, 1] and bench.parity_slots([4], 8) == [0]
    bad = dict(out)
    bad['tgt_kp'] = [k.clone() for k in out['tgt_kp']]
    bad['tgt_kp'][0][3, 1] += 1e-6
    r = bench.parity_check(cfg, Model(), pairs, bad, [0, 1])
    assert not r['ok'] and not r['keypoints_bit_exact']


def test_real_pairs_workload_is_deterministic_and_rigid():
    """bench.real_pairs (`bench.py --real`): the three shipped 3DMatch pairs (tests/golden fixtures) replicated under random rigid motions --
    slots 0-2 are the originals bit for bit, every replica is a float32 rigid image of its original (pairwise distances preserved to float32
    rounding), replicas differ from each other, and the workload is a function of the slot ids alone."""
    import sys
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    a, b = bench.real_pairs(7), bench.real_pairs(7)
    for (s1, t1), (s2, t2) in zip(a, b):
        assert np.array_equal(s1, s2) and np.array_equal(t1, t2) and s1.dtype == np.float32
    g = np.load(os.path.join(ROOT, 'tests', 'golden', '3dmatch_kitchen.npz'))
    assert np.array_equal(a[0][0], g['src']) and np.array_equal(a[0][1], g['tgt'])
    assert [len(s) for s, _ in a[:3]] == [len(s) for s, _ in a[3:6]] and not np.array_equal(a[0][0], a[3][0]) and not np.array_equal(a[3][0], a[6][0])
    rng = np.random.default_rng(0)
    i, j = rng.integers(0, len(a[0][0]), 200), rng.integers(0, len(a[0][0]), 200)
    d0 = np.linalg.norm(a[0][0][i].astype(np.float64) - a[0][0][j], axis=1)
    d3 = np.linalg.norm(a[3][0][i].astype(np.float64) - a[3][0][j], axis=1)
    assert np.abs(d0 - d3).max() < 5e-6                      # rigid: distances survive (float32 rounding of ~4 m coordinates)
    assert bench.real_pairs(2, first_id=3)[0][0].tobytes() == a[3][0].tobytes()


def test_settle_phase_has_a_fixed_length_and_reports_its_passes():
    """bench.settle_device: untimed set-up passes for a fixed number of seconds before the warm-up steps (a slow plateau must not end it early),
    nothing when the budget is 0; and the defaults the driver's flag-less run gets."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    r = bench.settle_device(lambda: (calls.append(1), time.sleep(0.01)), 0.15, sync=lambda: None)
    assert r['passes'] == len(calls) >= 5 and 0.15 <= r['seconds'] < 1.0 and len(r['first_ms']) == 3 and all(t >= 9.0 for t in r['last_ms'])
    assert bench.settle_device(lambda: calls.append(1), 0.0, sync=lambda: None)['passes'] == 0
    assert bench.DEFAULT_PAIRS == {'3dmatch': 64, 'modelnet': 128, 'lomatch': 64} and bench.DEFAULT_REPLICAS == {'3dmatch': 3, 'modelnet': 3, 'lomatch': 3} and bench.REDUCED_TOL['pose'] <= 0.1


def test_plan_pairs_equal_forwards_and_rank_emulation():
    """bench.plan_pairs: a lomatch shard is cut into the fewest forwards of at most --pairs pairs, then into EQUAL ones (223 -> 56 + 56 + 56 + 55, not
    64 x 3 + 31); the default configuration is three concurrent 64-pair forwards per rank and step; --emulate-rank-of 8 gives one GPU rank 0's shard of an 8-rank job (the strong-scaling prediction of DESIGN section 7)."""
    import argparse
    import sys
    import torch
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(config='lomatch', pairs=0, total_pairs=1781, emulate_rank_of=8)
    lomatch, per_fwd, ids, chunks, per_step = bench.plan_pairs(a, 0, 1, torch.device('cpu'))
    assert lomatch and per_fwd == 64 and ids.tolist() == list(range(0, 1781, 8)) and [hi - lo for lo, hi in chunks] == [56, 56, 56, 55] and per_step == 223
    a = argparse.Namespace(config='lomatch', pairs=192, total_pairs=1781, emulate_rank_of=8)
    assert bench.plan_pairs(a, 0, 1, torch.device('cpu'))[3] == [(0, 112), (112, 223)]
    a = argparse.Namespace(config='lomatch', pairs=64, total_pairs=1781, emulate_rank_of=0)
    _, _, ids, chunks, per_step = bench.plan_pairs(a, 3, 8, torch.device('cpu'))
    assert ids.tolist() == list(range(3, 1781, 8)) and [hi - lo for lo, hi in chunks] == [56, 56, 56, 55] and per_step == 1781
    a = argparse.Namespace(config='3dmatch', pairs=0, total_pairs=1781, emulate_rank_of=0, replicas=3)
    _, per_fwd, ids, chunks, per_step = bench.plan_pairs(a, 1, 2, torch.device('cpu'))
    assert per_fwd == 64 and ids[0] == 192 and chunks == [(0, 64), (64, 128), (128, 192)] and per_step == 384


def test_replica_runner_orders_chunks_and_propagates_errors():
    """workload.ReplicaRunner on CPU with stand-in models: replica r runs chunks r, r + R, ...; run(n) returns the last pass's outputs in CHUNK
    order; poses() concatenates them in batch order; an exception on a worker thread surfaces on the caller."""
    import sys
    import threading
    import torch
    sys.path.insert(0, ROOT)
    from regtr_amd.workload import ReplicaRunner
    batch = {'src_xyz': [torch.full((2, 3), float(i)) for i in range(7)], 'tgt_xyz': [torch.zeros(2, 3) for _ in range(7)]}
    chunks = [(0, 3), (3, 5), (5, 7)]
    calls = []

    def make(tag):
        def model(b):
            calls.append((tag, threading.get_ident(), len(b['src_xyz'])))
            ids = torch.stack([x[0, 0] for x in b['src_xyz']])
            return {'pose': (torch.eye(3, 4)[None, None] + ids[None, :, None, None]).repeat(6, 1, 1, 1)}
        return model
    r = ReplicaRunner([make('a'), make('b')], batch, chunks, torch.device('cpu'))
    outs = r.run(2)
    assert len(outs) == 3 and [o['pose'].shape[1] for o in outs] == [3, 2, 2]
    assert torch.equal(r.poses(outs)[:, 0, 0], 1 + torch.arange(7.0))
    assert sorted(c[0] + str(c[2]) for c in calls) == ['a2', 'a2', 'a3', 'a3', 'b2', 'b2']       # replica a: chunks 0 and 2, replica b: chunk 1, two passes
    assert len({c[1] for c in calls}) == 2
    one = ReplicaRunner([make('c')], batch, chunks, torch.device('cpu'))
    assert torch.equal(one.poses(one.run(1)), r.poses(outs))

    def broken(b):
        raise ValueError('boom')
    with pytest.raises(ValueError, match='boom'):
        ReplicaRunner([make('a'), broken], batch, chunks, torch.device('cpu')).run(1)


def test_forward_compulsory_bytes_matches_survey_appendix_b():
    """workload.forward_compulsory_bytes (the denominator of the bench line's forward_traffic.ratio) on the red-kitchen pair's level sizes: SURVEY.md
    Appendix B's compulsory column sums to 119 MB for the eleven KPConv blocks; + preprocessing 1.2 MB, tokens 1.5 MB, cross-encoder / head weights 27.4 MB."""
    import sys
    sys.path.insert(0, ROOT)
    from regtr_amd import RegTR, load_config
    from regtr_amd.workload import forward_compulsory_bytes
    cfg = load_config(os.path.join(ROOT, 'regtr_amd', 'conf', '3dmatch.yaml'))
    m = RegTR(cfg)
    total = forward_compulsory_bytes(m, [38061, 10088, 2753, 751], list(cfg.neighborhood_limits), 751)
    blocks = total - (sum(24 * a + 12 * b for a, b in ((38061, 10088), (10088, 2753), (2753, 751))) + 751 * 256 * 8
                      + sum(p.numel() * 4 for n, p in m.named_parameters() if not n.startswith('kpf_encoder')))
    assert abs(blocks / 1e6 - 119.0) < 1.5, blocks / 1e6
    assert 145e6 < total < 153e6
