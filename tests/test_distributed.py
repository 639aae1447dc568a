"""CPU, gloo, world_size 2: the N > 1 path of bench.py / test.py -- pair sharding and the single pose gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_pairs, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from regtr_amd.distributed import gather_poses, shard_pairs
    mine = shard_pairs(n_pairs, rank, world)
    # stand-in for the per-pair forward: a pose that is a known function of the pair id
    poses = torch.stack([torch.arange(12, dtype=torch.float32) + 100.0 * i for i in mine]) if mine else torch.zeros(0, 12)
    ids = torch.tensor(mine, dtype=torch.int32)
    calls = []
    for name in ('all_gather', 'all_gather_into_tensor', 'all_reduce', 'broadcast', 'gather', 'all_to_all'):
        def counted(*a, _f=getattr(dist, name), _n=name, **k):
            calls.append(_n)
            return _f(*a, **k)
        setattr(dist, name, counted)
    all_poses, all_ids = gather_poses(poses, ids, n_pairs)      # ONE all_gather_into_tensor: ragged shards padded to ceil(n / world), ids carry validity
    assert calls == ['all_gather_into_tensor'], calls                      # ONE collective, no count exchange
    q.put((rank, all_poses.numpy().copy(), all_ids.numpy().copy()))      # by value: torch tensors travel as fds that die with the worker
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_pairs', [7, 8, 1, 0])
def test_pose_gather_world2(n_pairs):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs: p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = torch.stack([torch.arange(12, dtype=torch.float32) + 100.0 * i for i in range(n_pairs)]) if n_pairs else torch.zeros(0, 12)
    for _, poses, ids in results:
        assert ids.tolist() == list(range(n_pairs))
        assert torch.equal(torch.from_numpy(poses), exp)


def _overflow_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from regtr_amd.distributed import gather_poses
    n_local = 4 if rank == 1 else 2                  # a sharding bug on rank 1: 4 rows where a 4-pair set on 2 ranks allows 2
    res = []
    try:
        gather_poses(torch.zeros(n_local, 12), torch.arange(n_local, dtype=torch.int32), 4)
    except ValueError as e:
        res.append(str(e))
    try:
        gather_poses(torch.zeros(2, 12), torch.arange(2, dtype=torch.int32))           # n_total is required under a process group
    except TypeError as e:
        res.append('TypeError')
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_pose_gather_overflow_raises_on_every_rank():
    """A rank holding more rows than the shard capacity must not raise alone BEFORE the collective (the other ranks would wait in the
    all_gather for ever): the collective completes and every rank raises; and a missing n_total is an error, not a guess."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert len(results[r]) == 2 and 'rank(s) [1]' in results[r][0] and results[r][1] == 'TypeError', results


def test_shard_pairs_partition():
    from regtr_amd.distributed import shard_pairs
    for n, w in [(1781, 8), (5, 8), (0, 2)]:
        parts = [shard_pairs(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


class _StubModel(torch.nn.Module):
    """Stands in for RegTR in the harness test: a 'pose' that is a known function of the pair's clouds."""

    def forward(self, batch):
        B = len(batch['src_xyz'])
        pose = torch.zeros(6, B, 3, 4)
        for b in range(B):
            pose[-1, b, :, 3] = batch['src_xyz'][b].mean(0)
            pose[-1, b, :, :3] = torch.eye(3) * float(len(batch['tgt_xyz'][b]))
        return {'pose': pose}


def _harness_worker(rank, world, port, n_pairs, batch, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from regtr_amd import harness
    pairs = harness.SyntheticPairs(n_pairs, points=1500)
    poses, ids, timing = harness.run_test(_StubModel(), pairs, batch, torch.device('cpu'))
    q.put((rank, poses.copy(), ids.copy(), timing['world']))
    dist.barrier()
    dist.destroy_process_group()


def test_harness_run_test_world2():
    """test.py's loop on 2 ranks (gloo): ragged shards (5 pairs, batches of 2), every rank ends with all poses in pair order."""
    import numpy as np
    from regtr_amd import harness
    n_pairs = 5
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_harness_worker, args=(r, 2, port, n_pairs, 2, q)) for r in range(2)]
    for p in procs: p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pairs = harness.SyntheticPairs(n_pairs, points=1500)
    for _, poses, ids, world in results:
        assert world == 2 and ids.tolist() == list(range(n_pairs)) and poses.shape == (n_pairs, 3, 4)
        for i in range(n_pairs):
            it = pairs[i]
            assert np.allclose(poses[i][:, 3], it['src_xyz'].mean(0), atol=1e-5)
            assert poses[i][0, 0] == float(len(it['tgt_xyz']))


def _harness_worker8(rank, world, port, n_pairs, batch, root, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from regtr_amd import harness
    pairs = harness.SyntheticPairs(n_pairs, points=300)
    calls = []
    for name in ('all_gather', 'all_gather_into_tensor', 'all_reduce', 'broadcast', 'gather', 'all_to_all'):
        def counted(*a, _f=getattr(dist, name), _n=name, **k):
            calls.append(_n)
            return _f(*a, **k)
        setattr(dist, name, counted)
    poses, ids, timing = harness.run_test(_StubModel(), pairs, batch, torch.device('cpu'))
    assert calls == ['all_gather_into_tensor'], calls          # ONE collective for the whole set
    if rank == 0:
        recs = [{'src_path': pairs[int(i)]['src_path'], 'tgt_path': pairs[int(i)]['tgt_path'], 'pose': p} for p, i in zip(poses, ids)]
        harness.write_est_log(root, '3DLoMatch', recs)
    q.put((rank, len(harness.shard_pairs(n_pairs, rank, world)), ids.tolist() == list(range(n_pairs))))
    dist.barrier()
    dist.destroy_process_group()


def test_harness_run_test_world8_ragged_1781_equals_world1(tmp_path):
    """The configs[3] rehearsal that needs no 8-GPU node: the 1781-pair 3DLoMatch-size set through test.py's loop (harness.run_test) on
    EIGHT gloo ranks -- ragged shards of 223 / 222 rows (1781 = 5 x 223 + 3 x 222), ONE collective -- writes the same est.log files, byte for
    byte, as the one-process run."""
    import numpy as np
    from regtr_amd import harness
    n_pairs, world = 1781, 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_harness_worker8, args=(r, world, port, n_pairs, 64, str(tmp_path / 'w8'), q)) for r in range(world)]
    for p in procs: p.start()
    results = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in results] == [223] * 5 + [222] * 3 and all(r[2] for r in results)
    pairs = harness.SyntheticPairs(n_pairs, points=300)
    poses, ids, _ = harness.run_test(_StubModel(), pairs, 64, torch.device('cpu'))
    recs = [{'src_path': pairs[int(i)]['src_path'], 'tgt_path': pairs[int(i)]['tgt_path'], 'pose': p} for p, i in zip(poses, ids)]
    harness.write_est_log(str(tmp_path / 'w1'), '3DLoMatch', recs)
    files = sorted(os.path.relpath(os.path.join(d, f), str(tmp_path / 'w1')) for d, _, fs in os.walk(str(tmp_path / 'w1')) for f in fs)
    assert files and all(f.endswith('est.log') for f in files)
    for f in files:
        assert open(os.path.join(str(tmp_path / 'w1'), f), 'rb').read() == open(os.path.join(str(tmp_path / 'w8'), f), 'rb').read(), f
