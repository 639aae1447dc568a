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
    all_poses, all_ids = gather_poses(poses, ids)
    q.put((rank, all_poses.clone(), all_ids.clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_pairs', [7, 8, 1])
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
    exp = torch.stack([torch.arange(12, dtype=torch.float32) + 100.0 * i for i in range(n_pairs)])
    for _, poses, ids in results:
        assert ids.tolist() == list(range(n_pairs))
        assert torch.equal(poses, exp)


def test_shard_pairs_partition():
    from regtr_amd.distributed import shard_pairs
    for n, w in [(1781, 8), (5, 8), (0, 2)]:
        parts = [shard_pairs(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
