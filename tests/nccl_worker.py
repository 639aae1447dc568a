"""Worker of tests/test_gpu_nccl.py: one rank of an RCCL ("nccl") process group whose ranks ALL use cuda:0 -- exercises
dist.init_process_group('nccl', device_id=...), the device-tensor all_gather_into_tensor of regtr_amd.distributed.gather_poses and a
barrier.  Exit code 77 = RCCL refuses several ranks on one GPU (the caller skips, loudly); 0 = every rank got every pose."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    n_pairs = int(sys.argv[1])
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group('nccl', device_id=dev)
        from regtr_amd.distributed import gather_poses, shard_pairs
        mine = shard_pairs(n_pairs, rank, world)
        poses = (torch.arange(12, dtype=torch.float32, device=dev)[None] + 100.0 * torch.tensor(mine, dtype=torch.float32, device=dev)[:, None])
        ids = torch.tensor(mine, dtype=torch.int32, device=dev)
        all_poses, all_ids = gather_poses(poses.reshape(-1, 12), ids, n_pairs)
        torch.cuda.synchronize()
    except Exception as e:      # noqa: BLE001
        msg = repr(e)
        print(f'rank {rank}: {msg}', flush=True)
        if world > 1 and any(k in msg for k in ('Duplicate GPU', 'duplicate', 'invalid usage', 'ncclInvalidUsage', 'unhandled')):
            sys.exit(77)
        raise
    assert all_ids.tolist() == list(range(n_pairs)), all_ids
    exp = torch.arange(12, dtype=torch.float32, device=dev)[None] + 100.0 * torch.arange(n_pairs, dtype=torch.float32, device=dev)[:, None]
    assert torch.equal(all_poses, exp)
    dist.barrier()
    dist.destroy_process_group()
    print(f'rank {rank}/{world}: gathered {n_pairs} poses over RCCL on {torch.cuda.get_device_name(0)}', flush=True)


if __name__ == '__main__':
    main()
