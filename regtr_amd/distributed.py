"""Pair-level data parallelism: one process per GPU, independent scan pairs sharded across ranks, ONE collective --
the gather of the per-pair poses (RCCL over xGMI on MI355X; backend "nccl" is RCCL on ROCm, gloo on CPU in tests).

The reference has no distributed code at all (single process, one pair per step: trainer.py:177-211, conf
test_batch_size: 1); this is the shard point its test loop offers.  No collective runs inside the forward.
"""
import torch
import torch.distributed as dist


def shard_pairs(n_pairs, rank, world):
    """Pair i -> rank i % world (round robin keeps per-rank work balanced when pairs are sorted by size)."""
    return list(range(rank, n_pairs, world))


def gather_poses(poses, pair_ids):
    """poses (n_local, 12) f32 and pair_ids (n_local,) i32 of this rank -> every rank gets (n_total, 12), (n_total,)
    ordered by pair id.  Ranks may hold different counts (ragged shards): counts are exchanged first, buffers padded to
    the maximum so a single all_gather per tensor suffices."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        order = torch.argsort(pair_ids)
        return poses[order], pair_ids[order]
    world = dist.get_world_size()
    dev = poses.device
    n_local = torch.tensor([poses.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    buf = torch.zeros((n_max, 13), dtype=torch.float32, device=dev)
    buf[:poses.shape[0], :12] = poses
    buf[:poses.shape[0], 12] = pair_ids.to(torch.float32)        # ids < 2^24 are exact in f32
    out = torch.empty((world * n_max, 13), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, buf)
    rows = torch.cat([out[r * n_max:r * n_max + counts[r]] for r in range(world)])
    ids = rows[:, 12].to(torch.int32)
    order = torch.argsort(ids)
    return rows[order, :12].contiguous(), ids[order]
