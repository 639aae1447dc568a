"""Pair-level data parallelism: one process per GPU, independent scan pairs sharded across ranks, ONE collective --
the gather of the per-pair poses (RCCL over xGMI on MI355X; backend "nccl" is RCCL on ROCm, gloo on CPU in tests): ONE
all_gather_into_tensor of (pose | pair id) rows padded to the common shard capacity.

The reference has no distributed code at all (single process, one pair per step: trainer.py:177-211, conf
test_batch_size: 1); this is the shard point its test loop offers.  No collective runs inside the forward.
"""
import torch
import torch.distributed as dist


def shard_pairs(n_pairs, rank, world):
    """Pair i -> rank i % world (round robin keeps per-rank work balanced when pairs are sorted by size)."""
    return list(range(rank, n_pairs, world))


def shard_capacity(n_pairs, world):
    """Largest shard of shard_pairs(n_pairs, ., world): what every rank pads its payload to."""
    return (n_pairs + world - 1) // world


def gather_poses(poses, pair_ids, n_total=None):
    """poses (n_local, 12) f32 and pair_ids (n_local,) i32 of this rank -> every rank gets (n_total, 12), (n_total,) ordered by pair id.
    ONE collective: every rank pads its (pose | id) rows to the common capacity ceil(n_total / world) -- known on every rank without
    talking, since the shards are i % world -- with id = -1 marking padding, and a single all_gather_into_tensor moves the payload; no
    count exchange, no host read-back before the collective.
    n_total: pairs in the whole set (all ranks).  REQUIRED once a process group exists: the capacity must be the same number on every
    rank BEFORE the collective, and it cannot be derived from a rank's own row count when shards are ragged (mismatched all_gather sizes
    hang or corrupt).  A rank that holds more rows than the capacity (a sharding bug) does not raise on its own -- the others would
    block in the collective for ever: it sends id -2 in its first row, the collective completes, and EVERY rank raises."""
    if not (dist.is_available() and dist.is_initialized()):
        order = torch.argsort(pair_ids)
        return poses[order], pair_ids[order]
    world = dist.get_world_size()           # (a one-rank group still runs the collective: same code path at every world size)
    if n_total is None:
        raise TypeError('gather_poses: n_total (pairs in the whole set) is required when a process group is initialised')
    dev = poses.device
    n_local = poses.shape[0]
    cap = shard_capacity(int(n_total), world)
    overflow = n_local > cap
    n_send = min(n_local, cap)
    buf = torch.zeros((max(cap, 1), 13), dtype=torch.float32, device=dev)      # (>= 1 row: the overflow marker needs one)
    buf[:, 12] = -1.0                                         # padding rows carry id -1
    buf[:n_send, :12] = poses[:n_send]
    buf[:n_send, 12] = pair_ids[:n_send].to(torch.float32)    # ids < 2^24 are exact in f32
    if overflow:
        buf[0, 12] = -2.0
    out = torch.empty((world * buf.shape[0], 13), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, buf)
    if bool((out[:, 12] == -2.0).any()):
        bad = torch.nonzero(out.view(world, -1, 13)[:, 0, 12] == -2.0).flatten().tolist()
        raise ValueError(f'gather_poses: rank(s) {bad} hold more rows than the shard capacity {cap} of a {n_total}-pair set on {world} ranks '
                         '(pairs must be sharded i % world: regtr_amd.distributed.shard_pairs)')
    rows = out[out[:, 12] >= 0]
    ids = rows[:, 12].to(torch.int32)
    order = torch.argsort(ids)
    return rows[order, :12].contiguous(), ids[order]
