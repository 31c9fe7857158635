"""Frame sharding over GPUs (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI).

The path shards by frames: A1 -> A2 -> A3a of frame i depend only on frame i, so
rank r owns the contiguous block of frames [r*F/G, (r+1)*F/G) and no collective touches the per-frame
work.  The single exchange is for the combined merge (pose.cpp:530), which averages per-frame voxel
points from many frames inside one XY cell: every rank needs the per-frame voxel clouds of all
ranks, in global frame order, so that the merge sums the same points in the same order as a
single-GPU run (bit-identical result).  Variable shard sizes: counts first, then one padded
all_gather_into_tensor.
"""
import torch
import torch.distributed as dist


def shard_range(n_frames_total, rank, world):
    """contiguous block of frames owned by `rank` (keeps frame order and fly-over locality)"""
    base, rem = divmod(n_frames_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_points(local, group=None):
    """local: [n,4] int32 tensor (16-byte points) on this rank's device (or CPU with gloo).
    Returns (list of per-rank tensors in rank order, counts list)."""
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = [int(c) for c in counts.tolist()]
    m = max(max(counts), 1)
    padded = torch.empty((m, 4), dtype=torch.int32, device=local.device)
    padded[: local.shape[0]] = local
    gathered = torch.empty((world * m, 4), dtype=torch.int32, device=local.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    return [gathered[r * m: r * m + counts[r]] for r in range(world)], counts


def exchange_cloud_big(ctx, device, group=None):
    """After every rank accumulated its own frames: rebuild cloud_big on every rank as the
    concatenation of all ranks' per-frame voxel clouds in rank (= global frame) order.
    Returns the total number of points."""
    local = ctx.cloudBigRead(device=device)
    shards, counts = all_gather_points(local, group)
    ctx.cloudBigReset()
    for s in shards:
        if s.shape[0]:
            ctx.cloudBigAppend(s)
    return sum(counts)
