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


def _all_to_all_points(send, send_counts, group=None):
    """send: [n,4] int32, grouped by destination rank (send_counts[r] points for rank r).
    Returns the received points, segments in source-rank order, and the receive counts."""
    world = dist.get_world_size(group)
    sc = torch.tensor(send_counts, dtype=torch.int64, device=send.device)
    rc = torch.empty(world, dtype=torch.int64, device=send.device)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(v) for v in rc.tolist()]
    recv = torch.empty((sum(recv_counts), 4), dtype=torch.int32, device=send.device)
    dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=list(send_counts), group=group)
    return recv, recv_counts


def merge_partitioned(ctx, device, group=None, gather_result=True, comm_device=None):
    """The reference's final merge (pose.cpp:530) over frames sharded across ranks, without replicating it:

      1. all-reduce (min/max) of the ranks' cloud_big bounding boxes -> the box PCL would see;
      2. every rank stably reorders its cloud by index slice of the combined grid over that box
         (slice r = the r-th of `world` equal ranges of the linear voxel index);
      3. one all-to-all (RCCL over xGMI: every peer pair uses its own link) moves slice r to rank r;
         segments arrive in source-rank order = global frame order;
      4. every rank merges its slice with the grid over the global box;
      5. all-gather of the (small) merged slices; rank order = ascending voxel index.

    Returns (merged [M,4] int32 tensor or this rank's slice if gather_result is False, total points merged).
    Bit-identical to a single-GPU run over all frames.  `comm_device` (rehearsal with the gloo backend
    on one GPU): run the collectives on copies on that device instead of `device`."""
    import numpy as np
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    cdev = comm_device if comm_device is not None else device
    mn, mx, n_local = ctx.cloudBigBBox()
    # one small all-gather carries every rank's box and count (instead of two all-reduces): 6 floats + the count
    # split into two 24-bit halves, exact in fp32
    mine_hdr = torch.tensor(np.concatenate([mn, mx, [float(n_local & 0xFFFFFF), float(n_local >> 24)]]), dtype=torch.float32,
                            device=cdev)
    hdr = torch.empty(world * 8, dtype=torch.float32, device=cdev)  # (flat: gloo wants the concatenated shape)
    dist.all_gather_into_tensor(hdr, mine_hdr, group=group)
    hdr = hdr.cpu().numpy().reshape(world, 8)
    gmin = hdr[:, 0:3].min(axis=0).astype(np.float32)  # empty ranks contribute (+inf, -inf)
    gmax = hdr[:, 3:6].max(axis=0).astype(np.float32)
    total = int(sum(int(h[6]) + (int(h[7]) << 24) for h in hdr))
    if total == 0:
        return torch.empty((0, 4), dtype=torch.int32, device=device), 0
    counts, status = ctx.cloudBigPartition(gmin, gmax, world)
    if status & 1:  # PCL's overflow guard on the global box: the merge returns its input unchanged
        counts = [0] * world
        counts[rank] = n_local  # everything stays where it is; rank order is already global order
    # Streams: cloudBigPartition returns after its own stream has drained (it reads the counts back), so the
    # collective below may run on any stream; it is drained in turn before the library adopts what it received.
    zero_copy = comm_device is None and hasattr(ctx, "cloudBigView")
    if zero_copy:
        # send straight out of cloud_big, receive straight into the library's second cloud buffer
        send = ctx.cloudBigView()
        sc = torch.tensor(counts, dtype=torch.int64, device=device)
        allc = torch.empty(world * world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(allc, sc, group=group)   # row s = what rank s sends to everybody
        recv_counts = [int(v) for v in allc.view(world, world)[:, rank].tolist()]
        n_recv = sum(recv_counts)
        recv = ctx.cloudBigRecvBuffer(n_recv)
        dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=list(counts), group=group)
        if torch.device(device).type == "cuda":
            torch.cuda.current_stream(device).synchronize()
        ctx.cloudBigAdopt(n_recv)
    else:
        send = ctx.cloudBigRead(device=device).to(cdev)
        recv, _ = _all_to_all_points(send, counts, group)
        ctx.cloudBigReset()
        if recv.shape[0]:
            ctx.cloudBigAppend(recv.to(device))
    mine = ctx.finalize(device=device, gmin=gmin, gmax=gmax)
    if not gather_result:
        return mine, total
    shards, _ = all_gather_points(mine.to(cdev), group)
    return torch.cat(shards).to(device), total
