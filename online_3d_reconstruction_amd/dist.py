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


# what the last merge_partitioned of this process did: collectives issued and host round trips (stream drains / blocking
# read-backs) before the final gather — the figures DESIGN.md section 5 quotes; tests assert them
last_stats = {}


def _header_tensor(ctx, cdev, dev_path):
    """this rank's 32-byte header {min xyz, max xyz (f32), count (i64)} as a uint8 tensor on the collective's device"""
    import numpy as np
    if dev_path:
        return ctx.cloudBigHeaderDev()  # stays in HBM: no round trip
    mn, mx, n_local = ctx.cloudBigBBox()
    raw = np.concatenate([np.asarray(mn, np.float32), np.asarray(mx, np.float32)]).tobytes() + np.int64(n_local).tobytes()
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(cdev)


def _parse_headers(raw, world):
    """[world*32] uint8 (host) -> (gmin, gmax, per-rank counts)"""
    import numpy as np
    b = raw.reshape(world, 32)
    boxes = np.ascontiguousarray(b[:, :24]).view(np.float32).reshape(world, 6)
    counts = np.ascontiguousarray(b[:, 24:]).view(np.int64).reshape(world)
    live = counts > 0
    if not live.any():
        return np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32), counts
    return boxes[live, 0:3].min(axis=0).astype(np.float32), boxes[live, 3:6].max(axis=0).astype(np.float32), counts


class ExchangeError(RuntimeError):
    """merge_partitioned failed on some rank; EVERY rank raises at the same point of the protocol (`rank` = the first
    rank that reported a failure, `code` = its negative error code, `own` = it was this rank)."""

    def __init__(self, rank, code, own, text=""):
        super().__init__(f"merge_partitioned: rank {rank} failed with code {code}; every rank left the exchange together"
                         + (f" ({text})" if text else ""))
        self.rank, self.code, self.own = rank, code, own


_I64_MAX = (1 << 63) - 1
_ROW_EXTRA = 5  # status | this rank's error code | receive-buffer capacity | (merge, gather buffers: torch's here)


def _err_code(e):
    c = getattr(e, "code", None)
    return int(c) if isinstance(c, int) and c < 0 else -3


def merge_partitioned(ctx, device, group=None, gather_result=True, comm_device=None):
    """The reference's final merge (pose.cpp:530) over frames sharded across ranks, without replicating it:

      1. all-gather of the ranks' 32-byte headers (bounding box + size of cloud_big) -> the box PCL would see;
      2. every rank stably reorders its cloud by index slice of the combined grid over that box
         (slice r = the r-th of `world` equal ranges of the linear voxel index);
      3. all-gather of the slice-count vectors, ONE host read-back (headers + count matrix), then one all-to-all
         (RCCL over xGMI: every peer pair uses its own link) moves slice r to rank r; segments arrive in
         source-rank order = global frame order;
      4. every rank merges its slice with the grid over the global box (second and last host round trip: the
         merged slice's size);
      5. all-gather of the (small) merged slices; rank order = ascending voxel index.

    With a GPU context (cloudBigHeaderDev / cloudBigPartitionDev) steps 1-3 keep their data in HBM and every call is
    stream-ordered: the host waits once for the count matrix and once for the merged size.  Contexts without those
    methods (CPU stand-ins in the gloo tests) and rehearsals with `comm_device` (collectives on host copies) run the
    same protocol on host values.

    FAILURE IS COLLECTIVE (the same wire format as o3dr_merge_partitioned, include/o3dr.h): a rank whose own step raises
    keeps taking part in the collectives that remain, with its error code in place of its data - a negative count in its
    header, a word next to its slice counts, a negative merged size - and every rank raises ExchangeError at the same point.
    The receive buffer's capacity travels with the slice counts: only when some rank has to grow it (known to all) does
    one more 8-byte all-gather carry the outcome of that allocation before the all-to-all.

    Returns (merged [M,4] int32 tensor or this rank's slice if gather_result is False, total points merged).
    Bit-identical to a single-GPU run over all frames.  `last_stats` holds what this call did and moved."""
    import numpy as np
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    cdev = comm_device if comm_device is not None else device
    on_gpu = torch.device(device).type == "cuda"
    dev_path = comm_device is None and hasattr(ctx, "cloudBigPartitionDev")
    stats = {"collectives": 0, "host_syncs_before_final_gather": 0, "device_resident": bool(dev_path), "agreement_rounds": 0}
    RW = world + _ROW_EXTRA
    local = {"code": 0, "text": ""}

    def note(e):
        if local["code"] == 0:
            local["code"], local["text"] = _err_code(e), repr(e)

    def leave(first_rank, code):
        last_stats.clear()
        last_stats.update(stats)
        raise ExchangeError(first_rank, int(code), local["code"] != 0, local["text"])

    # Stream order: the library works on the context's stream, torch's collectives are ordered against torch's current
    # stream.  When the two are the same stream (bench.py hands the context torch's), nothing else is needed; else the
    # host drains one for the other (counted below).
    same_stream = (not on_gpu) or (getattr(ctx, "stream_raw", 0) != 0 and
                                   ctx.stream_raw == torch.cuda.current_stream(device).cuda_stream)

    def lib_to_torch():  # library work issued so far must precede the next collective
        if dev_path and not same_stream:
            ctx.synchronize()
            stats["host_syncs_before_final_gather"] += 1

    def torch_to_lib():  # ... and the collective must precede the library's next work
        if on_gpu and comm_device is None and not same_stream:
            torch.cuda.current_stream(device).synchronize()
            stats["host_syncs_before_final_gather"] += 1

    # 1. headers
    try:
        mine_hdr = _header_tensor(ctx, cdev, dev_path)
    except Exception as e:  # noqa: BLE001 - whatever it is, the peers must not be left inside the all-gather
        note(e)
        raw = np.concatenate([np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32)]).tobytes() + np.int64(local["code"]).tobytes()
        mine_hdr = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(cdev)
    if not dev_path:
        stats["host_syncs_before_final_gather"] += 1  # (host values: the box and count were read back)
    lib_to_torch()
    hdrs = torch.empty(world * 32, dtype=torch.uint8, device=cdev)
    dist.all_gather_into_tensor(hdrs, mine_hdr, group=group)
    stats["collectives"] += 1
    # 2. partition (+ status word, this rank's error code, its receive buffer's capacity)
    row_head = None
    # two_phase: slice sizes first (nothing moves), the slices placed in ONE pass once the counts are known, with room for
    # what arrives in front of and behind the rank's own slice, which is then never sent to itself (include/o3dr.h)
    two_phase = dev_path and comm_device is None and hasattr(ctx, "cloudBigPlaceSlices") and hasattr(ctx, "cloudBigRawView")
    stats["two_phase"] = bool(two_phase)
    if dev_path:
        torch_to_lib()
        if local["code"] == 0:
            try:
                if two_phase:
                    row_head = ctx.cloudBigSliceCountsDev(hdrs, world)  # int64 [world + 1] in HBM, asynchronous, nothing moved
                else:
                    row_head = ctx.cloudBigPartitionDev(hdrs, world)    # ... the cloud reordered by slice
            except Exception as e:  # noqa: BLE001
                note(e)
        lib_to_torch()
    else:
        gmin, gmax, hcounts = _parse_headers(hdrs.cpu().numpy(), world)
        if (hcounts < 0).any():  # (host values: every rank sees the failed header here)
            bad = int(np.nonzero(hcounts < 0)[0][0])
            leave(bad, hcounts[bad])
        if int(hcounts.sum()) == 0:
            last_stats.clear()
            last_stats.update(stats)
            return torch.empty((0, 4), dtype=torch.int32, device=device), 0
        try:
            counts, status = ctx.cloudBigPartition(gmin, gmax, world)
            row_head = torch.tensor(list(counts) + [int(status)], dtype=torch.int64, device=cdev)
        except Exception as e:  # noqa: BLE001
            note(e)
        stats["host_syncs_before_final_gather"] += 1
    if row_head is None:
        row_head = torch.zeros(world + 1, dtype=torch.int64, device=cdev)
    zero_copy = comm_device is None and hasattr(ctx, "cloudBigView")
    recv_cap = ctx.cloudBigCapacity()[1] if (zero_copy and hasattr(ctx, "cloudBigCapacity")) else (0 if zero_copy else _I64_MAX)
    row = torch.cat([row_head, torch.tensor([local["code"], recv_cap, _I64_MAX, _I64_MAX], dtype=torch.int64).to(cdev)])
    # 3. row matrix, the one read-back, the all-to-all
    matrix = torch.empty(world * RW, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(matrix, row, group=group)   # row s = what rank s sends to everybody (+ its status words)
    stats["collectives"] += 1
    both = torch.cat([hdrs.view(torch.int64), matrix]).cpu().numpy()  # headers + matrix: ONE blocking copy
    hdrs_h = both[: world * 4].view(np.uint8)
    matrix_h = both[world * 4:].reshape(world, RW)
    if dev_path:
        stats["host_syncs_before_final_gather"] += 1
    gmin, gmax, hcounts = _parse_headers(hdrs_h, world)
    for r in range(world):  # a failure anywhere so far: every rank sees it here and leaves before the all-to-all
        code = int(hcounts[r]) if hcounts[r] < 0 else int(matrix_h[r, world + 1])
        if code < 0:
            leave(r, code)
    total = int(hcounts.sum())
    n_local = int(hcounts[rank])
    if total == 0:
        last_stats.clear()
        last_stats.update(stats)
        return torch.empty((0, 4), dtype=torch.int32, device=device), 0
    overflow = bool((matrix_h[:, world] & 1).any())         # (the same global box on every rank: all agree)
    if overflow:  # PCL's overflow guard on the global box: the merge returns its input unchanged
        sends = np.diag(hcounts.astype(np.int64))           # everything stays where it is; rank order is already global order
    else:
        sends = matrix_h[:, :world].astype(np.int64)        # sends[s, r] = points rank s sends to rank r
    counts = [int(v) for v in sends[rank]]
    recv_counts = [int(v) for v in sends[:, rank]]
    n_recv = sum(recv_counts)
    stats.update({"points_local": n_local, "points_sent_off_rank": n_local - counts[rank],
                  "points_received_off_rank": n_recv - recv_counts[rank], "bytes_sent": 16 * (n_local - counts[rank]),
                  "bytes_received": 16 * (n_recv - recv_counts[rank]), "points_into_merge": n_recv, "points_all_ranks": total})
    # does any rank have to grow its receive buffer?  (every rank evaluates every rank: no disagreement)
    n_off = n_local - counts[rank]
    if two_phase:  # the buffer holds what arrives AND what leaves; a rank that neither sends nor receives moves nothing
        off_all = hcounts.astype(np.int64) - np.diag(sends)
        need_all = np.where((off_all == 0) & (sends.sum(axis=0) == np.diag(sends)), 0, sends.sum(axis=0) + off_all)
    else:
        need_all = sends.sum(axis=0)
    need_mine = int(need_all[rank])
    any_grows = bool((need_all > matrix_h[:, world + 2]).any()) and not overflow
    recv = None
    if any_grows:
        if zero_copy:
            try:
                recv = ctx.cloudBigRecvBuffer(need_mine)
            except Exception as e:  # noqa: BLE001
                note(e)
        ok = torch.empty(world, dtype=torch.int64, device=cdev)
        lib_to_torch()
        dist.all_gather_into_tensor(ok, torch.tensor([local["code"]], dtype=torch.int64).to(cdev), group=group)
        stats["collectives"] += 1
        stats["agreement_rounds"] = 1
        ok_h = ok.cpu().numpy()
        stats["host_syncs_before_final_gather"] += 1
        for r in range(world):
            if ok_h[r] < 0:
                leave(r, ok_h[r])
    if two_phase and not overflow:
        # the slices placed in one pass, with gaps for what arrives; two all-to-alls (to the higher ranks / to the lower ranks:
        # each receives into ONE contiguous gap, which all_to_all_single needs) move only what changes rank
        if hasattr(ctx, "cloudBigAssumeSize"):
            ctx.cloudBigAssumeSize(n_local)
        n_before, n_after, own = sum(recv_counts[:rank]), sum(recv_counts[rank + 1:]), counts[rank]
        moves = need_mine != 0
        empty = torch.empty((0, 4), dtype=torch.int32, device=device)
        lo_out = hi_out = to_lo = to_hi = empty
        if moves:
            send_start = ctx.cloudBigPlaceSlices(rank, counts, n_before, n_after)
            raw = ctx.cloudBigRawView()
            n_lo = sum(counts[:rank])
            lo_out, hi_out = raw[:n_before], raw[n_before + own: n_before + own + n_after]
            to_lo, to_hi = raw[send_start: send_start + n_lo], raw[send_start + n_lo: send_start + n_off]
        lib_to_torch()
        zeros_lo, zeros_hi = [0] * (rank + 1), [0] * (world - rank)
        dist.all_to_all_single(lo_out, to_hi, output_split_sizes=recv_counts[:rank] + zeros_hi,
                               input_split_sizes=zeros_lo + counts[rank + 1:], group=group)
        dist.all_to_all_single(hi_out, to_lo, output_split_sizes=zeros_lo + recv_counts[rank + 1:],
                               input_split_sizes=counts[:rank] + zeros_hi, group=group)
        stats["collectives"] += 2
        torch_to_lib()
        if moves:
            ctx.cloudBigSetSize(n_recv)                     # stream-ordered
    elif zero_copy:
        # send straight out of cloud_big, receive straight into the library's second cloud buffer
        if hasattr(ctx, "cloudBigAssumeSize"):
            ctx.cloudBigAssumeSize(n_local)                 # (its own header told the host: no round trip for the view)
        send = ctx.cloudBigView()
        if recv is None:
            recv = ctx.cloudBigRecvBuffer(n_recv)           # (fits: every rank checked every rank's capacity)
        lib_to_torch()
        dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=list(counts), group=group)
        stats["collectives"] += 1
        torch_to_lib()
        ctx.cloudBigAdopt(n_recv)                           # stream-ordered
    else:
        send = ctx.cloudBigRead(device=device).to(cdev)
        recv = torch.empty((n_recv, 4), dtype=torch.int32, device=cdev)
        dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=list(counts), group=group)
        stats["collectives"] += 1
        ctx.cloudBigReset()
        if recv.shape[0]:
            ctx.cloudBigAppend(recv.to(device))
    # 4. local merge of the slice (the round trip for its size is inside).  A failure here travels with the merged sizes
    #    of the final gather, so that no rank waits in a collective the failed one never enters.
    mine = None
    try:
        if dev_path:
            mine = ctx.finalize(device=device, gmin=gmin, gmax=gmax, n_hint=n_recv)
        else:
            mine = ctx.finalize(device=device, gmin=gmin, gmax=gmax)
    except Exception as e:  # noqa: BLE001
        note(e)
    stats["host_syncs_before_final_gather"] += 1
    last_stats.clear()
    last_stats.update(stats)
    if not gather_result:  # (no collective follows: a local failure is this rank's alone)
        if local["code"]:
            leave(rank, local["code"])
        return mine, total
    # 5. final gather: sizes (or error codes), then the slices padded to the largest
    lib_to_torch()
    n_mine = local["code"] if local["code"] else int(mine.shape[0])
    sizes = torch.empty(world, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(sizes, torch.tensor([n_mine], dtype=torch.int64).to(cdev), group=group)
    last_stats["collectives"] += 1
    sizes_h = [int(v) for v in sizes.tolist()]
    for r in range(world):
        if sizes_h[r] < 0:
            leave(r, sizes_h[r])
    m = max(max(sizes_h), 1)
    padded = torch.empty((m, 4), dtype=torch.int32, device=cdev)
    padded[: mine.shape[0]] = mine.to(cdev)
    gathered = torch.empty((world * m, 4), dtype=torch.int32, device=cdev)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    last_stats["collectives"] += 1
    return torch.cat([gathered[r * m: r * m + sizes_h[r]] for r in range(world)]).to(device), total
