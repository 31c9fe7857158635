"""CPU tests of the N>1 path: world_size-2 gloo ranks shard frames, exchange their per-frame voxel
clouds with the same helper the GPU path uses, and the merged result equals the single-process one."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from online_3d_reconstruction_amd import dist as o3dist
    from online_3d_reconstruction_amd import synth
    from oracle import orc
    F_total, rows, cols, jump, vs = 5, 240, 400, 2, 0.05
    Q = synth.camera_Q(rows, cols)
    a, b = o3dist.shard_range(F_total, rank, world)
    clouds = []
    for i in range(a, b):  # per-frame work of this rank's shard (oracle stands in for the GPU here)
        d, c = synth.make_frame(i, rows, cols)
        clouds.append(orc.create_and_transform_pt_cloud(d, c, Q, synth.make_pose(i), vs, jump_pixels=jump)[0])
    local = np.concatenate(clouds) if clouds else np.zeros(0, orc.POINT)
    t = torch.from_numpy(local.view(np.int32).reshape(-1, 4).copy())
    shards, counts = o3dist.all_gather_points(t)
    assert counts[rank] == len(local)
    big = np.concatenate([s.numpy().view(orc.POINT).reshape(-1) for s in shards])
    small, _ = orc.downsample_pt_cloud(big, vs, True, 1)
    np.save(os.path.join(out_dir, f"small_{rank}.npy"), small)
    np.save(os.path.join(out_dir, f"big_{rank}.npy"), big)
    dist.destroy_process_group()


def test_shard_ranges_cover_all_frames_in_order():
    from online_3d_reconstruction_amd.dist import shard_range
    for total in (0, 1, 5, 200, 2000, 2001):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_two_rank_exchange_equals_single_process(tmp_path, orc):
    from online_3d_reconstruction_amd import synth
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    F_total, rows, cols, jump, vs = 5, 240, 400, 2, 0.05
    Q = synth.camera_Q(rows, cols)
    clouds = []
    for i in range(F_total):
        d, c = synth.make_frame(i, rows, cols)
        clouds.append(orc.create_and_transform_pt_cloud(d, c, Q, synth.make_pose(i), vs, jump_pixels=jump)[0])
    big = np.concatenate(clouds)
    small, _ = orc.downsample_pt_cloud(big, vs, True, 1)
    for r in range(world):
        got_big = np.load(tmp_path / f"big_{r}.npy")
        got_small = np.load(tmp_path / f"small_{r}.npy")
        assert np.array_equal(got_big.view(np.uint32), big.view(np.uint32))
        assert np.array_equal(got_small.view(np.uint32), small.view(np.uint32))


# ---- partitioned merge (dist.merge_partitioned): the protocol with a CPU stand-in for the context -----
class CpuCtx:
    """Implements the handful of Context methods merge_partitioned uses, on the CPU oracle, so the
    exchange logic can run under gloo without a GPU.  (Test infrastructure only.)"""

    def __init__(self, orc, voxel_size):
        self.orc, self.vs = orc, voxel_size
        self.cloud = np.zeros(0, orc.POINT)

    def cloudBigBBox(self):
        c = self.cloud
        if len(c) == 0:
            return np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32), 0
        return (np.array([c[a].min() for a in "xyz"], np.float32), np.array([c[a].max() for a in "xyz"], np.float32), len(c))

    def _keys(self, gmin, gmax):
        # PCL's index for a grid over the GLOBAL box: two corner sentinels force that box
        aug = np.zeros(len(self.cloud) + 2, self.orc.POINT)
        aug[: len(self.cloud)] = self.cloud
        aug["z"] += np.float32(500)
        for k, a in enumerate("xyz"):
            aug[a][-2] = gmin[k] + (np.float32(500) if a == "z" else np.float32(0))
            aug[a][-1] = gmax[k] + (np.float32(500) if a == "z" else np.float32(0))
        keys, min_b, div_b, st = self.orc.voxel_keys(aug, [self.vs, self.vs, 1000.0])
        return keys[:-2], div_b, st

    def cloudBigPartition(self, gmin, gmax, n_parts):
        if len(self.cloud) == 0:
            return [0] * n_parts, 0
        keys, div_b, st = self._keys(gmin, gmax)
        if st:
            return [0] * n_parts, 1
        cells = int(div_b[0]) * int(div_b[1]) * int(div_b[2])
        part = np.minimum(keys.astype(np.uint64) * np.uint64(n_parts) // np.uint64(cells), n_parts - 1).astype(np.int64)
        order = np.argsort(part, kind="stable")
        self.cloud = self.cloud[order]
        return [int((part == p).sum()) for p in range(n_parts)], 0

    def cloudBigRead(self, device=None):
        return torch.from_numpy(self.cloud.view(np.int32).reshape(-1, 4).copy())

    def cloudBigReset(self):
        self.cloud = np.zeros(0, self.orc.POINT)

    def cloudBigAppend(self, t):
        self.cloud = np.concatenate([self.cloud, t.numpy().view(self.orc.POINT).reshape(-1)])

    def finalize(self, device=None, gmin=None, gmax=None):
        out, _ = self.orc.downsample_pt_cloud(self.cloud, self.vs, True, 1)
        return torch.from_numpy(out.view(np.int32).reshape(-1, 4).copy())


class CpuCtxZeroCopy(CpuCtx):
    """+ the zero-copy plumbing of the real context (o3dr_cloud_big_view / _recv_buffer / _adopt): the send buffer IS
    the cloud, the receive buffer belongs to the context, adopt makes its first n points the new cloud"""

    recv_cap = 1 << 40  # points the receive buffer "holds" (a warmed-up context: nothing has to grow)

    def cloudBigCapacity(self):
        return len(self.cloud), self.recv_cap

    def cloudBigView(self):
        self._send = torch.from_numpy(self.cloud.view(np.int32).reshape(-1, 4))  # aliases the cloud: no copy
        return self._send

    def cloudBigRecvBuffer(self, n_points):
        self._recv = torch.full((max(int(n_points), 1) + 3, 4), -1, dtype=torch.int32)  # (larger than asked, like the library's)
        return self._recv[: int(n_points)]

    def cloudBigAdopt(self, n_points):
        assert 0 <= n_points <= self._recv.shape[0]
        self.cloud = self._recv[: int(n_points)].numpy().view(self.orc.POINT).reshape(-1).copy()


class CpuCtxDev(CpuCtxZeroCopy):
    """+ the device-resident small data of the real context (o3dr_cloud_big_header_dev / _partition_dev / _assume_size):
    the header and the slice counts are tensors the collectives take as they are; finalize takes the size it is told"""

    def cloudBigHeaderDev(self):
        mn, mx, n = self.cloudBigBBox()
        raw = np.concatenate([mn, mx]).astype(np.float32).tobytes() + np.int64(n).tobytes()
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8)

    def cloudBigPartitionDev(self, hdrs, n_parts):
        from online_3d_reconstruction_amd.dist import _parse_headers
        gmin, gmax, counts = _parse_headers(hdrs.numpy(), hdrs.numel() // 32)
        if int(counts.sum()) == 0 or len(self.cloud) == 0:
            return torch.zeros(n_parts + 1, dtype=torch.int64)
        cnt, st = self.cloudBigPartition(gmin, gmax, n_parts)
        return torch.tensor(list(cnt) + [st], dtype=torch.int64)

    def cloudBigAssumeSize(self, n):
        assert n == len(self.cloud)

    def finalize(self, device=None, gmin=None, gmax=None, n_hint=None):
        assert n_hint is None or n_hint == len(self.cloud)
        return super().finalize(device, gmin, gmax)


class CpuCtxTwoPhase(CpuCtxDev):
    """+ the partition in two halves of the real context (o3dr_cloud_big_slice_counts_dev / _place_slices / _raw_view /
    _set_size): sizes first with nothing moved, then one placement [gap | own slice | gap | leaving slices]"""

    def cloudBigSliceCountsDev(self, hdrs, n_parts):
        from online_3d_reconstruction_amd.dist import _parse_headers
        gmin, gmax, counts = _parse_headers(hdrs.numpy(), hdrs.numel() // 32)
        self._part = None
        if int(counts[counts > 0].sum()) == 0 or len(self.cloud) == 0:
            return torch.zeros(n_parts + 1, dtype=torch.int64)
        keys, div_b, st = self._keys(gmin, gmax)
        if st:
            return torch.tensor([0] * n_parts + [1], dtype=torch.int64)
        cells = int(div_b[0]) * int(div_b[1]) * int(div_b[2])
        self._part = np.minimum(keys.astype(np.uint64) * np.uint64(n_parts) // np.uint64(cells), n_parts - 1).astype(np.int64)
        return torch.tensor([int((self._part == p).sum()) for p in range(n_parts)] + [0], dtype=torch.int64)

    def cloudBigPlaceSlices(self, own_part, counts, n_before, n_after):
        n_local, own = len(self.cloud), counts[own_part]
        assert sum(counts) == n_local
        send_start = n_before + own + n_after
        if n_before == 0 and n_after == 0 and own == n_local:
            return send_start  # (nothing moves: the cloud stays as it is)
        buf = np.zeros(send_start + (n_local - own) + 3, self.orc.POINT)
        buf["x"] = np.float32(np.nan)  # (a gap that is never filled would show)
        buf[n_before: n_before + own] = self.cloud[self._part == own_part]
        off = send_start
        for p in range(len(counts)):
            if p != own_part:
                buf[off: off + counts[p]] = self.cloud[self._part == p]
                off += counts[p]
        self._raw_pts = buf
        self._raw = torch.from_numpy(buf.view(np.int32).reshape(-1, 4))
        return send_start

    def cloudBigRawView(self):
        return self._raw

    def cloudBigSetSize(self, n):
        self.cloud = self._raw_pts[: int(n)].copy()
        assert not np.isnan(self.cloud["x"]).any()


class CpuCtxCold(CpuCtxDev):
    """a context whose receive buffer has to grow on this call (the first exchange of a run)"""
    recv_cap = 0


class Boom(RuntimeError):
    code = -6  # O3DR_ERR_ALLOC


class CpuCtxFailing(CpuCtxTwoPhase):
    """raises at one step of the exchange, like a failed allocation inside the library would (the two-phase form, with a
    receive buffer that has to grow)"""
    fail_at = None
    recv_cap = 0

    def cloudBigSliceCountsDev(self, hdrs, n_parts):
        if self.fail_at == "partition":
            raise Boom("partition")
        return super().cloudBigSliceCountsDev(hdrs, n_parts)

    def cloudBigHeaderDev(self):
        if self.fail_at == "header":
            raise Boom("header")
        return super().cloudBigHeaderDev()

    def cloudBigPartitionDev(self, hdrs, n_parts):
        if self.fail_at == "partition":
            raise Boom("partition")
        return super().cloudBigPartitionDev(hdrs, n_parts)

    def cloudBigRecvBuffer(self, n_points):
        if self.fail_at == "recv_buffer":
            raise Boom("recv_buffer")
        return super().cloudBigRecvBuffer(n_points)

    def finalize(self, device=None, gmin=None, gmax=None, n_hint=None):
        if self.fail_at == "finalize":
            raise Boom("finalize")
        return super().finalize(device, gmin, gmax, n_hint)


def _worker_partitioned(rank, world, port, out_dir, zero_copy=False):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from online_3d_reconstruction_amd import dist as o3dist
    from online_3d_reconstruction_amd import synth
    from oracle import orc
    F_total, rows, cols, jump, vs = 7, 240, 400, 2, 0.05
    Q = synth.camera_Q(rows, cols)
    a, b = o3dist.shard_range(F_total, rank, world)
    ctx = {False: CpuCtx, True: CpuCtxZeroCopy, "dev": CpuCtxDev, "cold": CpuCtxCold, "two_phase": CpuCtxTwoPhase}[zero_copy](orc, vs)
    for i in range(a, b):
        d, c = synth.make_frame(i, rows, cols)
        ctx.cloud = np.concatenate([ctx.cloud, orc.create_and_transform_pt_cloud(d, c, Q, synth.make_pose(i), vs, jump_pixels=jump)[0]])
    n_before = len(ctx.cloud)
    merged, total = o3dist.merge_partitioned(ctx, torch.device("cpu"))
    st = o3dist.last_stats
    # 5 collectives in all (headers, slice counts, all-to-all, merged sizes, merged slices); with the device-resident
    # small data the host waits twice before the final gather: for the count matrix and for the merged slice's size
    # (a context whose receive buffer must grow - "cold" - adds the 8-byte agreement all-gather and its read-back)
    # (the two-phase form sends to the higher and to the lower ranks in two all-to-alls: each receives into one gap)
    cold = zero_copy == "cold"
    assert st["collectives"] == (6 if zero_copy in ("cold", "two_phase") else 5)
    assert st["device_resident"] == (zero_copy in ("dev", "cold", "two_phase")) and st["two_phase"] == (zero_copy == "two_phase")
    assert st["agreement_rounds"] == (1 if cold else 0)
    assert st["host_syncs_before_final_gather"] == {False: 3, True: 3, "dev": 2, "cold": 3, "two_phase": 2}[zero_copy], st
    # what the exchange moved: every point of this rank either stayed or was sent; what entered the merge is what arrived
    assert st["points_local"] == n_before and 0 <= st["points_sent_off_rank"] <= n_before
    assert st["bytes_sent"] == 16 * st["points_sent_off_rank"] and st["bytes_received"] == 16 * st["points_received_off_rank"]
    assert st["points_into_merge"] == n_before - st["points_sent_off_rank"] + st["points_received_off_rank"]
    assert st["points_all_ranks"] == total
    if zero_copy:  # counts matrix -> receive counts: what arrived is this rank's slice of everybody's cloud
        sent = torch.tensor([n_before], dtype=torch.int64)
        got = torch.tensor([len(ctx.cloud)], dtype=torch.int64)
        dist.all_reduce(sent)
        dist.all_reduce(got)
        assert int(sent) == int(got) == total
    np.save(os.path.join(out_dir, f"merged_{rank}.npy"), merged.numpy().view(orc.POINT).reshape(-1))
    np.save(os.path.join(out_dir, f"total_{rank}.npy"), np.array([total]))
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("zero_copy", [False, True, "dev", "cold", "two_phase"])
def test_partitioned_merge_equals_single_process(tmp_path, orc, zero_copy):
    """3 ranks, 7 frames: slices exchanged all-to-all, merged locally, gathered == one-process merge; with the copying
    exchange (what a rehearsal over gloo uses) and with the zero-copy one the GPU path takes (send view, library-owned
    receive buffer, adopt), and with the header and slice counts as device-resident tensors ("dev": the protocol of the
    real context, two host waits before the final gather)"""
    from online_3d_reconstruction_amd import synth
    world = 3
    mp.spawn(_worker_partitioned, args=(world, _free_port(), str(tmp_path), zero_copy), nprocs=world, join=True)
    F_total, rows, cols, jump, vs = 7, 240, 400, 2, 0.05
    Q = synth.camera_Q(rows, cols)
    clouds = []
    for i in range(F_total):
        d, c = synth.make_frame(i, rows, cols)
        clouds.append(orc.create_and_transform_pt_cloud(d, c, Q, synth.make_pose(i), vs, jump_pixels=jump)[0])
    big = np.concatenate(clouds)
    small, _ = orc.downsample_pt_cloud(big, vs, True, 1)
    for r in range(world):
        got = np.load(tmp_path / f"merged_{r}.npy")
        assert int(np.load(tmp_path / f"total_{r}.npy")[0]) == len(big)
        assert len(got) == len(small) and np.array_equal(got.view(np.uint32), small.view(np.uint32))


def _worker_failing(rank, world, port, out_dir, fail_at, fail_rank):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from online_3d_reconstruction_amd import dist as o3dist
    from online_3d_reconstruction_amd import synth
    from oracle import orc
    rows, cols, jump, vs = 240, 400, 4, 0.05
    Q = synth.camera_Q(rows, cols)
    ctx = CpuCtxFailing(orc, vs)
    d, c = synth.make_frame(rank, rows, cols)
    ctx.cloud = orc.create_and_transform_pt_cloud(d, c, Q, synth.make_pose(rank), vs, jump_pixels=jump)[0]
    n_before = len(ctx.cloud)
    if rank == fail_rank:
        ctx.fail_at = fail_at
    try:
        o3dist.merge_partitioned(ctx, torch.device("cpu"))
        outcome = ("returned", -1, 0, False)
    except o3dist.ExchangeError as e:
        outcome = ("raised", e.rank, e.code, e.own)
    # the process group is still usable: no rank is stuck inside a collective of the exchange
    t = torch.tensor([1], dtype=torch.int64)
    dist.all_reduce(t)
    assert int(t) == world
    # ... and a second exchange, with nothing failing, goes through (finalize failures come after the all-to-all: the
    # clouds were exchanged by then, so only the count of all points is checked)
    ctx.fail_at = None
    merged, total = o3dist.merge_partitioned(ctx, torch.device("cpu"))
    np.save(os.path.join(out_dir, f"outcome_{rank}.npy"), np.array([outcome[0] == "raised", outcome[1], outcome[2], outcome[3], total, n_before, len(merged)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_at", ["header", "partition", "recv_buffer", "finalize"])
def test_partitioned_merge_failure_on_one_rank_is_collective(tmp_path, fail_at):
    """A rank whose own step fails must not leave its peers inside a collective: every rank raises
    ExchangeError naming the failing rank and its code, at the same point of the protocol; the process group stays
    usable and the next exchange succeeds."""
    world, fail_rank = 3, 1
    mp.spawn(_worker_failing, args=(world, _free_port(), str(tmp_path), fail_at, fail_rank), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"outcome_{r}.npy") for r in range(world)]
    for r in range(world):
        raised, bad_rank, code, own, total, n_before, n_merged = (int(v) for v in outs[r])
        assert raised == 1 and bad_rank == fail_rank and code == -6 and own == int(r == fail_rank)
        assert n_merged > 0
    assert len({int(o[4]) for o in outs}) == 1 and int(outs[0][4]) == sum(int(o[5]) for o in outs)
