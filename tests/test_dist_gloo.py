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
