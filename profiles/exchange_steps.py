#!/usr/bin/env python3
"""What the exchange's passes over a rank's own cloud cost on one MI355X (no links involved): the headline's cloud_big
(200 dense frames, 98 M points) cut into 8 index slices over its own box.
  old:  o3dr_cloud_big_partition_dev (count + move)  +  the copy of the own slice into the receive buffer (what a send to
        oneself is; timed here as a device copy of the same size)
  new:  o3dr_cloud_big_slice_counts_dev (count)  +  o3dr_cloud_big_place_slices (one move, gaps for what arrives)
python3 profiles/exchange_steps.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import online_3d_reconstruction_amd as o3dr  # noqa: E402
from online_3d_reconstruction_amd import synth  # noqa: E402

F, W = 200, 8
Q = synth.camera_Q()
disp_h, bgr_h = synth.make_frames(0, F)
poses_h = synth.make_poses(0, F)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = o3dr.Context(0, Q=Q, params=o3dr.Params(jump_pixels=1, voxel_size=0.05, sor_enable=False), stream=stream)
disp, bgr, poses = (torch.from_numpy(a).to(dev) for a in (disp_h, bgr_h, poses_h))
ctx.cloudBigReserve(F * ctx.max_points(720, 1280))


def fill():
    ctx.cloudBigReset()
    ctx.accumulateFrames(disp, bgr, poses)


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        fill()
        hdr = ctx.cloudBigHeaderDev()
        hdrs = hdr.repeat(W)  # eight ranks with this rank's box: the global box is its own, the slices eight equal index ranges
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(hdrs)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, out


res = {}
ms, row = timed(lambda h: ctx.cloudBigPartitionDev(h, W))
res["old_partition_count_and_move_ms"] = round(ms, 3)
counts = [int(v) for v in row.cpu().tolist()][:W]
n = sum(counts)
ms, _ = timed(lambda h: ctx.cloudBigSliceCountsDev(h, W))
res["new_slice_counts_ms"] = round(ms, 3)


def place(h):
    ctx.cloudBigSliceCountsDev(h, W)
    torch.cuda.synchronize()
    ctx.cloudBigAssumeSize(n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.cloudBigPlaceSlices(3, counts, 1000, 1000)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


best = 1e9
for _ in range(5):
    fill()
    hdr = ctx.cloudBigHeaderDev()
    best = min(best, place(hdr.repeat(W)))
res["new_place_slices_ms"] = round(best, 3)
# the copy a send to oneself amounts to: 95 % of the cloud, device to device
own = int(0.95 * n)
a = torch.empty((own, 4), dtype=torch.int32, device=dev)
b = torch.empty_like(a)
for _ in range(2):
    b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
b.copy_(a)
e1.record()
torch.cuda.synchronize()
res["old_own_slice_copy_ms_(95%_of_the_cloud,_device_copy)"] = round(e0.elapsed_time(e1), 3)
res["points"] = n
print(json.dumps(res))
