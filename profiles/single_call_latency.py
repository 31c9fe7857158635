#!/usr/bin/env python3
"""Latency of the per-frame entry points (what the reference's threads call once per accepted frame) on one dense 720p frame:
o3dr_create_and_transform_pt_cloud with inputs and outputs in HBM / in host memory, and the batched call with 1, 7 and 32
frames per call.  python profiles/single_call_latency.py [--sor]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import online_3d_reconstruction_amd as o3dr  # noqa: E402
from online_3d_reconstruction_amd import synth  # noqa: E402

rows, cols = 720, 1280
Q = synth.camera_Q(rows, cols)
disp_h, bgr_h = synth.make_frames(0, 32, rows, cols)
poses_h = synth.make_poses(0, 32)
dev = torch.device("cuda", 0)
SOR = "--sor" in sys.argv  # the reference's default for jump_pixels > 0; off here unless asked for: launch overheads are the subject
ctx = o3dr.Context(0, Q=Q, params=o3dr.Params(jump_pixels=1, voxel_size=0.05, sor_enable=SOR))
disp, bgr, poses = (torch.from_numpy(a).to(dev) for a in (disp_h, bgr_h, poses_h))
out = {}


def timed(name, fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    out[name] = round((time.perf_counter() - t0) / reps * 1e6, 1)


timed("create_and_transform_device_us", lambda: ctx.createAndTransformPtCloud(disp[0], bgr[0], poses_h[0]))
timed("create_and_transform_host_us", lambda: ctx.createAndTransformPtCloud(disp_h[0], bgr_h[0], poses_h[0]))
# the reference's own frame sizes: --jump_pixels 15 (3 404 candidates) and 10 (7 480): one launch of one workgroup
# (kernels/small.inc) unless O3DR_SMALL=0
for jump in (15, 10):
    ctx.set_params(o3dr.Params(jump_pixels=jump, voxel_size=0.05, sor_enable=SOR))
    timed(f"create_and_transform_jump{jump}_device_us", lambda: ctx.createAndTransformPtCloud(disp[0], bgr[0], poses_h[0]), 300)
    timed(f"create_and_transform_jump{jump}_host_us", lambda: ctx.createAndTransformPtCloud(disp_h[0], bgr_h[0], poses_h[0]), 300)
small_cloud = ctx.createAndTransformPtCloud(disp_h[0], bgr_h[0], poses_h[0])
timed("downsample_combined_small_cloud_host_us", lambda: ctx.downsamplePtCloud(small_cloud, True), 300)
ctx.set_params(o3dr.Params(jump_pixels=1, voxel_size=0.05, sor_enable=SOR))
out["small_path"] = os.environ.get("O3DR_SMALL", "1") != "0"
ctx.cloudBigReserve(32 * ctx.max_points(rows, cols))
for nf in (1, 7, 32):
    def step(nf=nf):
        ctx.cloudBigReset()
        ctx.accumulateFrames(disp[:nf], bgr[:nf], poses[:nf])
    timed(f"accumulate_{nf}_frames_device_us", step, 50)
out["sor"] = SOR
print(json.dumps(out))
