#!/usr/bin/env python3
"""200 calls of o3dr_create_and_transform_pt_cloud on one --jump_pixels 15 (or argv[1]) frame with the outlier removal on:
run under `rocprofv3 --kernel-trace --stats` to see what the one-launch path's kernels take (profiles/README.md)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import online_3d_reconstruction_amd as o3dr  # noqa: E402
from online_3d_reconstruction_amd import synth  # noqa: E402

jump = int(sys.argv[1]) if len(sys.argv) > 1 else 15
sor = "--no-sor" not in sys.argv
Q = synth.camera_Q()
d, c = synth.make_frame(0)
T = synth.make_pose(0)
ctx = o3dr.Context(0, Q=Q, params=o3dr.Params(jump_pixels=jump, voxel_size=0.05, sor_enable=sor))
dd, cc = torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()
for _ in range(5):
    out = ctx.createAndTransformPtCloud(dd, cc, T)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    out = ctx.createAndTransformPtCloud(dd, cc, T)
torch.cuda.synchronize()
print(f"jump {jump} sor {sor}: {len(out)} points out, {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call")
