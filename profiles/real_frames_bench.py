import sys, time, json
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import torch
import online_3d_reconstruction_amd as o3dr
from online_3d_reconstruction_amd import synth
from conftest import load_frame
names = ["1246", "1248", "1249", "1251", "1255"]
fr = [load_frame(n) for n in names]
F = 200
disp = np.stack([fr[i % 5][0] for i in range(F)])
bgr = np.stack([fr[i % 5][1] for i in range(F)])
poses = synth.make_poses(0, F)
Q = np.load('tests/golden/cam13calib_Q.npy')
dev = torch.device('cuda', 0)
ctx = o3dr.Context(0, Q=Q, params=o3dr.Params(jump_pixels=1, voxel_size=0.05, sor_enable=False), stream=torch.cuda.current_stream())
d, c, p = torch.from_numpy(disp).to(dev), torch.from_numpy(bgr).to(dev), torch.from_numpy(poses).to(dev)
ctx.cloudBigReserve(F * 748000)
def step():
    ctx.cloudBigReset(); ctx.accumulateFrames(d, c, p); n, _ = ctx.cloudBigSize(); out = ctx.finalize(device=dev); return n, out.shape[0]
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): n, m = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(json.dumps({"what": "200 dense 720p frames built from the five REAL accepted frames of config 1 (bundled disparities/images, lawn-mower poses)", "ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(F / dt, 1), "per_frame_voxels_total": n, "merged_cells": m}))
# the same frames with the reference's statistical outlier removal on (its literal per-frame path)
ctx.set_params(o3dr.Params(jump_pixels=1, voxel_size=0.05, sor_enable=True))
step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2): n, m = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
print(json.dumps({"what": "the same 200 real frames with statistical outlier removal on", "ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(F / dt, 1), "per_frame_voxels_total": n, "merged_cells": m}))
