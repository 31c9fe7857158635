"""Does the GPU overlap two independent pipelines?  Two contexts with their own streams take 100 frames each from two host
threads (accumulate only: the per-frame path); compared with one context taking the 200 frames.  Informational."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + "/profiles")
from bench import generate_frames  # noqa: E402
from online_3d_reconstruction_amd import synth  # noqa: E402
F = 200
disp_h, bgr_h = generate_frames(0, F, 720, 1280, 0.0, 16)
poses_h = synth.make_poses(0, F)
import torch  # noqa: E402
import online_3d_reconstruction_amd as o3dr  # noqa: E402
dev = torch.device("cuda", 0)
Q = synth.camera_Q()
P = o3dr.Params(jump_pixels=1, voxel_size=0.05, min_points_per_voxel=1, sor_enable=False)
disp, bgr, poses = (torch.from_numpy(a).to(dev) for a in (disp_h, bgr_h, poses_h))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
one = o3dr.Context(0, Q=Q, params=P, stream=torch.cuda.current_stream())
two = [o3dr.Context(0, Q=Q, params=P, stream=s) for s in streams]
n_cand = one.max_points(720, 1280)
one.cloudBigReserve(F * n_cand)
for c in two:
    c.cloudBigReserve(F // 2 * n_cand)

def run_one():
    one.cloudBigReset()
    one.accumulateFrames(disp, bgr, poses)

def half(i):
    c = two[i]
    c.cloudBigReset()
    c.accumulateFrames(disp[i * 100:(i + 1) * 100], bgr[i * 100:(i + 1) * 100], poses[i * 100:(i + 1) * 100])

def run_two():
    ts = [threading.Thread(target=half, args=(i,)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()

for name, fn in (("one context, 200 frames", run_one), ("two contexts, 100 frames each, two threads", run_two),
                 ("one context, 200 frames", run_one), ("two contexts, 100 frames each, two threads", run_two)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per 200 frames (accumulate only)", flush=True)
