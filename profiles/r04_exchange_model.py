"""How many points does the multi-GPU exchange move?  CPU model (numpy) of dist.merge_partitioned / o3dr_merge_partitioned on
the synthetic lawn-mower track: every frame's points are sampled on a coarse pixel grid (every 8th pixel, the frame's real
disparities and pose), ranks own contiguous blocks of frames, the combined grid (voxel_size 0.05) is laid over the global
box and its linear index cut into W equal slices.  Reports, per configuration:
  now      : fraction of all points whose slice is not their own rank (what the all-to-all moves today; the 4 x 50 row can
             be checked against the measured profiles/r04_rehearsal_4ranks.json: 0.26)
  bbox     : the same if points in cells outside every OTHER rank's bounding box stayed where they are (merged locally)
  tiles    : the same with the test done on 32 x 32-cell tiles (1.6 m) of the combined grid instead of bounding boxes
    python profiles/r04_exchange_model.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from online_3d_reconstruction_amd import synth  # noqa: E402
from online_3d_reconstruction_amd.dist import shard_range  # noqa: E402


def frame_xy(i, Q, step=8):
    d, _ = synth.make_frame(i)
    T = synth.make_pose(i).astype(np.float64)
    ys, xs = np.mgrid[20:700:step, 160:1260:step]
    dd = d[ys, xs].astype(np.float64)
    ok = dd > 64
    w = Q[3, 2] * dd
    X = (xs + Q[0, 3]) / w
    Y = (ys + Q[1, 3]) / w
    Z = Q[2, 3] / w
    wx = T[0, 0] * X + T[0, 1] * Y + T[0, 2] * Z + T[0, 3]
    wy = T[1, 0] * X + T[1, 1] * Y + T[1, 2] * Z + T[1, 3]
    return wx[ok], wy[ok]


def model(W, frames_per_rank, vs=0.05, cache={}):
    Q = synth.camera_Q()
    F = W * frames_per_rank
    cx, cy, rk = [], [], []
    for r in range(W):
        a, b = shard_range(F, r, W)
        for i in range(a, b):
            if i not in cache:
                x, y = frame_xy(i, Q)
                cache[i] = (np.floor(x / vs).astype(np.int64), np.floor(y / vs).astype(np.int64))
            ix, iy = cache[i]
            cx.append(ix)
            cy.append(iy)
            rk.append(np.full(len(ix), r, np.int32))
    cx, cy, rk = np.concatenate(cx), np.concatenate(cy), np.concatenate(rk)
    mnx, mny = cx.min(), cy.min()
    dx, dy = cx.max() - mnx + 1, cy.max() - mny + 1
    idx = (cx - mnx) + (cy - mny) * dx
    sl = np.minimum(idx * W // (dx * dy), W - 1)
    moved_now = sl != rk
    # bounding boxes of the ranks (in cells)
    boxes = [(cx[rk == r].min(), cx[rk == r].max(), cy[rk == r].min(), cy[rk == r].max()) for r in range(W)]
    in_other_box = np.zeros(len(cx), bool)
    for r, (x0, x1, y0, y1) in enumerate(boxes):
        in_other_box |= (rk != r) & (cx >= x0) & (cx <= x1) & (cy >= y0) & (cy <= y1)
    # 32 x 32-cell tiles occupied per rank
    tx, ty = cx >> 5, cy >> 5
    t0x, t0y = tx.min(), ty.min()
    ntx = tx.max() - t0x + 1
    tid = (tx - t0x) + (ty - t0y) * ntx
    occ = np.zeros((W, int(tid.max()) + 1), bool)
    occ[rk, tid] = True
    others = occ.sum(axis=0)[tid] - 1 > 0  # some other rank has points in this tile
    n = len(cx)
    per_rank_now = [int((moved_now & (rk == r)).sum()) for r in range(W)]
    return {"W": W, "frames_per_rank": frames_per_rank, "now": round(float(moved_now.mean()), 4),
            "bbox": round(float((moved_now & in_other_box).mean()), 4), "tiles": round(float((moved_now & others).mean()), 4),
            "busiest_rank_share_now": round(max(per_rank_now) / n * W, 3),
            "shared_fraction_tiles": round(float(others.mean()), 4)}


if __name__ == "__main__":
    import json
    for W, fpr in ((4, 50), (2, 100), (8, 25), (8, 200), (8, 250)):
        print(json.dumps(model(W, fpr)), flush=True)
