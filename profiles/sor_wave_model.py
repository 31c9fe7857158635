#!/usr/bin/env python3
"""A numpy model of one wave of k_sor_knn (kernels/sor.inc): the shared candidate stream over the rings of an XY grid, per-lane
bounds, the merge trigger, the conservative stop and the hand-over to k_sor_knn_left - to COUNT what the kernel does
(candidates per wave, merges per wave, appended entries per lane, rings walked), not to compute results.  Used in round 3
to see where the kernel's vector instructions go: 7.2 per candidate (ISA), ~1 580 per merge, ~1.5 k for the square roots;
on synthetic frame 0 with 30 points per column it counts ~1 250 candidates and 7.0 merges per wave = 21.5 k instructions,
against 21-23 k measured (SQ_INSTS_VALU).  Statistics only: the points come from a plain float64 reprojection here.

    python profiles/sor_wave_model.py [points per column = 30] [waves sampled = 150] [--real=1248]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_3d_reconstruction_amd import synth  # noqa: E402

REAL = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--real=")), None)
ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
CELLPTS = float(ARGS[0]) if len(ARGS) > 0 else 30.0
N_WAVES = int(ARGS[1]) if len(ARGS) > 1 else 150
PRUNE_AT, CHECK_EVERY, K, RING_CAP, MIN_LIVE = 48, 16, 51, 2, 12


def frame_points(index, rows=720, cols=1280, bb=20, min_disp=64):
    Q = synth.camera_Q(rows, cols)
    disp, _ = synth.make_frame(index, rows, cols)
    if REAL:  # one of the bundled real frames (tests/golden, data only) under the synthetic pose, like real_frames_bench.py
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        Q = np.load(os.path.join(root, "tests", "golden", "cam13calib_Q.npy"))
        disp = np.load(os.path.join(root, "tests", "golden", f"frame_{REAL}.npz"))["disp"]
        rows, cols = disp.shape
    T = np.asarray(synth.make_pose(index), np.float64).reshape(4, 4)
    cs = cols // 8
    v, u = np.mgrid[bb:rows - bb, cs:cols - bb]
    d = disp[bb:rows - bb, cs:cols - bb].astype(np.float64)
    ok = d > min_disp
    uvd1 = np.stack([u[ok], v[ok], d[ok], np.ones(ok.sum())], 1).astype(np.float64)
    X = uvd1 @ Q.T
    P = X[:, :3] / X[:, 3:4]
    return (np.c_[P, np.ones(len(P))] @ T.T)[:, :3].astype(np.float32)


def main():
    P = frame_points(0)
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    n = len(x)
    ex, ey = float(x.max()) - float(x.min()), float(y.max()) - float(y.min())
    h = np.sqrt(CELLPTS * ex * ey / n)
    gx, gy = int(ex / h) + 1, int(ey / h) + 1
    inv = np.float32(1.0 / h)
    cx = np.clip(((x - x.min()) * inv).astype(np.int32), 0, gx - 1)
    cy = np.clip(((y - y.min()) * inv).astype(np.int32), 0, gy - 1)
    cell = cy * gx + cx
    order = np.argsort(cell, kind="stable")
    sx, sy, sz, scx, scy = x[order], y[order], z[order], cx[order], cy[order]
    cnt = np.bincount(cell, minlength=gx * gy)
    first = np.concatenate([[0], np.cumsum(cnt)])
    print(f"n {n}  h {h:.4f}  grid {gx} x {gy}  points per occupied column {cnt[cnt > 0].mean():.1f}")
    waves = np.random.default_rng(0).choice(n // 64, N_WAVES, replace=False)
    tot_c, tot_m, rings, acc, left_n = [], [], [], [], 0
    for w in waves:
        q = np.arange(w * 64, w * 64 + 64)
        qx, qy, qz = sx[q], sy[q], sz[q]
        kept = np.full((64, 64), np.inf, np.float32)
        nk = np.zeros(64, int)
        worst = np.full(64, np.inf, np.float32)
        add = [[] for _ in range(64)]
        st = {"cand": 0, "merge": 0}
        nacc = np.zeros(64, int)
        max_ring = 0

        def merge():
            st["merge"] += 1
            for l in range(64):
                if add[l]:
                    kept[l] = np.sort(np.concatenate([kept[l], np.array(add[l], np.float32)]))[:64]
                    nk[l] = min(64, nk[l] + len(add[l]))
                    add[l].clear()
                if np.isfinite(kept[l][K - 1]):
                    worst[l] = kept[l][K - 1]

        for grow in np.unique(scy[q]):
            mine = scy[q] == grow
            ca, cb = scx[q][mine].min(), scx[q][mine].max()
            live = mine.copy()
            rmax = max(gx, gy)
            for r in range(rmax + 1):
                if r == 0:
                    segs = [(grow, ca, cb)]
                else:
                    segs = [(grow - r, ca - r, cb + r), (grow + r, ca - r, cb + r)]
                    segs += [(grow - r + 1 + ((i - 2) >> 1),) + ((cb + r,) * 2 if i & 1 else (ca - r,) * 2) for i in range(2, 4 * r)]
                for yy, c0, c1 in segs:
                    if yy < 0 or yy >= gy:
                        continue
                    c0, c1 = max(c0, 0), min(c1, gx - 1)
                    if c0 > c1:
                        continue
                    s, e = first[yy * gx + c0], first[yy * gx + c1 + 1]
                    for j0 in range(s, e, 64):
                        for b in range(0, min(64, e - j0), CHECK_EVERY):
                            js = np.arange(j0 + b, min(j0 + b + CHECK_EVERY, e))
                            dx, dy, dz = qx[:, None] - sx[js][None, :], qy[:, None] - sy[js][None, :], qz[:, None] - sz[js][None, :]
                            d2 = (dx * dx + dy * dy) + dz * dz
                            st["cand"] += len(js)
                            ok = d2 < np.where(live, worst, -1.0)[:, None]
                            for l in np.nonzero(ok.any(1))[0]:
                                add[l].extend(d2[l][ok[l]].tolist())
                                nacc[l] += ok[l].sum()
                            if max(len(a) for a in add) > PRUNE_AT:
                                merge()
                if r > 0 and 3.14159 * r * r * CELLPTS >= K:
                    if any(live[l] and nk[l] + len(add[l]) >= K for l in range(64)):
                        if any(len(a) for a in add):
                            merge()
                        bd = 0.999 * r * h
                        live &= ~(live & (nk >= K) & np.isfinite(worst) & (worst.astype(np.float64) <= bd * bd))
                max_ring = max(max_ring, r)
                if not live.any():
                    break
                if RING_CAP <= r < rmax and live.sum() < MIN_LIVE:
                    left_n += live.sum()
                    break
        tot_c.append(st["cand"])
        tot_m.append(st["merge"])
        rings.append(max_ring)
        acc.append(nacc)
    acc = np.array(acc)
    c, m = np.mean(tot_c), np.mean(tot_m)
    print(f"candidates per wave {c:.0f}  merges per wave {m:.2f}  last ring (histogram) {np.bincount(rings)}  "
          f"appended per lane {acc.mean():.0f} (wave's busiest lane {acc.max(1).mean():.0f})  handed over per wave {left_n / len(waves):.2f}")
    print(f"vector instructions per wave ~ 7.2 x {c:.0f} + 1580 x {m:.2f} + 1500 = {7.2 * c + 1580 * m + 1500:.0f}")


if __name__ == "__main__":
    main()
