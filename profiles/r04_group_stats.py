"""Round 4, option (a) of VERDICT round 3 item 2: what a 3-pass sort of the 21-bit GROUP key `index >> 7` leaves for the
sums kernel.  Numpy over the oracle's per-frame voxel indices (CPU only): records per group, voxels per group, how full
64-record steps cut at group boundaries would be, and how the top digit of an MSD partition (option (b)) spreads.

    python profiles/r04_group_stats.py            # synthetic frames 0, 57 and the real frames 1248, 1255
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from online_3d_reconstruction_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402


def frame_keys(disp, bgr, Q, T, vs=0.05):
    world = orc.transform_pt_cloud(orc.create_single_img_pt_cloud(disp, bgr, Q, jump_pixels=1), T)
    leaf = np.float32(vs / 5)
    keys, min_b, div_b, st = orc.voxel_keys(world, [leaf, leaf, leaf])
    return keys.astype(np.int64), div_b, st


def describe(name, keys, div_b, gbits=7):
    n = len(keys)
    cells = int(div_b[0]) * int(div_b[1]) * int(div_b[2])
    nbits = int(np.ceil(np.log2(cells)))
    order = np.argsort(keys, kind="stable")
    sk = keys[order]
    vox = np.count_nonzero(np.diff(sk)) + 1
    grp = sk >> gbits
    gstart = np.flatnonzero(np.diff(grp, prepend=-1))
    gsize = np.diff(np.append(gstart, n))
    vox_heads = np.diff(sk, prepend=-1) != 0
    vpg = np.add.reduceat(vox_heads.astype(np.int64), gstart)
    # steps of <= 64 records cut at group boundaries (a group longer than 64 takes ceil(size / 64) steps of its own)
    steps = 0
    fill = 0
    cur = 0
    for s in gsize:
        if s > 64:
            if cur:
                steps += 1
                cur = 0
            steps += -(-int(s) // 64)
            continue
        if cur + s > 64:
            steps += 1
            cur = 0
        cur += int(s)
    if cur:
        steps += 1
    fill = n / (64.0 * steps)
    # option (b): records per bucket of the TOP 7 bits of the index (one global MSD pass), and of the top 14
    top7 = np.bincount((sk >> max(nbits - 7, 0)).astype(np.int64))
    top14 = np.bincount((sk >> max(nbits - 14, 0)).astype(np.int64))
    q = lambda a, p: int(np.percentile(a, p))
    print(f"{name}: {n} points, {vox} voxels ({n / vox:.2f} points per voxel), grid {tuple(int(v) for v in div_b)} = {nbits} bits")
    print(f"  groups of 2^{gbits} cells: {len(gsize)} groups, records per group mean {gsize.mean():.1f} median {q(gsize, 50)} "
          f"p90 {q(gsize, 90)} p99 {q(gsize, 99)} max {gsize.max()}; voxels per group mean {vpg.mean():.1f} max {vpg.max()}")
    print(f"  records in groups above 64: {gsize[gsize > 64].sum() / n:.3f} of all; 64-record steps cut at group boundaries: "
          f"{steps} steps, {fill:.2f} full")
    nz7, nz14 = top7[top7 > 0], top14[top14 > 0]
    print(f"  MSD top 7 bits: {len(nz7)} non-empty buckets, mean {nz7.mean():.0f} max {nz7.max()} records; "
          f"top 14 bits: {len(nz14)} buckets, mean {nz14.mean():.0f} p99 {q(nz14, 99)} max {nz14.max()}")


def main():
    Q = synth.camera_Q()
    for i in (0, 57):
        d, c = synth.make_frame(i)
        k, div_b, st = frame_keys(d, c, Q, synth.make_pose(i))
        describe(f"synthetic frame {i}", k, div_b)
    from conftest import load_frame
    from test_cli_pose import pose_row_for_image
    Qr = np.load(os.path.join(ROOT, "tests", "golden", "cam13calib_Q.npy"))
    for name in ("1248", "1255"):
        d, c = load_frame(name)
        _, row = pose_row_for_image(int(name))
        k, div_b, st = frame_keys(d, c, Qr, synth.generate_tmat(row[3:6], row[6:10]))
        describe(f"real frame {name}", k, div_b)


if __name__ == "__main__":
    main()
