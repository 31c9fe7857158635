"""Regenerates the fixtures in this directory from the DATA files bundled with the reference
(run once in the build container, where /root/reference is mounted; the GPU box only sees the
committed outputs).  Only data travels: decoded pixels of bundled PNGs, the calibration numbers,
and the vertices of the bundled output cloud.  No reference source is read or copied.

    python tests/golden/make_golden.py
"""
import os
import re

import numpy as np
from PIL import Image

REF = "/root/reference/build"
OUT = os.path.dirname(os.path.abspath(__file__))


def read_Q(path):
    txt = open(path).read()
    m = re.search(r"Q: !!opencv-matrix.*?data: \[(.*?)\]", txt, re.S)
    return np.array([float(v) for v in m.group(1).replace("\n", " ").split(",")], np.float64).reshape(4, 4)


def load_pair(name):
    disp = np.array(Image.open(f"{REF}/disparities/{name}.png").convert("L"))  # IMREAD_GRAYSCALE, :548
    rgb = np.array(Image.open(f"{REF}/images/{name}.png").convert("RGB"))
    return disp, np.ascontiguousarray(rgb[:, :, ::-1])  # cv::imread gives B,G,R (:526)


def read_ply_vertices(path):
    raw = open(path, "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    header = raw[:end].decode()
    n = int(re.search(r"element vertex (\d+)", header).group(1))
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    v = np.frombuffer(raw, dt, n, end)
    return header, v, raw[end + n * dt.itemsize:]


def main():
    Q = read_Q(f"{REF}/data_files/cam13calib.yml")
    np.save(f"{OUT}/cam13calib_Q.npy", Q)

    # synthetic pair shipped with the reference (two disparities, two colours)
    d, c = load_pair("B")
    np.savez_compressed(f"{OUT}/frame_B.npz", disp=d, bgr=c)

    # all seven real frames inside config 1's range (1230-1280) that have both image and disparity.  1239 and 1240
    # carry the only real invalid pixels (11 759 / 1 978 in the ROI) and are rejected by the variance gate of
    # pose.cpp:187-196; the other five are what config 1 really accepts.
    for name in ("1239", "1240", "1246", "1248", "1249", "1251", "1255"):
        d, c = load_pair(name)
        np.savez_compressed(f"{OUT}/frame_{name}.npz", disp=d, bgr=c)

    # bundled output cloud: layout fixture + "one point per 0.05 m XY cell" invariant
    header, v, tail = read_ply_vertices(f"{REF}/cloud.ply")
    np.savez_compressed(f"{OUT}/cloud_ply.npz", header=np.frombuffer(header.encode(), np.uint8),
                        vertices=v, tail=np.frombuffer(tail, np.uint8))
    # pose / image-time tables (data): the timestamp -> pose binding of the CLI depends on the whole
    # table (binary-search path), so they are kept whole, gzip-compressed
    import gzip
    for name in ("pose.txt", "images.txt"):
        raw = open(f"{REF}/data_files/{name}", "rb").read()
        with gzip.GzipFile(f"{OUT}/{name}.gz", "wb", mtime=0) as g:
            g.write(raw)
    # positions an earlier run of the reference wrote, one "tx,ty,tz" line per accepted image (the translation of the
    # pose row its timestamp search bound to that image): lines 1-25 are images 1242..1266, lines 26-30 are images
    # 1295, 1297, 1298, 1299, 1301; later lines belong to other runs appended to the same file and are not kept
    lines = open(f"{REF}/output/hexPosMAVLink.txt").read().splitlines()[:30]
    with open(f"{OUT}/hexPosMAVLink_first30.txt", "w") as f:
        f.write("\n".join(lines) + "\n")
    # outputs of cv::bilateralFilter the reference holds: build/images/1248.png through the call of pose_functions.cpp:1044
    # with blur_kernel 15 / 31 (decoded pixels, B,G,R like cv::imread); the input is frame_1248.npz's `bgr`
    for d in (15, 31):
        t = np.array(Image.open(f"{REF}/output/bilateralFiltered_{d}.png").convert("RGB"))
        np.savez_compressed(f"{OUT}/bilateralFiltered_{d}.npz", bgr=np.ascontiguousarray(t[:, :, ::-1]))
    # build/cloud_uavpos.ply (pose.cpp:551-553): 42 feature-matched positions followed by the 42 MAVLink positions of the
    # same accepted images, as "x y z r g b" text lines (%.9g round-trips a float32)
    _, v, _ = read_ply_vertices(f"{REF}/cloud_uavpos.ply")
    with open(f"{OUT}/cloud_uavpos_vertices.txt", "w") as f:
        for p in v:
            f.write("%.9g %.9g %.9g %d %d %d\n" % (p["x"], p["y"], p["z"], p["r"], p["g"], p["b"]))
    print("Q =", Q.ravel())
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(f"{OUT}/{f}"))


if __name__ == "__main__":
    main()
