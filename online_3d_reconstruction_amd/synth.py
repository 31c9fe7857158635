"""Deterministic synthetic inputs for the benchmark configs (SURVEY.md section 8d / BASELINE.md 3).

Every frame is a pure function of (seed, frame index), so a rank can generate exactly its own
shard.  Nothing here is on the measured path: frames are generated once and uploaded to HBM before
the timed region starts.
"""
import numpy as np

# Q of the bundled rig, build/data_files/cam13calib.yml:91-97 (values; row-major 4x4)
CAM13_Q = np.array([1., 0., 0., -5.2425751876831055e+02,
                    0., 1., 0., -5.1381009292602539e+02,
                    0., 0., 0., 4.2300101518980237e+03,
                    0., 0., 1.6853548938735339e+00, 0.], np.float64).reshape(4, 4)

# camera mounting constants, pose.h:142-147
TRANS_HI = (-0.300, -0.040, -0.350)
THETA_XI = -1.1408 * 3.141592653589793238463 / 180
THETA_YI = 1.1945 * 3.141592653589793238463 / 180


def camera_Q(rows=720, cols=1280):
    """cam13 Q; for other resolutions the principal point scales with the image, f and Q32 stay."""
    Q = CAM13_Q.copy()
    Q[0, 3] *= cols / 1280.0
    Q[1, 3] *= rows / 720.0
    return Q


def make_frame(index, rows=720, cols=1280, seed=1234, invalid_frac=0.0):
    """One (disparity u8 [H,W], colour BGR u8 [H,W,3]) pair.

    d = clip(round(108 + 6 sin(2 pi x/257) cos(2 pi y/193) + box bumps(+14) + U{-1,0,1}), 0, 255),
    matching the bundled frames' 99-121 range; `invalid_frac` of the pixels are zeroed (<= 64)."""
    rng = np.random.default_rng([seed, index])
    x = np.arange(cols, dtype=np.float64)[None, :]
    y = np.arange(rows, dtype=np.float64)[:, None]
    d = 108.0 + 6.0 * np.sin(2 * np.pi * (x + 17.0 * index) / 257.0) * np.cos(2 * np.pi * y / 193.0)
    for _ in range(3):  # "buildings": +14 disparity levels (~2.5 m)
        bw, bh = int(rng.integers(cols // 16, cols // 6)), int(rng.integers(rows // 12, rows // 5))
        bx, by = int(rng.integers(0, cols - bw)), int(rng.integers(0, rows - bh))
        d[by:by + bh, bx:bx + bw] += 14.0
    d = d + rng.integers(-1, 2, size=(rows, cols))
    disp = np.clip(np.rint(d), 0, 255).astype(np.uint8)
    if invalid_frac > 0:
        disp[rng.random((rows, cols)) < invalid_frac] = 0
    crng = np.random.default_rng([4321, seed, index])
    bgr = crng.integers(0, 256, size=(rows, cols, 3), dtype=np.uint8)
    return disp, bgr


def make_frames(start, count, rows=720, cols=1280, seed=1234, invalid_frac=0.0):
    disp = np.empty((count, rows, cols), np.uint8)
    bgr = np.empty((count, rows, cols, 3), np.uint8)
    for i in range(count):
        disp[i], bgr[i] = make_frame(start + i, rows, cols, seed, invalid_frac)
    return disp, bgr


def _m4(rows):
    return np.array(rows, np.float32).reshape(4, 4)


def _mul(a, b):
    """4x4 float product, coefficient (i,j) = ((a_i0 b_0j + a_i1 b_1j) + a_i2 b_2j) + a_i3 b_3j in fp32."""
    out = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            s = np.float32(a[i, 0] * b[0, j])
            for k in range(1, 4):
                s = np.float32(s + np.float32(a[i, k] * b[k, j]))
            out[i, j] = s
    return out


def generate_tmat(t, q):
    """Host restatement of Pose::generateTmat (pose_functions.cpp:1178-1356): quaternion (qx,qy,qz,qw) +
    translation + camera mounting -> 4x4 float pose.  The rotation is built in fp64 (:1281-1303),
    narrowed into float 4x4 factors and multiplied left to right in fp32 (:1341).  The reference's
    Eigen product may order/fuse the inner sums differently (agreement ~1e-6)."""
    tx, ty, tz = (float(v) for v in t)
    qx, qy, qz, qw = (float(v) for v in q)
    sqw, sqx, sqy, sqz = qw * qw, qx * qx, qy * qy, qz * qz
    if not (0.99 <= sqw + sqx + sqy + sqz <= 1.01):
        raise ValueError("quaternion should be homogeneous")  # :1278
    rot = np.zeros((3, 3), np.float64)
    rot[0, 0] = sqx - sqy - sqz + sqw
    rot[1, 1] = -sqx + sqy - sqz + sqw
    rot[2, 2] = -sqx - sqy + sqz + sqw
    rot[0, 1] = 2.0 * (qx * qy + qz * qw)
    rot[1, 0] = 2.0 * (qx * qy - qz * qw)
    rot[0, 2] = 2.0 * (qx * qz - qy * qw)
    rot[2, 0] = 2.0 * (qx * qz + qy * qw)
    rot[1, 2] = 2.0 * (qy * qz + qx * qw)
    rot[2, 1] = 2.0 * (qy * qz - qx * qw)
    rot = rot.T  # :1305
    c, s = np.cos, np.sin
    r_xi = _m4([[1, 0, 0, 0], [0, c(THETA_XI), -s(THETA_XI), 0], [0, s(THETA_XI), c(THETA_XI), 0], [0, 0, 0, 1]])
    r_yi = _m4([[c(THETA_YI), 0, s(THETA_YI), 0], [0, 1, 0, 0], [-s(THETA_YI), 0, c(THETA_YI), 0], [0, 0, 0, 1]])
    r_invert_i = _m4([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])
    r_invert_y = _m4([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    t_hi = _m4([[1, 0, 0, TRANS_HI[0]], [0, 1, 0, TRANS_HI[1]], [0, 0, 1, TRANS_HI[2]], [0, 0, 0, 1]])
    r_flip_xy = _m4([[0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    r_wh = np.eye(4, dtype=np.float32)
    r_wh[:3, :3] = rot.astype(np.float32)
    t_wh = _m4([[1, 0, 0, tx], [0, 1, 0, ty], [0, 0, 1, tz], [0, 0, 0, 1]])
    m = t_wh
    for f in (r_wh, r_invert_y, r_flip_xy, t_hi, r_invert_i, r_yi, r_xi):  # :1341, left to right
        m = _mul(m, f)
    return m


def make_pose(index, seed=99):
    """Lawn-mower track at z = 22 m: 0.98 m between frames (observed 1248 -> 1249), 40 frames per leg,
    legs 6 m apart, yaw along the leg, +-1 degree roll/pitch noise."""
    rng = np.random.default_rng([seed, index])
    leg, k = divmod(index, 40)
    along = 0.98 * (k if leg % 2 == 0 else 39 - k)
    tx, ty, tz = along, 6.0 * leg, 22.0
    yaw = 0.0 if leg % 2 == 0 else np.pi
    roll, pitch = np.deg2rad(rng.uniform(-1, 1, 2))
    cy, sy, cp, sp, cr, sr = np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(roll / 2), np.sin(roll / 2)
    qw = cr * cp * cy + sr * sp * sy
    qx = sr * cp * cy - cr * sp * sy
    qy = cr * sp * cy + sr * cp * sy
    qz = cr * cp * sy - sr * sp * cy
    return generate_tmat((tx, ty, tz), (qx, qy, qz, qw))


def make_poses(start, count, seed=99):
    return np.stack([make_pose(start + i, seed) for i in range(count)]).astype(np.float32)
