"""ctypes binding of the CPU oracle (oracle/_build/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker.  Nothing under online_3d_reconstruction_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liborc.so")

POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")])
ORDER_STABLE = 0
ORDER_STDSORT = 1
STATUS_VOXEL_OVERFLOW = 1


def build(force=False):
    """Compile the oracle with gcc/g++ (seconds)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i64, i32, u32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_double
        L.orc_create_single_img_pt_cloud.restype = i64
        L.orc_create_single_img_pt_cloud.argtypes = [vp, i64, vp, i64, i32, i32, vp, i32, i32, dbl, i32,
                                                     vp, i32, vp]
        L.orc_create_single_img_pt_cloud_f64.restype = i64
        L.orc_create_single_img_pt_cloud_f64.argtypes = L.orc_create_single_img_pt_cloud.argtypes
        L.orc_transform_pt_cloud.restype = None
        L.orc_transform_pt_cloud.argtypes = [vp, i64, vp, vp]
        L.orc_voxel_grid.restype = i64
        L.orc_voxel_grid.argtypes = [vp, i64, vp, u32, i32, vp, vp]
        L.orc_downsample_pt_cloud.restype = i64
        L.orc_downsample_pt_cloud.argtypes = [vp, i64, dbl, i32, u32, i32, vp, vp]
        L.orc_create_and_transform_pt_cloud.restype = i64
        L.orc_create_and_transform_pt_cloud.argtypes = [vp, i64, vp, i64, i32, i32, vp, i32, i32, dbl, i32,
                                                        vp, i32, vp, dbl, i32, i32, vp, vp, vp]
        L.orc_statistical_outlier_removal.restype = i64
        L.orc_statistical_outlier_removal.argtypes = [vp, i64, i32, dbl, vp, vp, i32]
        L.orc_run_frames.restype = i64
        L.orc_run_frames.argtypes = [vp, i64, i64, vp, i64, i64, i32, i32, vp, i32, i32, dbl, i32, vp, i32, dbl, u32, i32, i32,
                                     vp, vp, vp]
        L.orc_run_frames_order.restype = i64
        L.orc_run_frames_order.argtypes = [vp, i64, i64, vp, i64, i64, i32, i32, vp, i32, i32, dbl, i32, vp, i32, dbl, u32, i32, i32,
                                           i32, vp, vp, vp]
        L.orc_voxel_keys.restype = u32
        L.orc_voxel_keys.argtypes = [vp, i64, vp, vp, vp, vp]
        L.orc_bilateral_filter_u8.restype = None
        L.orc_bilateral_filter_u8.argtypes = [vp, i64, i32, i32, i32, dbl, dbl, i32, vp, i64]
        L.orc_bilateral_filter_u8c3.restype = None
        L.orc_bilateral_filter_u8c3.argtypes = [vp, i64, i32, i32, i32, dbl, dbl, i32, vp, i64]
        L.orc_disparity_variance.restype = dbl
        L.orc_disparity_variance.argtypes = [vp, i64, i32, i32, i32, i32, dbl]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def grid_shape(rows, cols, bounding_box=20, cutout_ratio=8, jump_pixels=1):
    """(Ny, Nx, cols_start_aft_cutout) of the grid pass, pose_functions.cpp:1094-1096,638."""
    cs = int(cols / cutout_ratio)
    if jump_pixels <= 0:
        return 0, 0, cs
    ny = max(0, -(-(rows - 2 * bounding_box) // jump_pixels))
    nx = max(0, -(-(cols - bounding_box - cs) // jump_pixels))
    return ny, nx, cs


def _kp(kp_xy):
    if kp_xy is None:
        return np.zeros((0, 2), np.float32)
    return np.ascontiguousarray(kp_xy, np.float32).reshape(-1, 2)


def create_single_img_pt_cloud(disp, bgr, Q, bounding_box=20, cutout_ratio=8, min_disparity=64.0,
                               jump_pixels=1, kp_xy=None):
    """disp: u8 [H,W], or float64 [H,W] for the --use_segment_labels form (CV_64F disparities)"""
    f64 = np.asarray(disp).dtype == np.float64
    disp = np.ascontiguousarray(disp, np.float64 if f64 else np.uint8)
    bgr = np.ascontiguousarray(bgr, np.uint8)
    rows, cols = disp.shape
    assert bgr.shape == (rows, cols, 3)
    Q = np.ascontiguousarray(Q, np.float64).reshape(16)
    kp = _kp(kp_xy)
    ny, nx, cs = grid_shape(rows, cols, bounding_box, cutout_ratio, jump_pixels)
    out = np.empty(ny * nx + len(kp), POINT)
    fn = lib().orc_create_single_img_pt_cloud_f64 if f64 else lib().orc_create_single_img_pt_cloud
    n = fn(_p(disp), disp.strides[0], _p(bgr), bgr.strides[0], rows, cols, _p(Q), bounding_box, cs, float(min_disparity),
           jump_pixels, _p(kp), len(kp), _p(out))
    return out[:n].copy()


def transform_pt_cloud(pts, T):
    pts = np.ascontiguousarray(pts, POINT)
    T = np.ascontiguousarray(T, np.float32).reshape(16)
    out = np.empty_like(pts)
    lib().orc_transform_pt_cloud(_p(pts), len(pts), _p(T), _p(out))
    return out


def voxel_grid(pts, leaf, min_points=0, order=ORDER_STABLE):
    pts = np.ascontiguousarray(pts, POINT)
    leaf = np.ascontiguousarray(leaf, np.float32).reshape(3)
    out = np.empty(max(len(pts), 1), POINT)
    st = C.c_uint32(0)
    n = lib().orc_voxel_grid(_p(pts), len(pts), _p(leaf), min_points, order, _p(out), C.byref(st))
    return out[:n].copy(), st.value


def downsample_pt_cloud(pts, voxel_size, combined, min_points_per_voxel=1, order=ORDER_STABLE):
    pts = np.ascontiguousarray(pts, POINT)
    out = np.empty(max(len(pts), 1), POINT)
    st = C.c_uint32(0)
    n = lib().orc_downsample_pt_cloud(_p(pts), len(pts), float(voxel_size), int(bool(combined)),
                                      min_points_per_voxel, order, _p(out), C.byref(st))
    return out[:n].copy(), st.value


def create_and_transform_pt_cloud(disp, bgr, Q, T, voxel_size, bounding_box=20, cutout_ratio=8,
                                  min_disparity=64.0, jump_pixels=1, kp_xy=None, dont_downsample=False,
                                  order=ORDER_STABLE):
    disp = np.ascontiguousarray(disp, np.uint8)
    bgr = np.ascontiguousarray(bgr, np.uint8)
    rows, cols = disp.shape
    Q = np.ascontiguousarray(Q, np.float64).reshape(16)
    T = np.ascontiguousarray(T, np.float32).reshape(16)
    kp = _kp(kp_xy)
    ny, nx, cs = grid_shape(rows, cols, bounding_box, cutout_ratio, jump_pixels)
    cap = ny * nx + len(kp)
    scratch = np.empty(2 * max(cap, 1), POINT)
    out = np.empty(max(cap, 1), POINT)
    st = C.c_uint32(0)
    n = lib().orc_create_and_transform_pt_cloud(_p(disp), disp.strides[0], _p(bgr), bgr.strides[0], rows, cols,
                                                _p(Q), bounding_box, cs, float(min_disparity), jump_pixels,
                                                _p(kp), len(kp), _p(T), float(voxel_size),
                                                int(bool(dont_downsample)), order, _p(scratch), _p(out),
                                                C.byref(st))
    return out[:n].copy(), st.value


def voxel_keys(pts, leaf):
    pts = np.ascontiguousarray(pts, POINT)
    leaf = np.ascontiguousarray(leaf, np.float32).reshape(3)
    keys = np.empty(max(len(pts), 1), np.uint32)
    min_b = np.zeros(3, np.int32)
    div_b = np.zeros(3, np.int32)
    st = lib().orc_voxel_keys(_p(pts), len(pts), _p(leaf), _p(keys), _p(min_b), _p(div_b))
    return keys[:len(pts)], min_b, div_b, st


def statistical_outlier_removal(pts, mean_k=50, stddev_mul=1.0, brute=False):
    """(kept points, mean neighbour distance of every input point) — pose_functions.cpp:1679-1684"""
    pts = np.ascontiguousarray(pts, POINT)
    out = np.empty(max(len(pts), 1), POINT)
    dist = np.zeros(max(len(pts), 1), np.float32)
    n = lib().orc_statistical_outlier_removal(_p(pts), len(pts), mean_k, float(stddev_mul), _p(out), _p(dist), int(bool(brute)))
    return out[:n].copy(), dist[:len(pts)]


def run_frames(disp, bgr, Q, poses, voxel_size, jump_pixels=1, min_points_per_voxel=1, sor=False, threads=7,
               bounding_box=20, cutout_ratio=8, min_disparity=64.0, want_clouds=True, order=ORDER_STABLE):
    """A7 + final merge with the reference's thread fan-out, inside the C oracle (pthreads).
    Returns (cloud_big, merged) or just the merged count when want_clouds is False (timing).
    order: summation order of every voxel grid (ORDER_STDSORT = the reference's own std::sort order)."""
    disp = np.ascontiguousarray(disp, np.uint8)
    bgr = np.ascontiguousarray(bgr, np.uint8)
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    F, rows, cols = disp.shape
    Q = np.ascontiguousarray(Q, np.float64).reshape(16)
    ny, nx, cs = grid_shape(rows, cols, bounding_box, cutout_ratio, jump_pixels)
    n_big = C.c_int64(0)
    big = merged = None
    if want_clouds:
        big = np.empty(max(F * ny * nx, 1), POINT)
        merged = np.empty(max(F * ny * nx, 1), POINT)
    m = lib().orc_run_frames_order(_p(disp), disp.strides[0], disp.strides[1], _p(bgr), bgr.strides[0], bgr.strides[1], rows,
                                   cols, _p(Q), bounding_box, cs, float(min_disparity), jump_pixels, _p(poses), F,
                                   float(voxel_size), min_points_per_voxel, int(bool(sor)), threads, int(order),
                                   _p(big) if want_clouds else None, C.byref(n_big), _p(merged) if want_clouds else None)
    if want_clouds:
        return big[: n_big.value].copy(), merged[:m].copy()
    return m


BILATERAL_SSE3 = 0
BILATERAL_SCALAR = 1


def bilateral_filter(disp, d, sigma_color, sigma_space, order=BILATERAL_SSE3):
    """cv::bilateralFilter on a u8 image (OpenCV 3.1 restatement) — pose_functions.cpp:1044"""
    disp = np.ascontiguousarray(disp, np.uint8)
    rows, cols = disp.shape
    out = np.empty_like(disp)
    lib().orc_bilateral_filter_u8(_p(disp), disp.strides[0], rows, cols, int(d), float(sigma_color), float(sigma_space),
                                  int(order), _p(out), out.strides[0])
    return out


def bilateral_filter_bgr(bgr, d, sigma_color, sigma_space, order=BILATERAL_SSE3):
    """cv::bilateralFilter on a CV_8UC3 image: the cn == 3 branch of the same OpenCV 3.1 invoker (the reference holds
    outputs of it: build/output/bilateralFiltered_15.png / _31.png)"""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    rows, cols, cn = bgr.shape
    assert cn == 3
    out = np.empty_like(bgr)
    lib().orc_bilateral_filter_u8c3(_p(bgr), bgr.strides[0], rows, cols, int(d), float(sigma_color), float(sigma_space),
                                    int(order), _p(out), out.strides[0])
    return out


def blur_disparity(disp, blur_kernel, order=BILATERAL_SSE3):
    """the call of pose_functions.cpp:1044: bilateralFilter(d=bk, sigmaColor=bk*2, sigmaSpace=bk/2 (integer division))"""
    return bilateral_filter(disp, blur_kernel, blur_kernel * 2, blur_kernel // 2, order)


def disparity_variance(disp, bounding_box=20, cutout_ratio=8, min_disparity=64.0):
    """Pose::getVariance(disp, false) — pose_functions.cpp:1007-1028"""
    disp = np.ascontiguousarray(disp, np.uint8)
    rows, cols = disp.shape
    return float(lib().orc_disparity_variance(_p(disp), disp.strides[0], rows, cols, bounding_box, int(cols / cutout_ratio),
                                              float(min_disparity)))
