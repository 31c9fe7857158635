"""ctypes loader of libo3dr.so.  Fails loudly when the HIP library is missing: no fallback."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "libo3dr.so")

POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")])

OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_NOT_CONFIGURED, ERR_ALLOC, ERR_INTERNAL, ERR_PEER = -1, -2, -3, -4, -5, -6, -7, -8
MEM_HOST, MEM_DEVICE = 0, 1
STATUS_VOXEL_OVERFLOW = 1
STATUS_INTERNAL = 0x80000000
K_COUNT, K_REPROJECT, K_KEYGEN, K_SORT_HIST, K_SORT_SCATTER, K_SEGMENT, K_CENTROID, K_OTHER, K_CENTROID_RUNS = range(9)
KERNEL_NAMES = ["reproject_count", "reproject_emit", "voxel_keys", "radix_hist", "radix_scatter", "run_segments",
                "centroid", "other", "centroid_runs"]


class O3drError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libo3dr error {code}: {text}")
        self.code = code


class ParamsStruct(C.Structure):
    _fields_ = [("min_disparity", C.c_double), ("voxel_size", C.c_double), ("bounding_box", C.c_int32),
                ("cutout_ratio", C.c_int32), ("jump_pixels", C.c_int32), ("min_points_per_voxel", C.c_uint32),
                ("dont_downsample", C.c_int32), ("sor_enable", C.c_int32), ("blur_kernel", C.c_int32), ("disparity_f64", C.c_int32)]


def lib_path():
    return _LIB


# every symbol include/o3dr.h declares: (name, restype, argtypes)
_vp, _i64, _i32, _u32, _f = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_float
_pi64, _pu32 = C.POINTER(C.c_int64), C.POINTER(C.c_uint32)
SYMBOLS = [
    ("o3dr_version", C.c_int, []),
    ("o3dr_last_error", C.c_char_p, []),
    ("o3dr_default_params", None, [C.POINTER(ParamsStruct)]),
    ("o3dr_ctx_create", C.c_int, [C.c_int, C.POINTER(_vp)]),
    ("o3dr_ctx_destroy", C.c_int, [_vp]),
    ("o3dr_ctx_set_stream", C.c_int, [_vp, _vp]),
    ("o3dr_ctx_synchronize", C.c_int, [_vp]),
    ("o3dr_set_camera", C.c_int, [_vp, _vp]),
    ("o3dr_set_params", C.c_int, [_vp, C.POINTER(ParamsStruct)]),
    ("o3dr_get_params", C.c_int, [_vp, C.POINTER(ParamsStruct)]),
    ("o3dr_create_single_img_pt_cloud", C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _i32, _vp, _i64, _pi64, _i32]),
    ("o3dr_max_points", _i64, [_vp, _i32, _i32]),
    ("o3dr_transform_pt_cloud", C.c_int, [_vp, _vp, _i64, _vp, _vp, _i32]),
    ("o3dr_reproject_transform", C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _i64, _pi64, _i32]),
    ("o3dr_voxel_grid", C.c_int, [_vp, _vp, _i64, _vp, _u32, _f, _vp, _i64, _pi64, _pu32, _i32]),
    ("o3dr_statistical_outlier_removal", C.c_int, [_vp, _vp, _i64, _vp, _i64, _pi64, _i32]),
    ("o3dr_bilateral_filter_u8", C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, C.c_double, C.c_double, _vp, _i64, _i32]),
    ("o3dr_disparity_variance", C.c_int, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _i32]),
    ("o3dr_downsample_pt_cloud", C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _pi64, _pu32, _i32]),
    ("o3dr_create_and_transform_pt_cloud", C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _i64, _pi64, _pu32, _i32]),
    ("o3dr_accumulate_frames", C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i32, _i32]),
    ("o3dr_accumulate_frames_kp", C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _i32]),
    ("o3dr_cloud_big_reserve", C.c_int, [_vp, _i64]),
    ("o3dr_cloud_big_reset", C.c_int, [_vp]),
    ("o3dr_cloud_big_size", C.c_int, [_vp, _pi64, _pu32]),
    ("o3dr_cloud_big_read", C.c_int, [_vp, _vp, _i64, _pi64, _i32]),
    ("o3dr_cloud_big_append", C.c_int, [_vp, _vp, _i64, _i32]),
    ("o3dr_cloud_big_transform", C.c_int, [_vp, _vp]),
    ("o3dr_finalize", C.c_int, [_vp, _vp, _i64, _pi64, _pu32, _i32]),
    ("o3dr_cloud_big_bbox", C.c_int, [_vp, _vp, _vp, _pi64]),
    ("o3dr_cloud_big_partition", C.c_int, [_vp, _vp, _vp, _i32, _pi64, _pu32]),
    ("o3dr_finalize_global", C.c_int, [_vp, _vp, _vp, _vp, _i64, _pi64, _pu32, _i32]),
    ("o3dr_cloud_big_view", C.c_int, [_vp, C.POINTER(_vp), _pi64]),
    ("o3dr_cloud_big_recv_buffer", C.c_int, [_vp, _i64, C.POINTER(_vp)]),
    ("o3dr_cloud_big_adopt", C.c_int, [_vp, _i64]),
    ("o3dr_cloud_big_header_dev", C.c_int, [_vp, _vp]),
    ("o3dr_cloud_big_assume_size", C.c_int, [_vp, _i64]),
    ("o3dr_cloud_big_partition_dev", C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    ("o3dr_cloud_big_slice_counts_dev", C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    ("o3dr_cloud_big_place_slices", C.c_int, [_vp, _i32, _i32, _vp, _i64, _i64, _pi64]),
    ("o3dr_cloud_big_set_size", C.c_int, [_vp, _i64]),
    ("o3dr_cloud_big_raw_view", C.c_int, [_vp, C.POINTER(_vp), _pi64]),
    ("o3dr_merge_partitioned", C.c_int, [_vp, _vp, _i32, _vp, _i64, _pi64, _pi64, C.POINTER(C.c_uint32), _i32]),
    ("o3dr_merge_partitioned_stats", C.c_int, [_vp, _pi64]),
    ("o3dr_cloud_big_capacity", C.c_int, [_vp, _pi64, _pi64]),
    ("o3dr_comm_init_all", C.c_int, [_i32, _vp, _vp]),
    ("o3dr_comm_destroy", C.c_int, [_vp]),
    ("o3dr_host_register", C.c_int, [_vp, _i64]),
    ("o3dr_host_unregister", C.c_int, [_vp]),
    ("o3dr_profile_enable", C.c_int, [_vp, _i32, _i32]),
    ("o3dr_profile_read", C.c_int, [_vp, _i32, C.POINTER(C.c_double), _pi64]),
    ("o3dr_profile_reset", C.c_int, [_vp]),
    ("o3dr_profile_stats", C.c_int, [_vp, _pi64]),
    ("o3dr_test_corrupt_next_gather", C.c_int, [_vp]),
    ("o3dr_test_sor_distances", C.c_int, [_vp, _vp, _i64]),
    ("o3dr_test_local_comm_create", C.c_int, [_i32, C.POINTER(_vp)]),
    ("o3dr_test_local_comm_destroy", C.c_int, [_vp]),
    ("o3dr_test_merge_partitioned_local", C.c_int, [_vp, _vp, _i32, _i32, _vp, _i64, _pi64, _pi64, C.POINTER(C.c_uint32), _i32]),
    ("o3dr_test_fail_at", C.c_int, [_vp, _i32]),
    ("o3dr_device_info", C.c_int, [_vp, C.c_char_p, _i32, C.POINTER(_i32), _pi64]),
]

_lib = None


def load_library():
    """dlopen libo3dr.so and bind every exported symbol; raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise O3drError(ERR_NO_DEVICE, f"{_LIB} is not built: run `python __graft_entry__.py` or `make` "
                                           "(there is no CPU fallback)")
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  A process must hold ONE HIP/HSA
        # runtime, so when torch is installed it is imported first: libo3dr's DT_NEEDED
        # libamdhip64.so.7 then binds to the copy torch already loaded instead of /opt/rocm's
        # (loading both leaves the second one with "No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_LIB)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != OK:
        raise O3drError(code, load_library().o3dr_last_error().decode(errors="replace"))
