"""Python mirror of the reference's operator surface for the hot path.

`Context` methods carry the reference's member-function names (pose.h:198,199,216,231):
createSingleImgPtCloud, transformPtCloud, downsamplePtCloud, createAndTransformPtCloud — plus the
fan-out/accumulate loop (pose.cpp:365-434) and the final merge (pose.cpp:527-532).  All compute
happens in libo3dr.so; numpy arrays are staged by the library, torch CUDA tensors are passed as
HBM pointers (torch is only device memory + streams here).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from ._lib import POINT


@dataclass
class Params:
    """The `Pose` members the hot path reads (pose.h:93-126), reference defaults."""
    min_disparity: float = 64.0
    voxel_size: float = 0.1
    bounding_box: int = 20
    cutout_ratio: int = 8
    jump_pixels: int = 10
    min_points_per_voxel: int = 1
    dont_downsample: bool = False
    sor_enable: bool = True      # the reference always runs it when !combined && jump_pixels > 0 (pose_functions.cpp:1673)
    blur_kernel: int = 1
    disparity_f64: bool = False  # --use_segment_labels: disparity images are float64

    def to_struct(self):
        return L.ParamsStruct(float(self.min_disparity), float(self.voxel_size), int(self.bounding_box),
                              int(self.cutout_ratio), int(self.jump_pixels), int(self.min_points_per_voxel),
                              int(bool(self.dont_downsample)), int(bool(self.sor_enable)), int(self.blur_kernel),
                              int(bool(self.disparity_f64)))


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    """(address, MEM kind, keepalive) of a numpy array or a torch tensor."""
    if _is_torch(x):
        assert x.is_contiguous()
        return x.data_ptr(), (L.MEM_DEVICE if x.is_cuda else L.MEM_HOST), x
    return x.ctypes.data, L.MEM_HOST, x


class Context:
    def __init__(self, device=0, Q=None, params=None, stream=None):
        self._lib = L.load_library()
        h = C.c_void_p()
        L.check(self._lib.o3dr_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.stream_raw = 0
        self.params = Params()
        if Q is not None:
            self.set_camera(Q)
        if params is not None:
            self.set_params(params)
        if stream is not None:
            self.set_stream(stream)

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.o3dr_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- configuration ----------------------------------------------------------------------------
    def set_camera(self, Q):
        Q = np.ascontiguousarray(Q, np.float64).reshape(16)
        L.check(self._lib.o3dr_set_camera(self._h, Q.ctypes.data))

    def set_params(self, params):
        s = params.to_struct()
        L.check(self._lib.o3dr_set_params(self._h, C.byref(s)))
        self.params = params

    def set_stream(self, stream):
        """stream: a torch.cuda.Stream, a raw hipStream_t integer, or None for the context's own."""
        raw = getattr(stream, "cuda_stream", stream)
        L.check(self._lib.o3dr_ctx_set_stream(self._h, C.c_void_p(raw or 0)))
        self.stream_raw = int(raw or 0)  # 0: the context's own stream (dist.py orders collectives against it)

    def synchronize(self):
        L.check(self._lib.o3dr_ctx_synchronize(self._h))

    def max_points(self, rows, cols):
        return int(self._lib.o3dr_max_points(self._h, rows, cols))

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, mem = C.c_int32(0), C.c_int64(0)
        L.check(self._lib.o3dr_device_info(self._h, name, 256, C.byref(cu), C.byref(mem)))
        return name.value.decode(), cu.value, mem.value

    # -- helpers ----------------------------------------------------------------------------------
    @staticmethod
    def _images(disp, bgr):
        if _is_torch(disp):
            rows, cols = disp.shape
            assert tuple(bgr.shape) == (rows, cols, 3)
            return rows, cols, disp.stride(0) * disp.element_size(), bgr.stride(0)
        rows, cols = disp.shape
        # u8 disparities, or float64 with Params.disparity_f64 (--use_segment_labels)
        assert disp.dtype in (np.uint8, np.float64) and bgr.dtype == np.uint8 and bgr.shape == (rows, cols, 3)
        assert disp.strides[1] == disp.itemsize and bgr.strides[2] == 1 and bgr.strides[1] == 3
        return rows, cols, disp.strides[0], bgr.strides[0]

    def _alloc_out(self, n, like):
        if _is_torch(like) and like.is_cuda:
            import torch
            return torch.empty((max(n, 1), 4), dtype=torch.int32, device=like.device)
        return np.empty(max(n, 1), POINT)

    @staticmethod
    def _kp(kp_xy, like):
        if kp_xy is None:
            return 0, 0, None
        if _is_torch(kp_xy):
            return kp_xy.data_ptr(), int(kp_xy.shape[0]), kp_xy
        kp = np.ascontiguousarray(kp_xy, np.float32).reshape(-1, 2)
        if _is_torch(like) and like.is_cuda:
            import torch
            t = torch.from_numpy(kp).to(like.device)
            return t.data_ptr(), len(kp), t
        return kp.ctypes.data, len(kp), kp

    @staticmethod
    def _T(T):
        return np.ascontiguousarray(T, np.float32).reshape(16)

    def _frame(self, fn_name, disp, bgr, T, kp_xy, with_status):
        rows, cols, dp, bp = self._images(disp, bgr)
        pd, mem, _k1 = _ptr(disp)
        pb, mem2, _k2 = _ptr(bgr)
        assert mem == mem2
        kp_ptr, n_kp, _k3 = self._kp(kp_xy, disp)
        cap = self.max_points(rows, cols) + n_kp
        out = self._alloc_out(cap, disp)
        po, _, _k4 = _ptr(out)
        n = C.c_int64(0)
        st = C.c_uint32(0)
        fn = getattr(self._lib, fn_name)
        args = [self._h, pd, dp, pb, bp, rows, cols]
        if T is not None:
            Tn = self._T(T)
            args.append(Tn.ctypes.data)
        args += [kp_ptr, n_kp, po, cap, C.byref(n)]
        if with_status:
            args.append(C.byref(st))
        args.append(mem)
        L.check(fn(*args))
        return out[: n.value], st.value

    # -- the reference's four entry points ---------------------------------------------------------
    def createSingleImgPtCloud(self, disp, bgr, kp_xy=None):
        """pose.h:198 / pose_functions.cpp:1030-1134 (camera-frame cloud of one frame)."""
        return self._frame("o3dr_create_single_img_pt_cloud", disp, bgr, None, kp_xy, False)[0]

    def transformPtCloud(self, pts, T):
        """pose.h:199 / pose_functions.cpp:1358-1362."""
        Tn = self._T(T)
        if not _is_torch(pts):
            pts = np.ascontiguousarray(pts, POINT)
        pi, mem, _k = _ptr(pts)
        n = int(pts.shape[0])
        out = self._alloc_out(n, pts)
        po, _, _k2 = _ptr(out)
        L.check(self._lib.o3dr_transform_pt_cloud(self._h, pi, n, Tn.ctypes.data, po, mem))
        return out[:n]

    def reprojectTransform(self, disp, bgr, T, kp_xy=None):
        """A1+A2 fused: createSingleImgPtCloud followed by transformPtCloud (pose.cpp:603-607)."""
        return self._frame("o3dr_reproject_transform", disp, bgr, T, kp_xy, False)[0]

    def downsamplePtCloud(self, pts, combinedPtCloud, return_status=False):
        """pose.h:216 / pose_functions.cpp:1654-1709."""
        if not _is_torch(pts):
            pts = np.ascontiguousarray(pts, POINT)
        pi, mem, _k = _ptr(pts)
        n_in = int(pts.shape[0])
        out = self._alloc_out(n_in, pts)
        po, _, _k2 = _ptr(out)
        n = C.c_int64(0)
        st = C.c_uint32(0)
        L.check(self._lib.o3dr_downsample_pt_cloud(self._h, pi, n_in, int(bool(combinedPtCloud)), po, max(n_in, 1),
                                                  C.byref(n), C.byref(st), mem))
        return (out[: n.value], st.value) if return_status else out[: n.value]

    def bilateralFilter(self, img, d, sigma_color, sigma_space):
        """cv::bilateralFilter on a u8 image (pose_functions.cpp:1044); numpy in/out or torch CUDA in/out."""
        if _is_torch(img):
            import torch
            assert img.dtype == torch.uint8 and img.dim() == 2 and img.stride(1) == 1
            out = torch.empty_like(img, memory_format=torch.contiguous_format)
            L.check(self._lib.o3dr_bilateral_filter_u8(self._h, img.data_ptr(), img.stride(0), img.shape[0], img.shape[1], int(d),
                                                       float(sigma_color), float(sigma_space), out.data_ptr(), out.stride(0),
                                                       L.MEM_DEVICE if img.is_cuda else L.MEM_HOST))
            if img.is_cuda:
                L.check(self._lib.o3dr_ctx_synchronize(self._h))  # the library's stream is not torch's
            return out
        img = np.asarray(img, np.uint8)
        assert img.ndim == 2 and img.strides[1] == 1
        out = np.empty(img.shape, np.uint8)
        L.check(self._lib.o3dr_bilateral_filter_u8(self._h, img.ctypes.data, img.strides[0], img.shape[0], img.shape[1], int(d),
                                                   float(sigma_color), float(sigma_space), out.ctypes.data, out.strides[0],
                                                   L.MEM_HOST))
        return out

    def disparityVariance(self, disp):
        """Pose::getVariance per frame (pose_functions.cpp:1007-1028) of a [F,H,W] or [H,W] u8 stack -> float64 [F]"""
        if disp.ndim == 2:
            disp = disp[None]
        F, rows, cols = (int(v) for v in disp.shape)
        if _is_torch(disp):
            assert disp.stride(2) == 1
            ptr, mem, fs, pitch = disp.data_ptr(), (L.MEM_DEVICE if disp.is_cuda else L.MEM_HOST), disp.stride(0), disp.stride(1)
        else:
            disp = np.asarray(disp, np.uint8)
            assert disp.strides[2] == 1
            ptr, mem, fs, pitch = disp.ctypes.data, L.MEM_HOST, disp.strides[0], disp.strides[1]
        out = np.zeros(F, np.float64)
        L.check(self._lib.o3dr_disparity_variance(self._h, ptr, pitch, fs, rows, cols, F, out.ctypes.data, mem))
        return out

    def statisticalOutlierRemoval(self, pts):
        """pcl::StatisticalOutlierRemoval, mean_k 50, 1 sigma (pose_functions.cpp:1679-1684)."""
        if not _is_torch(pts):
            pts = np.ascontiguousarray(pts, POINT)
        pi, mem, _k = _ptr(pts)
        n_in = int(pts.shape[0])
        out = self._alloc_out(n_in, pts)
        po, _, _k2 = _ptr(out)
        n = C.c_int64(0)
        L.check(self._lib.o3dr_statistical_outlier_removal(self._h, pi, n_in, po, max(n_in, 1), C.byref(n), mem))
        return out[: n.value]

    def voxelGrid(self, pts, leaf, min_points=0, z_offset=0.0, return_status=False):
        """pcl::VoxelGrid<PointXYZRGB> as the reference configures it (pose_functions.cpp:1689-1700)."""
        if not _is_torch(pts):
            pts = np.ascontiguousarray(pts, POINT)
        leaf = np.ascontiguousarray(leaf, np.float32).reshape(3)
        pi, mem, _k = _ptr(pts)
        n_in = int(pts.shape[0])
        out = self._alloc_out(n_in, pts)
        po, _, _k2 = _ptr(out)
        n = C.c_int64(0)
        st = C.c_uint32(0)
        L.check(self._lib.o3dr_voxel_grid(self._h, pi, n_in, leaf.ctypes.data, int(min_points), float(z_offset), po,
                                         max(n_in, 1), C.byref(n), C.byref(st), mem))
        return (out[: n.value], st.value) if return_status else out[: n.value]

    def createAndTransformPtCloud(self, disp, bgr, T, kp_xy=None, return_status=False):
        """pose.h:231 / pose.cpp:596-636."""
        out, st = self._frame("o3dr_create_and_transform_pt_cloud", disp, bgr, T, kp_xy, True)
        return (out, st) if return_status else out

    # -- fan-out / accumulate / final merge ---------------------------------------------------------
    def accumulateFrames(self, disp, bgr, poses, keypoints=None):
        """pose.cpp:365-434 for a stack of frames: disp [F,H,W] u8, bgr [F,H,W,3] u8, poses [F,4,4] f32;
        keypoints: optional list of F arrays [n_f,2] of (x,y) floats (KeyPoint::pt), used iff jump_pixels != 1."""
        F, rows, cols = disp.shape
        if _is_torch(disp):
            assert disp.is_contiguous() and bgr.is_contiguous() and poses.is_contiguous()
            es = disp.element_size()  # 1, or 8 with Params.disparity_f64
            dfs, dp, bfs, bp = disp.stride(0) * es, disp.stride(1) * es, bgr.stride(0), bgr.stride(1)
            assert poses.dtype.__str__() == "torch.float32" and poses.numel() == F * 16
        else:
            disp = np.ascontiguousarray(disp, np.float64 if np.asarray(disp).dtype == np.float64 else np.uint8)
            bgr = np.ascontiguousarray(bgr, np.uint8)
            poses = np.ascontiguousarray(poses, np.float32).reshape(F, 16)
            dfs, dp, bfs, bp = disp.strides[0], disp.strides[1], bgr.strides[0], bgr.strides[1]
        pd, mem, _k1 = _ptr(disp)
        pb, mem2, _k2 = _ptr(bgr)
        pp, mem3, _k3 = _ptr(poses)
        assert mem == mem2 == mem3
        if keypoints is not None:
            assert len(keypoints) == F
            kps = [np.ascontiguousarray(k, np.float32).reshape(-1, 2) for k in keypoints]
            offs = np.zeros(F + 1, np.int64)
            offs[1:] = np.cumsum([len(k) for k in kps])
            kp = np.concatenate(kps) if offs[-1] else np.zeros((0, 2), np.float32)
            if mem == L.MEM_DEVICE:
                import torch
                kp_t = torch.from_numpy(kp).to(disp.device)
                kp_ptr = kp_t.data_ptr()
            else:
                kp_ptr = kp.ctypes.data
            L.check(self._lib.o3dr_accumulate_frames_kp(self._h, pd, dfs, dp, pb, bfs, bp, rows, cols, pp, F, kp_ptr,
                                                        offs.ctypes.data, mem))
            if mem == L.MEM_DEVICE:
                L.check(self._lib.o3dr_ctx_synchronize(self._h))  # kp_t must outlive the launches
            return
        L.check(self._lib.o3dr_accumulate_frames(self._h, pd, dfs, dp, pb, bfs, bp, rows, cols, pp, F, mem))

    def cloudBigReserve(self, n_points):
        L.check(self._lib.o3dr_cloud_big_reserve(self._h, int(n_points)))

    def cloudBigReset(self):
        L.check(self._lib.o3dr_cloud_big_reset(self._h))

    def cloudBigSize(self):
        n = C.c_int64(0)
        st = C.c_uint32(0)
        L.check(self._lib.o3dr_cloud_big_size(self._h, C.byref(n), C.byref(st)))
        return n.value, st.value

    def cloudBigRead(self, device=None):
        n, _ = self.cloudBigSize()
        if device is not None:
            import torch
            out = torch.empty((max(n, 1), 4), dtype=torch.int32, device=device)
        else:
            out = np.empty(max(n, 1), POINT)
        po, mem, _k = _ptr(out)
        m = C.c_int64(0)
        L.check(self._lib.o3dr_cloud_big_read(self._h, po, max(n, 1), C.byref(m), mem))
        return out[: m.value]

    def cloudBigAppend(self, pts):
        if not _is_torch(pts):
            pts = np.ascontiguousarray(pts, POINT)
        pi, mem, _k = _ptr(pts)
        L.check(self._lib.o3dr_cloud_big_append(self._h, pi, int(pts.shape[0]), mem))

    def cloudBigTransform(self, T):
        Tn = self._T(T)
        L.check(self._lib.o3dr_cloud_big_transform(self._h, Tn.ctypes.data))

    # zero-copy views for the multi-GPU exchange (torch tensors over the library's HBM buffers)
    def cloudBigView(self):
        """[n,4] int32 CUDA tensor aliasing cloud_big (no copy); valid until the cloud is modified"""
        ptr = C.c_void_p()
        n = C.c_int64(0)
        L.check(self._lib.o3dr_cloud_big_view(self._h, C.byref(ptr), C.byref(n)))
        return _device_tensor(ptr.value, n.value, self.device)

    def cloudBigRecvBuffer(self, n_points):
        """[n_points,4] int32 CUDA tensor over the library's receive buffer"""
        ptr = C.c_void_p()
        L.check(self._lib.o3dr_cloud_big_recv_buffer(self._h, int(n_points), C.byref(ptr)))
        return _device_tensor(ptr.value, int(n_points), self.device)

    def cloudBigAdopt(self, n_points):
        """the first n_points of the receive buffer become cloud_big"""
        L.check(self._lib.o3dr_cloud_big_adopt(self._h, int(n_points)))

    def cloudBigHeaderDev(self):
        """this rank's 32-byte exchange header {min xyz, max xyz (f32), count (i64)} as a uint8 CUDA tensor; asynchronous"""
        import torch
        hdr = torch.empty(32, dtype=torch.uint8, device=torch.device("cuda", self.device))
        self._order_after_torch()
        L.check(self._lib.o3dr_cloud_big_header_dev(self._h, hdr.data_ptr()))
        return hdr

    def _order_after_torch(self):
        """The library writes on the context's stream into a block torch's allocator handed out on torch's current
        stream: when the two differ, the block's previous use on torch's stream must finish first."""
        import torch
        cur = torch.cuda.current_stream(self.device)
        if self.stream_raw != cur.cuda_stream:
            cur.synchronize()

    def cloudBigPartitionDev(self, hdrs, n_parts):
        """hdrs: the all-gathered headers ([world*32] uint8, CUDA).  Stable reorder of cloud_big by index slice of the
        combined grid over the box the headers span; -> int64 CUDA tensor [n_parts + 1]: slice counts, then the status
        word.  Asynchronous (no host round trip)."""
        import torch
        assert hdrs.is_cuda and hdrs.is_contiguous() and hdrs.numel() % 32 == 0
        counts = torch.empty(n_parts + 1, dtype=torch.int64, device=hdrs.device)
        self._order_after_torch()
        L.check(self._lib.o3dr_cloud_big_partition_dev(self._h, hdrs.data_ptr(), hdrs.numel() // 32, int(n_parts), counts.data_ptr()))
        self._keep = hdrs  # (the launches read it)
        return counts

    def cloudBigSliceCountsDev(self, hdrs, n_parts):
        """hdrs: the all-gathered headers ([world*32] uint8, CUDA).  Slice sizes of cloud_big over the box the headers span,
        NOTHING MOVED: -> int64 CUDA tensor [n_parts + 1] (sizes, then the status word).  Asynchronous."""
        import torch
        assert hdrs.is_cuda and hdrs.is_contiguous() and hdrs.numel() % 32 == 0
        counts = torch.empty(n_parts + 1, dtype=torch.int64, device=hdrs.device)
        self._order_after_torch()
        L.check(self._lib.o3dr_cloud_big_slice_counts_dev(self._h, hdrs.data_ptr(), hdrs.numel() // 32, int(n_parts), counts.data_ptr()))
        self._keep = hdrs  # (the launches read it)
        return counts

    def cloudBigPlaceSlices(self, own_part, counts, n_before, n_after):
        """lay cloud_big out as [n_before free | own slice | n_after free | leaving slices]; -> offset of the leaving slices"""
        arr = (C.c_int64 * len(counts))(*[int(v) for v in counts])
        off = C.c_int64(0)
        L.check(self._lib.o3dr_cloud_big_place_slices(self._h, len(counts), int(own_part), arr, int(n_before), int(n_after), C.byref(off)))
        return off.value

    def cloudBigSetSize(self, n_points):
        """the first n_points of the cloud buffer are the cloud (after the exchange filled the gaps); stream-ordered"""
        L.check(self._lib.o3dr_cloud_big_set_size(self._h, int(n_points)))

    def cloudBigRawView(self):
        """[capacity,4] int32 CUDA tensor over the whole cloud buffer (valid until the next call that may reallocate it)"""
        ptr = C.c_void_p()
        cap = C.c_int64(0)
        L.check(self._lib.o3dr_cloud_big_raw_view(self._h, C.byref(ptr), C.byref(cap)))
        return _device_tensor(ptr.value, cap.value, self.device)

    def cloudBigCapacity(self):
        """(points cloud_big holds, points the receive buffer holds) without reallocating"""
        a, b = C.c_int64(0), C.c_int64(0)
        L.check(self._lib.o3dr_cloud_big_capacity(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def mergePartitionedStats(self):
        """what this context's last o3dr_merge_partitioned moved (include/o3dr.h)"""
        out = (C.c_int64 * 8)()
        L.check(self._lib.o3dr_merge_partitioned_stats(self._h, out))
        keys = ("points_local", "points_sent_off_rank", "points_received_off_rank", "bytes_sent", "bytes_received",
                "points_into_merge", "agreement_rounds", "points_all_ranks")
        return dict(zip(keys, (int(v) for v in out)))

    def cloudBigAssumeSize(self, n_points):
        """the caller read this rank's header back: cloud_big holds exactly n_points (saves the library its own round trips)"""
        L.check(self._lib.o3dr_cloud_big_assume_size(self._h, int(n_points)))

    def cloudBigBBox(self):
        """(min xyz, max xyz, count) of cloud_big; (+inf, -inf, 0) when empty"""
        mn = np.empty(3, np.float32)
        mx = np.empty(3, np.float32)
        n = C.c_int64(0)
        L.check(self._lib.o3dr_cloud_big_bbox(self._h, mn.ctypes.data, mx.ctypes.data, C.byref(n)))
        return mn, mx, n.value

    def cloudBigPartition(self, gmin, gmax, n_parts):
        """stable reorder of cloud_big by index slice of the combined grid over [gmin, gmax]; -> (counts, status)"""
        gmin = np.ascontiguousarray(gmin, np.float32)
        gmax = np.ascontiguousarray(gmax, np.float32)
        counts = (C.c_int64 * n_parts)()
        st = C.c_uint32(0)
        L.check(self._lib.o3dr_cloud_big_partition(self._h, gmin.ctypes.data, gmax.ctypes.data, n_parts, counts, C.byref(st)))
        return [int(v) for v in counts], st.value

    def finalize(self, device=None, return_status=False, gmin=None, gmax=None, n_hint=None):
        """cloud_small = downsamplePtCloud(cloud_big, true) (pose.cpp:530); with gmin/gmax the grid is laid
        over that (global) bounding box instead of cloud_big's own (multi-GPU merge).  n_hint: the caller knows
        cloud_big's size (after an adopt): the output is sized without asking the device."""
        n = int(n_hint) if n_hint is not None else self.cloudBigSize()[0]
        if device is not None:
            import torch
            out = torch.empty((max(n, 1), 4), dtype=torch.int32, device=device)
        else:
            out = np.empty(max(n, 1), POINT)
        po, mem, _k = _ptr(out)
        m = C.c_int64(0)
        st = C.c_uint32(0)
        if gmin is None:
            L.check(self._lib.o3dr_finalize(self._h, po, max(n, 1), C.byref(m), C.byref(st), mem))
        else:
            gmin = np.ascontiguousarray(gmin, np.float32)
            gmax = np.ascontiguousarray(gmax, np.float32)
            L.check(self._lib.o3dr_finalize_global(self._h, gmin.ctypes.data, gmax.ctypes.data, po, max(n, 1), C.byref(m),
                                                  C.byref(st), mem))
        return (out[: m.value], st.value) if return_status else out[: m.value]

    # -- measurement hooks --------------------------------------------------------------------------
    def profileEnable(self, kernel_id=-1, enable=True):
        L.check(self._lib.o3dr_profile_enable(self._h, int(kernel_id), int(bool(enable))))

    def profileReset(self):
        L.check(self._lib.o3dr_profile_reset(self._h))

    def profileStats(self):
        """(sort record-passes, voxel-grid points in, voxel-grid points out) since profileReset()"""
        return self.profileStats4()[:3]

    def profileStats4(self):
        """profileStats() + the number of frames that took the pixel-window path"""
        return self.profileStatsAll()[:4]

    def profileStatsAll(self):
        """all eight counters of o3dr_profile_stats (include/o3dr.h)"""
        out = (C.c_int64 * 8)()
        L.check(self._lib.o3dr_profile_stats(self._h, out))
        return tuple(int(v) for v in out)

    def profileRead(self, kernel_id):
        ms = C.c_double(0)
        n = C.c_int64(0)
        L.check(self._lib.o3dr_profile_read(self._h, int(kernel_id), C.byref(ms), C.byref(n)))
        return ms.value, n.value


class _DevMem:
    """minimal __cuda_array_interface__ holder so torch can alias library-owned HBM"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n, 4), "typestr": "<i4", "data": (ptr, False), "version": 2}


def _device_tensor(ptr, n, device_index):
    import torch
    if n == 0 or not ptr:
        return torch.empty((0, 4), dtype=torch.int32, device=torch.device("cuda", device_index))
    return torch.as_tensor(_DevMem(ptr, n), device=torch.device("cuda", device_index))


def points_from_torch(t):
    """[N,4] int32 CUDA/CPU tensor -> numpy POINT array (host copy)."""
    return t.detach().cpu().numpy().view(POINT).reshape(-1)
