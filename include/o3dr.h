/*
 * o3dr.h — C ABI of libo3dr, the MI355X (gfx950) implementation of the per-frame
 * reconstruction hot path of pk17r/online_3d_reconstruction:
 *
 *   disparity image [-> bilateral filter] -> 3D back-projection through Q -> rigid
 *   transform into the world frame [-> statistical outlier removal] -> voxel-grid
 *   downsample -> accumulate -> 2.5-D global merge.
 *
 * The reference has no FFI/plugin interface; the seam is four `Pose` member
 * functions plus the fan-out/accumulate loop around them.  Every entry point
 * below names the reference interface (file:line under the reference tree) it
 * replaces.  Plain pointers and sizes only: no C++, HIP or torch types.
 *
 * Conventions
 *   - every function returns O3DR_OK (0) or a negative O3DR_ERR_* code; on error
 *     every `n_out` is set to 0 ("output cloud left empty", pose.cpp:620-635).
 *   - `mem` says where ALL data pointers of that call live: O3DR_MEM_HOST
 *     (copied into HBM staging buffers with hipMemcpyAsync straight from the
 *     caller's pointers: page-locked caller memory transfers by DMA, pageable
 *     memory goes through the HIP runtime's own bounce buffers; the batched
 *     o3dr_accumulate_frames uploads batch k+1 on a second stream while batch
 *     k computes) or O3DR_MEM_DEVICE (HBM pointers of the context's device;
 *     nothing is copied).  Small
 *     parameter arrays (Q, poses of the single-frame calls, leaf) are always
 *     host pointers; the batched `o3dr_accumulate_frames` takes its pose array
 *     in `mem` like the images.
 *   - images are OpenCV-layout: disparity CV_8UC1 row-major with a byte pitch (CV_64F
 *     with o3dr_params.disparity_f64: pitch and frame stride stay in bytes),
 *     colour CV_8UC3 interleaved B,G,R with a byte pitch (pose_functions.cpp:526,548).
 *   - points are 16 bytes: x,y,z float + packed colour (a<<24|r<<16|g<<8|b), the
 *     same packing pose_functions.cpp:1120-1121 stores in PointXYZRGB::rgb.
 *   - 4x4 matrices are ROW-major float[16] (only the top three rows are read),
 *     Q is ROW-major double[16] (pose.h:128, cam13calib.yml:91-97).
 *   - a context is single-threaded; the reference's 7 concurrent callers
 *     (pose.cpp:392-413) each own a context.  Work is issued on the context's
 *     stream; calls with host outputs synchronise before returning, calls with
 *     device outputs that report a count synchronise only for that count.
 */
#ifndef O3DR_H
#define O3DR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define O3DR_VERSION 100 /* 0.1.0 */

/* return codes */
#define O3DR_OK                  0
#define O3DR_ERR_INVALID_ARG    -1
#define O3DR_ERR_NO_DEVICE      -2 /* no usable gfx950 device / HIP runtime failure at create */
#define O3DR_ERR_HIP            -3 /* a HIP call failed; o3dr_last_error() has the text */
#define O3DR_ERR_CAPACITY       -4 /* caller's output buffer is too small */
#define O3DR_ERR_NOT_CONFIGURED -5 /* o3dr_set_camera not called yet */
#define O3DR_ERR_ALLOC          -6
#define O3DR_ERR_INTERNAL       -7 /* a device-side consistency guard tripped (O3DR_STATUS_INTERNAL): results are invalid */
#define O3DR_ERR_PEER           -8 /* o3dr_merge_partitioned: another rank failed; every rank left the exchange together */

/* `mem` values */
#define O3DR_MEM_HOST   0
#define O3DR_MEM_DEVICE 1

/* bits of `*status` */
#define O3DR_STATUS_VOXEL_OVERFLOW 1u /* PCL VoxelGrid "Leaf size is too small ... Integer indices
                                         would overflow": output = input, unfiltered [PCL 1.8
                                         filters/impl/voxel_grid.hpp applyFilter] */

#define O3DR_STATUS_INTERNAL 0x80000000u /* a gather guard found a record or point id outside its cloud (library
                                            bug, never expected); calls that read the status return
                                            O3DR_ERR_INTERNAL and an empty output */

typedef struct o3dr_point {
    float    x, y, z;
    uint32_t rgba;
} o3dr_point;

/* Hot-path parameters = the `Pose` members the four functions read. */
typedef struct o3dr_params {
    double   min_disparity;        /* pose.h:93  minDisparity = 64, strict '>' (pose_functions.cpp:1107) */
    double   voxel_size;           /* pose.h:118 voxel_size = 0.1 */
    int32_t  bounding_box;         /* pose.h:94  boundingBox = 20 */
    int32_t  cutout_ratio;         /* pose.h:126 cutout_ratio = 8; cols_start_aft_cutout=(int)(cols/ratio) pose_functions.cpp:638 */
    int32_t  jump_pixels;          /* pose.h:96  jump_pixels = 10; 0 = keypoints only, 1 = dense (no keypoint pass) */
    uint32_t min_points_per_voxel; /* pose.h:108 = 1; only the combined merge uses it (pose_functions.cpp:1693) */
    int32_t  dont_downsample;      /* --dont_downsample, pose.cpp:609 */
    int32_t  sor_enable;           /* statistical outlier removal of the per-frame path (pose_functions.cpp:1673-1686:
                                      mean_k 50, 1 sigma, active iff !combined && jump_pixels > 0).  1 = on, as in the
                                      reference (the default); 0 = off (the measured GPU configs) */
    int32_t  blur_kernel;          /* pose.h:98 blur_kernel = 1; > 1: the disparity image goes through
                                      cv::bilateralFilter(d = blur_kernel, sigmaColor = 2*blur_kernel,
                                      sigmaSpace = blur_kernel/2 (integer division)) first, pose_functions.cpp:1040-1047 */
    int32_t  disparity_f64;        /* --use_segment_labels (pose_functions.cpp:1037,1102): the disparity images handed to
                                      the frame calls are CV_64F (doubles, pitch and frame stride still in bytes) and
                                      are read with at<double>; 0 = CV_8UC1.  Not combinable with blur_kernel > 1
                                      (cv::bilateralFilter rejects CV_64F) */
} o3dr_params;

typedef struct o3dr_ctx o3dr_ctx; /* opaque */

/* ---- library ------------------------------------------------------------------------------ */
int         o3dr_version(void);
/* text of the last failure on the calling thread ("" if none) */
const char* o3dr_last_error(void);
/* fills *p with the reference defaults (pose.h:92-126) */
void        o3dr_default_params(o3dr_params* p);

/* ---- context ------------------------------------------------------------------------------ */
/* One context = one stream + workspaces on device `device_id`.  Fails with O3DR_ERR_NO_DEVICE
 * when there is no GPU: there is no CPU fallback behind this ABI. */
int o3dr_ctx_create(int device_id, o3dr_ctx** out_ctx);
int o3dr_ctx_destroy(o3dr_ctx* ctx);
/* Issue work on a caller-owned hipStream_t (e.g. torch's current stream); NULL restores the
 * context's own stream. */
int o3dr_ctx_set_stream(o3dr_ctx* ctx, void* hip_stream);
int o3dr_ctx_synchronize(o3dr_ctx* ctx);
/* Q of the rectified stereo pair: Pose::Q, pose.h:128, read at pose_functions.cpp:467-476 */
int o3dr_set_camera(o3dr_ctx* ctx, const double Q[16]);
int o3dr_set_params(o3dr_ctx* ctx, const o3dr_params* p);
int o3dr_get_params(o3dr_ctx* ctx, o3dr_params* p);

/* ---- A1: Pose::createSingleImgPtCloud (pose.h:198, pose_functions.cpp:1030-1134) ---------------
 * Camera-frame cloud of one frame: keypoint pass (iff jump_pixels != 1; `kp_xy` = n_kp pairs of
 * float KeyPoint::pt.x,.y, truncated to int like pose_functions.cpp:1061) followed by the row-major
 * grid pass (iff jump_pixels > 0).  `out_capacity` must be >= o3dr_max_points(rows, cols) + n_kp. */
int o3dr_create_single_img_pt_cloud(o3dr_ctx* ctx,
                                    const uint8_t* disp, int64_t disp_pitch,
                                    const uint8_t* bgr, int64_t bgr_pitch,
                                    int32_t rows, int32_t cols,
                                    const float* kp_xy, int32_t n_kp,
                                    o3dr_point* out, int64_t out_capacity, int64_t* n_out,
                                    int32_t mem);
/* number of grid-pass candidates for the context's current params: Ny*Nx of SURVEY section 8 */
int64_t o3dr_max_points(o3dr_ctx* ctx, int32_t rows, int32_t cols);

/* ---- disparity pre-passes -------------------------------------------------------------------
 * cv::bilateralFilter on a CV_8UC1 image — replaces the call at pose_functions.cpp:1044 (OpenCV 3.1
 * bilateralFilter_8u, BORDER_DEFAULT, fp32 sums grouped as an x86-64 build groups them).  d <= 0 derives the
 * radius from sigma_space like OpenCV; radius (d/2) above 64 is O3DR_ERR_INVALID_ARG.  src and dst must not
 * overlap.  With O3DR_MEM_DEVICE the call is asynchronous on the context's stream.  Runs inside the frame calls
 * by itself when o3dr_params.blur_kernel > 1. */
int o3dr_bilateral_filter_u8(o3dr_ctx* ctx, const uint8_t* src, int64_t src_pitch, int32_t rows, int32_t cols, int32_t d,
                             double sigma_color, double sigma_space, uint8_t* dst, int64_t dst_pitch, int32_t mem);
/* Pose::getVariance(disp, false) of n_frames disparity images (pose_functions.cpp:987-1028; the frame gate of
 * pose.cpp:187-196 rejects a frame when it exceeds 5), over the ROI set by o3dr_params.  variance_out: n_frames
 * doubles in HOST memory.  The mean is bit-identical to the reference's; the variance is summed per disparity
 * level instead of per pixel and agrees to fp64 rounding (within N * 2^-53 relative, N = ROI pixels). */
int o3dr_disparity_variance(o3dr_ctx* ctx, const uint8_t* disp, int64_t disp_pitch, int64_t disp_frame_stride, int32_t rows,
                            int32_t cols, int32_t n_frames, double* variance_out, int32_t mem);

/* ---- A2: Pose::transformPtCloud (pose.h:199, pose_functions.cpp:1358-1362) ----------------------
 * out[i].xyz = T * in[i].xyz in fp32, ((m0*x + m1*y) + m2*z) + m3, no fused multiply-add
 * [PCL 1.8 common/impl/transforms.hpp, dense branch]; colour copied.  in == out is allowed
 * (the in-place re-transform of cloud_big after ICP, pose.cpp:353). */
int o3dr_transform_pt_cloud(o3dr_ctx* ctx, const o3dr_point* in, int64_t n,
                            const float T[16], o3dr_point* out, int32_t mem);

/* ---- A1+A2 fused: what createAndTransformPtCloud does before downsampling (pose.cpp:603-607) -- */
int o3dr_reproject_transform(o3dr_ctx* ctx,
                             const uint8_t* disp, int64_t disp_pitch,
                             const uint8_t* bgr, int64_t bgr_pitch,
                             int32_t rows, int32_t cols,
                             const float T[16],
                             const float* kp_xy, int32_t n_kp,
                             o3dr_point* out, int64_t out_capacity, int64_t* n_out,
                             int32_t mem);

/* ---- A4: pcl::VoxelGrid<PointXYZRGB>::applyFilter as used at pose_functions.cpp:1689-1700 ------
 * One output point per occupied voxel with >= min_points points: fp32 centroid, truncated mean
 * colour, ascending linear voxel index (x fastest, then y, then z).  `z_offset` is added to z in
 * fp32 on load and subtracted in fp32 on store (pose_functions.cpp:1666,1702-1704; 0 = none).
 * Points of one voxel are summed in input order (see DESIGN.md "summation order").
 * `out_capacity` must be >= n_in (the overflow fallback returns the input unchanged). */
int o3dr_voxel_grid(o3dr_ctx* ctx, const o3dr_point* in, int64_t n_in,
                    const float leaf[3], uint32_t min_points, float z_offset,
                    o3dr_point* out, int64_t out_capacity, int64_t* n_out, uint32_t* status,
                    int32_t mem);

/* ---- A3b: pcl::StatisticalOutlierRemoval<PointXYZRGB> as configured at pose_functions.cpp:1679-1684 ---
 * (setMeanK(50), setStddevMulThresh(1.0)) on its own: exact 51-nearest-neighbour search, mean neighbour
 * distance per point, global mean + 1 sigma gate; inliers keep their order.  Clouds of <= 50 points
 * pass through (the reference reads past its neighbour list there).  out_capacity >= n_in. */
int o3dr_statistical_outlier_removal(o3dr_ctx* ctx, const o3dr_point* in, int64_t n_in, o3dr_point* out,
                                     int64_t out_capacity, int64_t* n_out, int32_t mem);

/* ---- A3a / A5: Pose::downsamplePtCloud (pose.h:216, pose_functions.cpp:1654-1709) --------------
 * combined == 0: per-frame mode, leaf (voxel_size/5)^3, min_points 0   (:1698)
 * combined != 0: 2.5-D merge, z += 500, leaf (voxel_size, voxel_size, 1000),
 *                min_points_per_voxel, z -= 500                        (:1666,1693-1694,1702-1704) */
int o3dr_downsample_pt_cloud(o3dr_ctx* ctx, const o3dr_point* in, int64_t n_in, int32_t combined,
                             o3dr_point* out, int64_t out_capacity, int64_t* n_out,
                             uint32_t* status, int32_t mem);

/* ---- A6: Pose::createAndTransformPtCloud (pose.h:231, pose.cpp:596-636) -------------------------
 * A1 -> A2 -> (A3a unless dont_downsample) for one frame into a caller-owned cloud. */
int o3dr_create_and_transform_pt_cloud(o3dr_ctx* ctx,
                                       const uint8_t* disp, int64_t disp_pitch,
                                       const uint8_t* bgr, int64_t bgr_pitch,
                                       int32_t rows, int32_t cols,
                                       const float T[16],
                                       const float* kp_xy, int32_t n_kp,
                                       o3dr_point* out, int64_t out_capacity, int64_t* n_out,
                                       uint32_t* status, int32_t mem);

/* ---- A7: fan-out + accumulate (pose.cpp:365-434) and the final merge (pose.cpp:527-532) --------
 * The context owns `cloud_big` in HBM.  o3dr_accumulate_frames runs A6 for `n_frames` frames
 * (frame f at base + f*frame_stride; pose f at poses + 16*f) and appends the per-frame results in
 * frame order, entirely on the device and asynchronously (no host round trip per frame).
 * o3dr_accumulate_frames_kp also runs the keypoint pass (active iff jump_pixels != 1, pose_functions.cpp:1057-1091):
 * frame f's keypoints are kp_xy[2*kp_offsets[f] .. 2*kp_offsets[f+1]) (x,y float pairs, memory kind `mem`);
 * kp_offsets is a HOST array of n_frames+1 non-decreasing entries; NULL = no keypoints.
 * `status_or` (host pointer, may be NULL) is written by o3dr_cloud_big_size/o3dr_finalize. */
int o3dr_accumulate_frames(o3dr_ctx* ctx,
                           const uint8_t* disp, int64_t disp_frame_stride, int64_t disp_pitch,
                           const uint8_t* bgr, int64_t bgr_frame_stride, int64_t bgr_pitch,
                           int32_t rows, int32_t cols,
                           const float* poses, int32_t n_frames, int32_t mem);
int o3dr_accumulate_frames_kp(o3dr_ctx* ctx,
                              const uint8_t* disp, int64_t disp_frame_stride, int64_t disp_pitch,
                              const uint8_t* bgr, int64_t bgr_frame_stride, int64_t bgr_pitch,
                              int32_t rows, int32_t cols,
                              const float* poses, int32_t n_frames,
                              const float* kp_xy, const int64_t* kp_offsets, int32_t mem);
/* reserve HBM for cloud_big (points); optional — it grows on demand */
int o3dr_cloud_big_reserve(o3dr_ctx* ctx, int64_t n_points);
int o3dr_cloud_big_reset(o3dr_ctx* ctx);
/* synchronises; *status = OR of the per-frame O3DR_STATUS_* bits since the last reset */
int o3dr_cloud_big_size(o3dr_ctx* ctx, int64_t* n, uint32_t* status);
int o3dr_cloud_big_read(o3dr_ctx* ctx, o3dr_point* out, int64_t out_capacity, int64_t* n_out, int32_t mem);
/* append externally produced points (a peer rank's shard after the RCCL all-gather, or a PLY) */
int o3dr_cloud_big_append(o3dr_ctx* ctx, const o3dr_point* pts, int64_t n, int32_t mem);
/* in-place rigid re-transform of cloud_big (pose.cpp:353) */
int o3dr_cloud_big_transform(o3dr_ctx* ctx, const float T[16]);
/* cloud_small = downsamplePtCloud(cloud_big, true) (pose.cpp:530); cloud_big is left intact */
int o3dr_finalize(o3dr_ctx* ctx, o3dr_point* out, int64_t out_capacity, int64_t* n_out,
                  uint32_t* status, int32_t mem);

/* ---- multi-GPU merge (frames sharded over ranks) ------------------------------
 * The reference's final merge (pose.cpp:530) runs one voxel grid over ALL frames' per-frame voxels.
 * With one rank per GPU: (1) every rank takes o3dr_cloud_big_bbox and the ranks min/max-reduce it;
 * (2) o3dr_cloud_big_partition lays the combined grid (params: voxel_size, +500 z offset) over that
 * GLOBAL box, cuts its linear index range into n_parts equal slices and stably reorders cloud_big so
 * that slice 0's points come first: counts[p] = points of slice p; (3) the ranks exchange slices
 * (all-to-all over RCCL; received segments appended in source-rank order keep global order);
 * (4) o3dr_finalize_global merges the local slice with the grid over the same global box.  The
 * concatenation of the ranks' outputs in rank order equals the single-GPU result bit for bit.
 * When PCL's overflow guard fires for the global box, *status carries O3DR_STATUS_VOXEL_OVERFLOW, the
 * cloud is left as is and all counts are 0 (the merge then returns its input: skip the exchange). */
int o3dr_cloud_big_bbox(o3dr_ctx* ctx, float mn[3], float mx[3], int64_t* n);
int o3dr_cloud_big_partition(o3dr_ctx* ctx, const float gmin[3], const float gmax[3], int32_t n_parts,
                             int64_t* counts, uint32_t* status);
int o3dr_finalize_global(o3dr_ctx* ctx, const float gmin[3], const float gmax[3], o3dr_point* out,
                         int64_t out_capacity, int64_t* n_out, uint32_t* status, int32_t mem);
/* Zero-copy plumbing for step (3): *ptr = HBM address of cloud_big and its length (the send buffer; valid
 * until the next append/partition/adopt); an HBM receive buffer for n_points points; and "the first
 * n_points of the receive buffer are the new cloud_big" (status bits are kept). */
int o3dr_cloud_big_view(o3dr_ctx* ctx, void** ptr, int64_t* n);
int o3dr_cloud_big_recv_buffer(o3dr_ctx* ctx, int64_t n_points, void** ptr);
int o3dr_cloud_big_adopt(o3dr_ctx* ctx, int64_t n_points);
/* The same two steps with their small data kept in HBM, so that the whole exchange needs ONE host read-back (the
 * all-to-all's sizes): o3dr_cloud_big_header_dev writes this rank's 32-byte header {float min[3], max[3]; int64 count}
 * to the DEVICE buffer hdr_dev; after the ranks all-gathered their headers, o3dr_cloud_big_partition_dev folds the
 * n_hdrs headers at hdrs_dev into the global box and partitions as above, leaving n_parts int64 slice counts
 * followed by one int64 status word (O3DR_STATUS_VOXEL_OVERFLOW: all counts 0, cloud unchanged) in the DEVICE buffer
 * counts_dev.  Both are asynchronous on the context's stream.  o3dr_cloud_big_adopt is stream-ordered as well: what
 * fills the receive buffer must be ordered before the stream's next work by the caller (same stream, or an event).
 * o3dr_cloud_big_assume_size: the caller read its own header back (with the all-to-all's sizes) and tells the library
 * the exact size of cloud_big, so that o3dr_cloud_big_view / o3dr_finalize* need no round trip of their own for it. */
int o3dr_cloud_big_header_dev(o3dr_ctx* ctx, void* hdr_dev);
/* The partition in two halves, so that a rank's own points move ONCE and are never sent to itself (round 4; what
 * o3dr_merge_partitioned and dist.merge_partitioned use):
 *   o3dr_cloud_big_slice_counts_dev - like o3dr_cloud_big_partition_dev, but nothing moves: n_parts int64 slice sizes + the
 *     status word in the DEVICE buffer counts_dev; the (slice, tile) table stays in the workspace;
 *   o3dr_cloud_big_place_slices - after the ranks exchanged their counts: `counts` = this rank's n_parts slice sizes (host),
 *     n_before / n_after = points it will receive from lower / higher ranks.  One pass lays cloud_big out as
 *     [n_before free | own slice | n_after free | the slices that leave, in rank order]; *send_offset = where those start.
 *     A rank that neither sends nor receives keeps its cloud untouched.  The exchange then receives straight into the gaps
 *     (o3dr_cloud_big_raw_view: address and capacity of the buffer) and o3dr_cloud_big_set_size(n_before + own + n_after)
 *     makes that prefix the cloud (stream-ordered, like o3dr_cloud_big_adopt). */
int o3dr_cloud_big_slice_counts_dev(o3dr_ctx* ctx, const void* hdrs_dev, int32_t n_hdrs, int32_t n_parts, int64_t* counts_dev);
int o3dr_cloud_big_place_slices(o3dr_ctx* ctx, int32_t n_parts, int32_t own_part, const int64_t* counts, int64_t n_before,
                                int64_t n_after, int64_t* send_offset);
int o3dr_cloud_big_set_size(o3dr_ctx* ctx, int64_t n_points);
int o3dr_cloud_big_raw_view(o3dr_ctx* ctx, void** ptr, int64_t* capacity_points);
int o3dr_cloud_big_assume_size(o3dr_ctx* ctx, int64_t n_points);
int o3dr_cloud_big_partition_dev(o3dr_ctx* ctx, const void* hdrs_dev, int32_t n_hdrs, int32_t n_parts, int64_t* counts_dev);

/* The whole exchange in ONE call, for C++ hosts (the reference's merge sits in its C++ main flow, pose.cpp:527-532): one
 * host thread and one context per GPU, `nccl_comm` = that GPU's ncclComm_t (RCCL over xGMI; libo3dr resolves RCCL with
 * dlopen at first use and has no link-time dependency on it).  Steps (1)-(4) above with the small data kept in HBM (two
 * all-gathers of a few bytes, ONE host read-back, the slices placed in one pass, one grouped send/receive all-to-all
 * straight into the gaps left for it - the rank's own slice is not sent -, the local merge over the global box), then,
 * with gather_result != 0, an all-gather of the merged
 * slices: `out` receives the whole merged cloud (the single-GPU result, bit for bit); with gather_result == 0 this
 * rank's slice.  A rank that does not want the result passes out = NULL, out_capacity = 0 (it still takes part in every
 * collective).  *n_total = points merged over all ranks.  Must be called by all ranks of the communicator.
 * o3dr_comm_init_all / o3dr_comm_destroy wrap ncclCommInitAll / ncclCommDestroy for single-process hosts (devices ==
 * NULL: 0 .. n_devices-1). */
int o3dr_merge_partitioned(o3dr_ctx* ctx, void* nccl_comm, int32_t gather_result, o3dr_point* out, int64_t out_capacity,
                           int64_t* n_out, int64_t* n_total, uint32_t* status, int32_t mem);
/* Failure is COLLECTIVE: a rank whose own step fails (its header, the partition, an allocation, the local merge) keeps
 * taking part in the collectives that remain with an error word in place of its data - the header's count, a word next
 * to the slice counts, the merged size - and every rank returns at the same point: the failing rank with its own code,
 * the others with O3DR_ERR_PEER.  No rank is left waiting inside a collective its peer never enters.  (Buffers that
 * have to grow between the count matrix and the all-to-all are agreed on with one more 8-byte all-gather, only in calls
 * in which some rank - known to all from the capacities sent with the counts - has to grow one.)  After an error every
 * context stays usable and cloud_big keeps its points (possibly reordered by slice: a later merge gives the same
 * result).  What cannot be made collective is a failure of RCCL itself (O3DR_ERR_HIP).
 * o3dr_merge_partitioned_stats: what the last call of this context moved - out[0] points of this rank before the
 * exchange, [1] points sent to other ranks, [2] points received from other ranks, [3] / [4] the same in bytes (what
 * crosses xGMI), [5] points entering this rank's merge, [6] agreement rounds (0 or 1), [7] points of all ranks.
 * o3dr_cloud_big_capacity: points the cloud buffer and the receive buffer hold without reallocating. */
int o3dr_merge_partitioned_stats(o3dr_ctx* ctx, int64_t out[8]);
int o3dr_cloud_big_capacity(o3dr_ctx* ctx, int64_t* cloud_points, int64_t* recv_points);
int o3dr_comm_init_all(int32_t n_devices, const int32_t* devices, void** comms_out);
int o3dr_comm_destroy(void* comm);
/* hipHostRegister / hipHostUnregister for hosts that link nothing but this ABI: page-locked frame stacks handed to
 * o3dr_accumulate_frames with O3DR_MEM_HOST cross PCIe by DMA instead of through the runtime's bounce buffers */
int o3dr_host_register(void* ptr, int64_t bytes);
int o3dr_host_unregister(void* ptr);

/* ---- measurement hooks (bench.py; not part of the reference surface) ------------------------ */
/* kernel ids for o3dr_profile_* */
#define O3DR_K_COUNT        0  /* grid-pass valid count per tile */
#define O3DR_K_REPROJECT    1  /* fused reproject + SE(3) + ordered compaction */
#define O3DR_K_KEYGEN       2  /* voxel linear index per point */
#define O3DR_K_SORT_HIST    3  /* radix digit histogram */
#define O3DR_K_SORT_SCATTER 4  /* radix stable scatter */
#define O3DR_K_SEGMENT      5  /* voxel run heads + counts */
#define O3DR_K_CENTROID     6  /* ordered per-voxel sums -> centroid */
#define O3DR_K_OTHER        7  /* scans, grid setup, copies */
#define O3DR_K_CENTROID_RUNS 8 /* ordered per-voxel sums over group runs of points, one wave per voxel group (whole-cloud calls) */
#define O3DR_K_NUM          9
/* Bracket every launch of kernel `kernel_id` (or all kernels if -1) with HIP events on the
 * context's stream; 0 launches are bracketed when disabled (the default). */
int o3dr_profile_enable(o3dr_ctx* ctx, int32_t kernel_id, int32_t enable);
/* Synchronises; total milliseconds and launch count of `kernel_id` since enable/reset. */
int o3dr_profile_read(o3dr_ctx* ctx, int32_t kernel_id, double* total_ms, int64_t* launches);
int o3dr_profile_reset(o3dr_ctx* ctx);
/* Synchronises; counters since the last o3dr_profile_reset, for algorithmic-byte accounting:
 * out[0] = sum over voxel grids of (records x radix passes actually run), out[1] = points that entered
 * voxel grids, out[2] = points that left them, out[3] = 0, out[4] = records that entered the sorts (points or runs
 * of points), out[5..7] = 0. */
int o3dr_profile_stats(o3dr_ctx* ctx, int64_t out[8]);
/* device name / arch / CU count of the context's device, for bench headers */
int o3dr_device_info(o3dr_ctx* ctx, char* name, int32_t name_len, int32_t* cu_count, int64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* O3DR_H */
