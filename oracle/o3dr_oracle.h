/*
 * o3dr_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's per-frame reconstruction path, used only as the parity
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * online_3d_reconstruction_amd/ may include, link or call this.
 *
 * PARITY PINNING: the reference ships no tests, golden vectors or fixtures for this path, and it
 * cannot be built here (OpenCV 3.1+CUDA, PCL 1.8, Boost, VTK, Eigen, FLANN are absent; SURVEY.md
 * section 8c) — so this oracle is pinned by analytic known-answer vectors derived from the bundled
 * calibration (cam13calib.yml) and by invariants of the bundled output cloud (build/cloud.ply),
 * not by outputs of the reference itself: "parity unpinned" in the sense of the task statement.
 */
#ifndef O3DR_ORACLE_H
#define O3DR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_point {
    float    x, y, z;
    uint32_t rgba; /* a<<24 | r<<16 | g<<8 | b */
} orc_point;

/* order in which the points of one voxel are summed */
#define ORC_ORDER_STABLE  0 /* ascending input index (canonical; what libo3dr implements) */
#define ORC_ORDER_STDSORT 1 /* libstdc++ std::sort on (idx, point index) comparing idx only — the
                               call PCL makes; implemented in o3dr_oracle_stdsort.cpp */

#define ORC_STATUS_VOXEL_OVERFLOW 1u

/* A1 — Pose::createSingleImgPtCloud, pose_functions.cpp:1030-1134.  Returns the point count.
 * `out` must hold n_kp + Ny*Nx points. */
int64_t orc_create_single_img_pt_cloud(const uint8_t* disp, int64_t disp_pitch,
                                       const uint8_t* bgr, int64_t bgr_pitch,
                                       int32_t rows, int32_t cols, const double Q[16],
                                       int32_t bounding_box, int32_t cols_start_aft_cutout,
                                       double min_disparity, int32_t jump_pixels,
                                       const float* kp_xy, int32_t n_kp, orc_point* out);

/* A1 with --use_segment_labels: the disparity image is CV_64F (pose_functions.cpp:1037,1064-1069,1100-1105) */
int64_t orc_create_single_img_pt_cloud_f64(const double* disp, int64_t disp_pitch_bytes,
                                           const uint8_t* bgr, int64_t bgr_pitch,
                                           int32_t rows, int32_t cols, const double Q[16],
                                           int32_t bounding_box, int32_t cols_start_aft_cutout,
                                           double min_disparity, int32_t jump_pixels,
                                           const float* kp_xy, int32_t n_kp, orc_point* out);

/* A2 — Pose::transformPtCloud, pose_functions.cpp:1358-1362 -> pcl::transformPointCloud (dense). */
void orc_transform_pt_cloud(const orc_point* in, int64_t n, const float T[16], orc_point* out);

/* A4 — pcl::VoxelGrid<PointXYZRGB>::applyFilter (PCL 1.8).  Returns the output count. */
int64_t orc_voxel_grid(const orc_point* in, int64_t n, const float leaf[3], uint32_t min_points,
                       int32_t order, orc_point* out, uint32_t* status);

/* A3b — pcl::StatisticalOutlierRemoval<PointXYZRGB> as configured at pose_functions.cpp:1679-1684
 * (setMeanK(50), setStddevMulThresh(1.0)); exact k-nearest-neighbour search like PCL's KdTreeFLANN.
 * Returns the number of points kept (input order preserved); distances_out (optional, n floats)
 * receives the mean neighbour distance of every input point.  brute != 0 forces the O(n^2) search
 * (self-check of the grid search in the tests). */
int64_t orc_statistical_outlier_removal(const orc_point* in, int64_t n, int32_t mean_k, double stddev_mul,
                                        orc_point* out, float* distances_out, int32_t brute);

/* A3a/A5 — Pose::downsamplePtCloud, pose_functions.cpp:1654-1709 (statistical outlier removal,
 * :1673-1686, is not applied: see SURVEY.md section 8a row A3b). */
int64_t orc_downsample_pt_cloud(const orc_point* in, int64_t n, double voxel_size, int32_t combined,
                                uint32_t min_points_per_voxel, int32_t order, orc_point* out,
                                uint32_t* status);

/* A6 — Pose::createAndTransformPtCloud, pose.cpp:596-636.  `scratch` holds 2*(n_kp+Ny*Nx) points. */
int64_t orc_create_and_transform_pt_cloud(const uint8_t* disp, int64_t disp_pitch,
                                          const uint8_t* bgr, int64_t bgr_pitch,
                                          int32_t rows, int32_t cols, const double Q[16],
                                          int32_t bounding_box, int32_t cols_start_aft_cutout,
                                          double min_disparity, int32_t jump_pixels,
                                          const float* kp_xy, int32_t n_kp, const float T[16],
                                          double voxel_size, int32_t dont_downsample, int32_t order,
                                          orc_point* scratch, orc_point* out, uint32_t* status);

/* A7 + pose.cpp:530 as the reference runs them: A6 for n_frames frames on `threads` frame-parallel
 * POSIX threads (pose.cpp:392-413 uses 7), results appended in frame order, then the combined merge.
 * cloud_big_out / merged_out may be NULL (timing only).  Returns the merged count; *n_big = sum of
 * per-frame outputs.  sor != 0 inserts the statistical outlier removal in front of every per-frame
 * voxel grid (pose_functions.cpp:1673-1686). */
int64_t orc_run_frames(const uint8_t* disp, int64_t disp_fstride, int64_t disp_pitch, const uint8_t* bgr,
                       int64_t bgr_fstride, int64_t bgr_pitch, int32_t rows, int32_t cols, const double Q[16],
                       int32_t bounding_box, int32_t cols_start_aft_cutout, double min_disparity,
                       int32_t jump_pixels, const float* poses, int32_t n_frames, double voxel_size,
                       uint32_t min_points_per_voxel, int32_t sor, int32_t threads, orc_point* cloud_big_out,
                       int64_t* n_big, orc_point* merged_out);

/* the same with the summation order of EVERY voxel grid (per frame and merge) chosen: ORC_ORDER_STDSORT is the
 * reference's actual std::sort order (pose_functions.cpp:1700), which no fixture of the reference pins */
int64_t orc_run_frames_order(const uint8_t* disp, int64_t disp_fstride, int64_t disp_pitch, const uint8_t* bgr,
                             int64_t bgr_fstride, int64_t bgr_pitch, int32_t rows, int32_t cols, const double Q[16],
                             int32_t bounding_box, int32_t cols_start_aft_cutout, double min_disparity,
                             int32_t jump_pixels, const float* poses, int32_t n_frames, double voxel_size,
                             uint32_t min_points_per_voxel, int32_t sor, int32_t threads, int32_t order,
                             orc_point* cloud_big_out, int64_t* n_big, orc_point* merged_out);

/* A1 pre-pass — cv::bilateralFilter(disp, out, d, sigma_color, sigma_space) as called at
 * pose_functions.cpp:1040-1047 with (blur_kernel, blur_kernel*2, blur_kernel/2): OpenCV 3.1.0
 * modules/imgproc/src/smooth.cpp bilateralFilter_8u, single channel, BORDER_DEFAULT (reflect-101),
 * non-IPP build (the IPP branch is compiled out in 3.1.0).  order: how the per-pixel float sums are grouped —
 * ORC_BILATERAL_SSE3: groups of four neighbours reduced pairwise with haddps, then added (x86-64 builds, which
 * the reference's CMakeCache.txt shows); ORC_BILATERAL_SCALAR: one neighbour after the other (builds without SSE3).
 * The weight tables use the host's exp().  PINNED by an output of the reference itself (round 4): the three-channel
 * branch of the same invoker, orc_bilateral_filter_u8c3 — same border, tables, neighbour order, four-at-a-time grouping,
 * cvRound — reproduces build/output/bilateralFiltered_15.png and _31.png (build/images/1248.png through the call of
 * pose_functions.cpp:1044 with blur_kernel 15 / 31) byte for byte in the SSE3 order; the scalar order does not. */
#define ORC_BILATERAL_SSE3   0
#define ORC_BILATERAL_SCALAR 1
void orc_bilateral_filter_u8(const uint8_t* src, int64_t src_pitch, int32_t rows, int32_t cols, int32_t d,
                             double sigma_color, double sigma_space, int32_t order, uint8_t* dst, int64_t dst_pitch);
/* CV_8UC3 (interleaved B,G,R) through the cn == 3 branch; test infrastructure for the pin above, not on the hot path */
void orc_bilateral_filter_u8c3(const uint8_t* src, int64_t src_pitch, int32_t rows, int32_t cols, int32_t d,
                               double sigma_color, double sigma_space, int32_t order, uint8_t* dst, int64_t dst_pitch);

/* frame gate — Pose::getVariance(disp, false), pose_functions.cpp:987-1028 (mean over the ROI of the
 * disparities > min_disparity, divided by the full ROI size; sequential fp64 sums in row-major order). */
double orc_disparity_variance(const uint8_t* disp, int64_t pitch, int32_t rows, int32_t cols, int32_t bounding_box,
                              int32_t cols_start_aft_cutout, double min_disparity);

/* voxel keys only (occupancy checks): writes the uint32 linear index PCL computes for each point,
 * returns 0, or ORC_STATUS_VOXEL_OVERFLOW when PCL would bail out (keys then undefined). */
uint32_t orc_voxel_keys(const orc_point* in, int64_t n, const float leaf[3], uint32_t* keys,
                        int32_t min_b[3], int32_t div_b[3]);

/* provided by o3dr_oracle_stdsort.cpp: sorts (idx, point index) pairs exactly as PCL does */
void orc_stdsort_pairs(uint32_t* idx, uint32_t* point_index, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
