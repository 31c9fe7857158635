// o3dr_device.h — device-side data structures and kernel launchers shared by the C-ABI layer.
// gfx950 only (64-wide wavefronts are assumed throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/o3dr.h"

namespace o3dr {

// ---- tiling constants -------------------------------------------------------------------------
constexpr int kWave = 64;
// reprojection: one 256-thread workgroup covers 1024 grid-pass candidates (4 per lane)
constexpr int kEmitThreads = 256;
constexpr int kEmitPerLane = 4;
constexpr int kEmitTile = kEmitThreads * kEmitPerLane;
// radix sort: 8 waves x 16 records per lane x 64 lanes = 8192 records per workgroup
constexpr int kSortWaves = 8;
constexpr int kSortThreads = kSortWaves * kWave;
constexpr int kSortRounds = 16;
constexpr int kSortWaveItems = kSortRounds * kWave;
constexpr int kSortTile = kSortWaves * kSortWaveItems;
// the scatter takes a tile in kScatParts parts, one workgroup of kScatWaves waves each (k_radix_scatter_lane): more and
// smaller workgroups per CU overlap their LDS phases better; the histograms keep the 8192-record tile
constexpr int kScatParts = 2;
constexpr int kScatWaves = kSortWaves / kScatParts;
constexpr int kScatThreads = kScatWaves * kWave;
constexpr int kScatTile = kScatWaves * kSortWaveItems;
constexpr int kMaxRadixBits = 7;           // digits are 1..7 bits wide, chosen per frame (k_voxel_geom).  Measured with
                                          // the ballot scatter: 7-bit passes 3.8 TB/s, 10-bit ones 2.5 TB/s (32-byte
                                          // output runs); the lane-counting scatter keeps 128 x 17 counters per wave
constexpr int kMaxRadix = 1 << kMaxRadixBits;
constexpr int kMaxPasses = 5;            // 5 x 7 bits covers a full 32-bit index
static_assert(kMaxRadix <= 2 * kSortThreads, "k_radix_scatter handles two digits per thread");
// generic per-point kernels
constexpr int kSlabClasses = 4;  // layout classes inside an emit tile of the fused batch path (slab_class)
constexpr int kXcds = 8;   // accelerator complex dies of an MI355X, each with its own L2 (xcd_chunk_item)
constexpr int kPtThreads = 256;
constexpr int kSegTile = 1024;  // sorted keys per workgroup in the run-head kernels
constexpr int kMinmaxBlocks = 1024;  // workgroups (= bounding-box slots) of the stand-alone min/max pass
constexpr int kSmallMax = 8192;      // points one workgroup takes through the whole path in one launch (kernels/small.inc)
// whole-cloud voxel grids (the merge): records of the sort are runs of consecutive points inside one GROUP of
// 2^kGroupBits consecutive voxel indices; one wave then sums a group, lane = voxel (k_centroid_groups)
constexpr int kGroupBits = 5;
constexpr int kGroupCells = 1 << kGroupBits;
static_assert(kGroupCells <= kWave, "one lane per voxel of a group");
constexpr int kGroupWaves = 4;            // groups per workgroup of k_centroid_groups
// grouped records are used when the runs average at least kGroupMinRunNum / kGroupMinRunDen points.  Round 4 measured the
// break-even again (round 2 had set 8): forced on BASELINE configs[4]'s shape (4.1 points per run) 9.03 -> 6.94 ms per
// 200-frame step, on configs[3]'s (raw pixel-order points, 1.6 per run) 9.49 -> 7.75 ms - the run sort moves fewer
// records and k_centroid_groups reads the cloud once, coalesced, where k_centroid gathers 16 bytes per 64-byte sector
constexpr int kGroupMinRunNum = 5, kGroupMinRunDen = 4;
constexpr int64_t kGroupMinCloud = 1 << 20;  // whole-cloud calls on fewer points sort the points
constexpr int64_t kGroupMinSlots = 1 << 20;  // result slots (16 bytes each) a context always has for the grouped path

// ---- per-frame voxel grid geometry (PCL VoxelGrid members), written by k_voxel_geom -----------
struct VoxelGeom {
    float inv[3];       // inverse_leaf_size_
    int32_t min_b[3];   // min_b_
    int32_t div_b[3];   // div_b_
    uint32_t mul1, mul2;  // divb_mul_[1], divb_mul_[2]
    uint32_t overflow;    // dx*dy*dz > INT32_MAX  -> output = input
    uint32_t n;           // points in this frame
    uint32_t passes;      // radix passes this frame's index needs (0 when overflow)
    uint32_t bpp;         // bits per pass
    uint32_t buf0;        // buffer the first pass reads (sorted records end in buffer (passes + buf0) & 1)
    uint32_t grouped;     // 1: the records are runs of consecutive points of one voxel group (n = number of records, payload =
                          // record id into Workspace::run_start); 0: the records are the points (payload = point id)
};

// ---- statistical outlier removal (A3b): search grid + threshold, written by k_sor_plan / k_sor_threshold ----
constexpr int kSorMeanK = 50;  // sor0.setMeanK(50), pose_functions.cpp:1681
#ifndef O3DR_SOR_CELLPTS
#define O3DR_SOR_CELLPTS 30.0
#endif
constexpr double kSorCellPoints = O3DR_SOR_CELLPTS;  // points per column of the search grid, on average
struct SorGeom {
    float mnx, mny, inv_h, h;
    int32_t gx, gy;
    uint32_t n, active;
    double threshold;
};

// disparity-byte table of the rectified-stereo fast path: alpha = 1./(Q[14]*d + Q[15]), z = float(t2*alpha + 0)
struct QLutEntry {
    double alpha;
    float z, pad;
};

// ---- arguments of the fused reprojection kernels (A1 + A2) ------------------------------------
struct ReprojectArgs {
    const uint8_t* disp;  // frame f at disp + f*disp_fstride
    const uint8_t* bgr;
    int64_t disp_pitch, bgr_pitch, disp_fstride, bgr_fstride;
    const float* poses;   // xf_mode 2: 16 floats per frame (row-major) in HBM
    int32_t xf_mode;      // 0: camera frame (A1 only); 1: T below (single frame); 2: poses[f]
    float T[12];          // top three rows of the 4x4 pose, row-major
    int32_t rows, cols, bb, cs, jump;
    int32_t Ny, Nx;       // grid-pass extent
    int32_t n_tiles;      // ceil(Ny*Nx / kEmitTile)
    int32_t vec4;         // 1: jump==1, Nx%4==0, cs%4==0, pitches%4==0 -> packed 4-pixel loads
    double Q[16];
    double min_disp;
    int32_t min_disp_u8;  // for disparity BYTES: d > min_disp  <=>  (int)d > min_disp_u8 (set next to min_disp)
    int64_t out_fstride;  // points between consecutive frames' output regions
    int64_t mm_stride;    // bounding-box slots per frame
    const QLutEntry* lut; // 256 entries in HBM when Q has the rectified-stereo sparsity, else nullptr
    int32_t disp_f64;     // disparity image holds doubles (CV_64F, --use_segment_labels) instead of bytes
    // fused batch path only: the points of an emit tile are written class by class (grid slabs, see slab_class)
    float slab_inv[3];    // inverse leaf of the grid the points are meant for
    int32_t slab_shift;   // log2 of the slab thickness in cells; < 0: one class, plain pixel order
};

// All per-batch device buffers.  Sizes are for `frames` frames of at most `cap` points each.
struct Workspace {
    int32_t frames = 0;
    int64_t cap = 0;
    int32_t n_emit_tiles = 0, n_sort_tiles = 0, n_seg_tiles = 0;
    o3dr_point* pts = nullptr;     // frames*cap       transformed points (A1+A2 output)
    uint32_t* keys[2] = {nullptr, nullptr};  // frames*cap ping-pong
    uint32_t* vals[2] = {nullptr, nullptr};  // frames*cap ping-pong
    uint32_t* seg_start = nullptr; // frames*(cap+1)   start of each voxel run in the sorted order
    uint32_t* tile_cnt = nullptr;  // frames*n_emit_tiles
    uint32_t* hist = nullptr;      // frames*kMaxRadix*n_sort_tiles
    uint32_t* hist_part = nullptr; // frames*kMaxRadix*n_sort_tiles: digit counts of the FIRST scatter part of every tile
    uint32_t* seg_cnt = nullptr;   // frames*n_seg_tiles (run heads per tile, then kept runs per tile)
    uint8_t* head_bits = nullptr;  // frames*n_seg_tiles*256: run-head flags, 4 records per byte (k_run_heads -> k_run_starts)
    uint32_t* scan_partial = nullptr;  // chunk sums of the multi-workgroup scan
    // statistical outlier removal (sor_frames clouds of at most sor_cap points)
    SorGeom* sor_geom = nullptr;        // frames
    float4* sor_xyz = nullptr;          // frames*cap      coordinates in cell order
    float* sor_dist = nullptr;          // frames*cap      mean neighbour distance per point
    uint32_t* sor_cell_first = nullptr; // frames*(sor_max_cells+1): points in cells below c (exclusive scan of the populations)
    float2* sor_cell_z = nullptr;       // frames*(sor_max_cells+1): z range of every cell's points (for the handed-over queries)
    uint32_t sor_max_cells = 0;
    double* sor_partial = nullptr;      // frames*256*2
    o3dr_point* sor_pts = nullptr;      // frames*cap      inliers
    uint32_t* sor_n = nullptr;          // frames          inlier count
    uint32_t* sor_left = nullptr;       // frames*cap      queries (cell-sorted index) left to k_sor_knn_left
    uint32_t* sor_left_cnt = nullptr;   // frames
    int64_t sor_cap = 0;                // points per frame the arrays above are laid out for
    uint32_t* keep_idx = nullptr;  // frames*cap  (only when min_points > 1)
    uint32_t* run_start = nullptr; // frames*(cap+1)  first point of every group run (grouped path), + sentinel
    uint32_t* grp_cnt = nullptr;   // grp_slots/64 + 2: output voxels per group -> exclusive prefix (grouped path, frames == 1)
    int64_t grp_slots = 0;         // 16-byte result slots available in `pts` for the grouped path (0: pts not reserved)
    uint32_t* n_runs = nullptr;    // frames
    uint32_t* n_grp_out = nullptr; // frames: output voxels of a grouped cloud
    VoxelGeom* geom_runs = nullptr;  // frames: geom with n = number of group runs, records starting in buffer 1
    float* out_mm = nullptr;         // frames*ceil(cap/256)*4*6: bounding boxes of what k_centroid's waves appended
    float* out_mm_partial = nullptr; // kBoxFoldBlocks*6
    int32_t* wave_gc = nullptr;      // frames*ceil(cap/256)*4*6: voxel groups of the first / last point k_centroid's waves appended
    float* mm = nullptr;           // frames*mm_stride*6  per-workgroup bounding boxes (min xyz, max xyz)
    int64_t mm_stride = 0;         // slots per frame
    uint32_t* n_valid = nullptr;   // frames     points per frame after A1
    uint32_t* n_kp = nullptr;      // frames     keypoint-pass points (single-frame API), else 0
    uint32_t* n_vox = nullptr;     // frames     voxel runs
    uint32_t* n_out = nullptr;     // frames     output points (kept runs, or n_valid on overflow)
    uint64_t* out_off = nullptr;   // frames     absolute output offset of each frame
    VoxelGeom* geom = nullptr;     // frames
    size_t bytes = 0;
};

// device-resident statistics for bench.py's byte accounting
struct SortStats {
    uint64_t sort_record_passes;  // sum over voxel jobs of points * radix passes
    uint64_t voxel_points_in;     // points entering voxel grids (not counting overflow/passthrough)
    uint64_t voxel_points_out;    // points leaving them
    uint64_t reserved;
    uint64_t sort_records;        // records entering the sorts (points, or runs of points)
    uint64_t pad[3];
};

// where cloud_big records the heads of its group runs while it is appended to (k_centroid): the flags (4 records per
// byte, see k_run_heads) and the merge grid they are meant for
struct CloudHeads {
    uint32_t* flags;  // nullptr: not recorded
    float inv[3];     // inverse leaf of the merge's grid
    float z_offset;
    int32_t* wave_gc; // workspace: groups of the first and the last point every wave of k_centroid appended (6 ints per wave)
};

// device-resident counters of the accumulating cloud
struct CloudCounters {
    uint64_t count;     // points in cloud_big / in the current output buffer
    uint32_t status;    // OR of O3DR_STATUS_* bits
    uint32_t pad;
};

// ---- launchers (o3dr_kernels.hip).  All are asynchronous on `s`. -------------------------------
struct Profiler;  // o3dr_api.hip

void launch_minmax_init(Profiler* pf, hipStream_t s, float* mm, int64_t mm_stride, int slot, uint32_t* n_kp, int frames);
// keypoint pass of `frames` frames (one workgroup each); kp_off = nullptr: a single frame with n_kp keypoints
void launch_keypoint_pass(Profiler* pf, hipStream_t s, const ReprojectArgs& a, const float* kp_xy, int n_kp,
                          o3dr_point* out, uint32_t* n_kp_out, float* mm, const int32_t* kp_off = nullptr, int frames = 1);
void launch_reproject(Profiler* pf, hipStream_t s, const ReprojectArgs& a, int frames, o3dr_point* out,
                      uint32_t* tile_cnt, const uint32_t* n_kp, uint32_t* n_valid, float* mm,
                      uint32_t* scan_partial);
// A1 + A2 of a batch with the per-frame grid's index produced in the same pass over the pixels as the points
// (bounding boxes and counts first, then PCL's geometry, then the points): for launch_voxel_grid with
// v.keys_ready = 1.  No keypoint pass (n_kp must hold zeros).
void launch_reproject_fused(Profiler* pf, hipStream_t s, Workspace& ws, const ReprojectArgs& a, int frames, int64_t cap,
                            const float leaf[3], bool conservative_box = true);
void launch_transform(Profiler* pf, hipStream_t s, const o3dr_point* in, int64_t n, const float* T16_host,
                      o3dr_point* out);
int launch_points_minmax(Profiler* pf, hipStream_t s, const o3dr_point* in, int64_t in_fstride,
                         const uint32_t* n_dev, int frames, int64_t cap, int64_t mm_stride, float* mm);
// voxel grid over `frames` independent clouds (cloud f = in + f*in_fstride, n_dev[f] points);
// results are appended at out_base[cc->count + ...] in frame order and cc->count is advanced.
struct VoxelArgs {
    const o3dr_point* in;
    int64_t in_fstride;
    const uint32_t* n_dev;
    int frames;
    int64_t cap;
    float leaf[3];
    uint32_t min_points;
    float z_offset;
    o3dr_point* out_base;
    CloudCounters* cc;
    int passthrough;  // dont_downsample: append the input unchanged
    int mm_used;      // bounding-box slots to fold per frame
    SortStats* stats; // optional device statistics
    int use_runs;     // whole-cloud calls (frames == 1): sort runs of consecutive points of one voxel group instead of points
                      // when they are long enough (decided on the device; 2: whenever the result slots allow it); needs
                      // ws.grp_slots result slots in ws.pts, which must not be the input
    float* cloud_box = nullptr;  // device, 6 floats: running bounding box of out_base's cloud, extended by this call
    CloudHeads cloud_heads = {nullptr, {0.f, 0.f, 0.f}, 0.f, nullptr};  // appending to cloud_big: record the group-run heads of what is appended
    const uint8_t* heads_in = nullptr;  // whole-cloud call on a cloud whose group-run heads are recorded already (for this leaf)
    int keys_ready = 0;  // launch_reproject_fused ran: ws.geom and the indices in ws.keys[0] exist
    int test_corrupt = 0;  // o3dr_test_corrupt_next_gather: poison one sorted payload before the gather (guard test)
};
constexpr int kBoxFoldBlocks = 1024;  // workgroups (and partial boxes) of the running-bounding-box fold
void launch_voxel_grid(Profiler* pf, hipStream_t s, Workspace& ws, const VoxelArgs& v);
void launch_set_counts(Profiler* pf, hipStream_t s, uint32_t* n_dev, uint32_t value, int frames);
// Small clouds (at most kSmallMax points) in ONE launch of ONE workgroup (kernels/small.inc).
//  launch_small_frame: A1 (+ A2 by a.xf_mode) of one frame, keypoints first; downsample != 0: followed by the voxel grid
//    with `leaf` (pts: scratch for kSmallMax points); out / cc receive the result and its size; n_out_dev / box_out6
//    (optional): the size as a plain word and the bounding box of the points, for launch_sor.
//  launch_small_voxel: the voxel grid of a cloud that already exists (n_in_dev == nullptr: n_in points), over its own
//    bounding box or over box6 (device).
void launch_small_frame(Profiler* pf, hipStream_t s, const ReprojectArgs& a, const float* kp_xy, int n_kp, int downsample,
                        const float leaf[3], o3dr_point* pts, o3dr_point* out, CloudCounters* cc, uint32_t* n_out_dev,
                        float* box_out6);
// statistical outlier removal of ONE cloud of at most kSmallMax points (n_dev[0] of them, cap >= that): preparation and
// closing stages as one workgroup each around the unchanged search kernels (5 launches instead of 30); inliers -> out in
// input order, their number -> n_out_dev[0]
void launch_sor_small(Profiler* pf, hipStream_t s, Workspace& ws, const o3dr_point* in, const uint32_t* n_dev, int64_t cap,
                      double stddev_mul, o3dr_point* out, uint32_t* n_out_dev);
void launch_small_voxel(Profiler* pf, hipStream_t s, const o3dr_point* in, const uint32_t* n_in_dev, uint32_t n_in,
                        const float* box6, const float leaf[3], uint32_t min_points, float z_offset, o3dr_point* out,
                        CloudCounters* cc);
// cv::bilateralFilter on u8 images; tab = color_weight[256] | space_weight[maxk] | tile offsets [maxk] (device)
constexpr int kBilMaxRadius = 64;
void launch_bilateral(Profiler* pf, hipStream_t s, const uint8_t* src, int64_t src_pitch, int64_t src_fstride, int rows,
                      int cols, int frames, int radius, int maxk, const float* tab, uint8_t* dst, int64_t dst_pitch,
                      int64_t dst_fstride);
int bilateral_tile_width(int radius);
void launch_disp_variance(Profiler* pf, hipStream_t s, const uint8_t* disp, int64_t pitch, int64_t fstride, int rows, int cols,
                          int frames, int bb, int cs, double min_disp, unsigned long long* hist, double* var_out);
void launch_bbox(Profiler* pf, hipStream_t s, const float* mm, int used, float* out6);
// statistical outlier removal of `frames` clouds (cloud f = in + f*in_fstride, n_dev[f] points, bounding boxes in ws.mm
// slots [0, mm_used) of frame f): inliers -> out + f*out_fstride in input order, counts -> n_out_dev[f], their bounding
// boxes -> ws.mm slots [0, returned value) of frame f.  The sort buffers of ws (keys/vals/hist/geom) are used.
int launch_sor(Profiler* pf, hipStream_t s, Workspace& ws, const o3dr_point* in, int64_t in_fstride, const uint32_t* n_dev,
               int frames, int64_t cap, int mm_used, double stddev_mul, o3dr_point* out, int64_t out_fstride,
               uint32_t* n_out_dev);
void launch_partition(Profiler* pf, hipStream_t s, Workspace& ws, const VoxelArgs& v, int n_parts, o3dr_point* out,
                      uint64_t* counts_dev, uint32_t* overflow_dev, const void* hdrs_dev = nullptr, int n_hdrs = 0);
// its two halves: slice sizes without moving anything (the (part, tile) table stays in ws for the second half), then the move
// with part p's records shifted by part_shift_dev[p] against the plain "parts one after the other" layout
void launch_partition_count(Profiler* pf, hipStream_t s, Workspace& ws, const VoxelArgs& v, int n_parts, uint64_t* counts_dev,
                            uint32_t* overflow_dev, const void* hdrs_dev = nullptr, int n_hdrs = 0);
void launch_partition_move(Profiler* pf, hipStream_t s, Workspace& ws, const VoxelArgs& v, int n_parts, o3dr_point* out,
                           const int64_t* part_shift_dev);
// the exchange's small data, kept on the device (kernels/multigpu.inc)
void launch_pack_header(hipStream_t s, const float* box6_dev, const CloudCounters* cc, void* hdr32_dev);
void launch_count_from_cc(hipStream_t s, const CloudCounters* cc, uint32_t* n_dev);
void launch_set_cloud_count(hipStream_t s, CloudCounters* cc, uint64_t n);

}  // namespace o3dr
