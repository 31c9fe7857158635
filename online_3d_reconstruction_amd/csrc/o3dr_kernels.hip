// o3dr_kernels.hip — hand-written gfx950 kernels of the reconstruction hot path.
//
//   K1  k_reproject_count / k_reproject_emit   A1+A2: (u,v,disparity) -> Q -> SE(3) -> ordered cloud
//       (batched A6: k_reproject_bbox_count, then k_reproject_emit also writes the voxel index and counts pass-0 digits)
//   K2  k_voxel_geom / k_voxel_keys_*          A4 steps 1-5: PCL VoxelGrid geometry and linear index
//       k_radix_hist / k_radix_scatter_lane    A4 step 6: stable LSD radix sort of (index, point id)
//       k_run_heads / k_run_starts / k_centroid  A4 steps 7-8: runs -> ordered fp32 centroid
//
// Everything here is HBM-bound integer/byte/fp32 work: no MFMA.  The whole translation unit is
// compiled with -ffp-contract=off because the reference arithmetic (unoptimised x86 build,
// SURVEY.md section 7 "hard parts") never fuses a multiply-add, and a 1-ulp difference in a world
// coordinate flips voxel occupancy.
//
// Reference lines each kernel follows are cited at the kernel.  Wavefront size is 64.
//
// The device code lives in kernels/*.inc, one file per subsystem, all included below into this one translation
// unit (a kernel and the launcher that names it must share a TU unless device code is built relocatable):
//   util         wave/workgroup scans and reductions, helpers of the sort and run records
//   reproject    K1: count / bbox + count / emit (+ index, first histogram) / keypoint pass / in-place transform
//   bookkeeping  slot initialisation, exclusive scans, bounding box of a cloud
//   prepass      bilateral filter and variance gate on the disparity image
//   voxel_index  PCL VoxelGrid geometry, linear indices (+ fused first histogram / run-head counts)
//   radix_sort   digit histograms and the stable lane-counting scatter
//   voxel_runs   run heads/starts, min_points filter, centroid kernels, run-compressed variants, running bbox
//   multigpu     bounding-box fold, index-slice partition
//   sor          statistical outlier removal
//   small        clouds of at most kSmallMax points: the whole path in one launch of one workgroup
// The launchers follow in this file.
#include <string.h>

#include "o3dr_device.h"
#include "o3dr_profile.h"

namespace o3dr {

#include "kernels/util.inc"
#include "kernels/reproject.inc"
#include "kernels/bookkeeping.inc"
#include "kernels/prepass.inc"
#include "kernels/voxel_index.inc"
#include "kernels/radix_sort.inc"
#include "kernels/voxel_runs.inc"
#include "kernels/multigpu.inc"
#include "kernels/sor.inc"
#include "kernels/small.inc"

// =================================================================================================
// launchers
// =================================================================================================
static inline int cdiv64(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// exclusive scan of `frames` rows of length L in place; totals[f] = row sum (+ add[f]).
// With geom != nullptr the rows are radix histograms of pass `pass` (live length per frame).
static void launch_scan(hipStream_t s, uint32_t* data, int64_t L, int64_t row_stride, int frames, uint32_t* totals,
                        const uint32_t* add, uint32_t* partial, const VoxelGeom* geom = nullptr, int pass = 0,
                        int n_tiles = 0)
{
    if (L <= 4 * kScanChunk) {  // one 1024-thread workgroup per row is faster than three launches up to ~16k words
        k_scan_rows<<<frames, 1024, 0, s>>>(data, L, row_stride, totals, add, geom, pass, n_tiles, 0);
        return;
    }
    const int n_chunks = cdiv64(L, kScanChunk);
    k_scan_chunk_sums<<<dim3(n_chunks, frames), 256, 0, s>>>(data, L, row_stride, n_chunks, partial, geom, pass, n_tiles);
    k_scan_rows<<<frames, 1024, 0, s>>>(partial, L, n_chunks, totals, add, geom, pass, n_tiles, 1);
    k_scan_chunk_apply<<<dim3(n_chunks, frames), 256, 0, s>>>(data, L, row_stride, n_chunks, partial, geom, pass, n_tiles);
}

void launch_minmax_init(Profiler* pf, hipStream_t s, float* mm, int64_t mm_stride, int slot, uint32_t* n_kp, int frames)
{
    ProfScope ps(pf, O3DR_K_OTHER, s);
    k_minmax_init<<<cdiv64(frames * 6, 256), 256, 0, s>>>(mm, mm_stride, slot, n_kp, frames);
}

void launch_set_counts(Profiler* pf, hipStream_t s, uint32_t* n_dev, uint32_t value, int frames)
{
    ProfScope ps(pf, O3DR_K_OTHER, s);
    k_set_counts<<<cdiv64(frames, 256), 256, 0, s>>>(n_dev, value, frames);
}

void launch_small_frame(Profiler* pf, hipStream_t s, const ReprojectArgs& a, const float* kp_xy, int n_kp, int downsample,
                        const float leaf[3], o3dr_point* pts, o3dr_point* out, CloudCounters* cc, uint32_t* n_out_dev,
                        float* box_out6)
{
    ProfScope ps(pf, O3DR_K_OTHER, s);
    SmallArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.reproject = 1;
    sa.kp_xy = kp_xy;
    sa.n_kp = n_kp;
    sa.downsample = downsample;
    for (int i = 0; i < 3; ++i) sa.leaf[i] = leaf[i];
    sa.pts = pts;
    sa.out = out;
    sa.cc = cc;
    sa.n_out_dev = n_out_dev;
    sa.box_out6 = box_out6;
    k_small<<<1, kSmallThreads, 0, s>>>(a, sa);
}

void launch_sor_small(Profiler* pf, hipStream_t s, Workspace& ws, const o3dr_point* in, const uint32_t* n_dev, int64_t cap,
                      double stddev_mul, o3dr_point* out, uint32_t* n_out_dev)
{
    ProfScope ps(pf, O3DR_K_OTHER, s);
    const int64_t cell_stride = (int64_t)ws.sor_max_cells + 1;
    uint32_t max_cells = (uint32_t)(cap / 2 > 1024 ? cap / 2 : 1024);  // (as launch_sor: the grid is only a search structure)
    if (max_cells > ws.sor_max_cells) max_cells = ws.sor_max_cells;
    k_sor_small_prep<<<1, kSmallThreads, 0, s>>>(in, n_dev, max_cells, ws.sor_geom, ws.geom, ws.vals[0], ws.sor_cell_first, ws.sor_xyz,
                                                 ws.sor_left_cnt);
    k_sor_knn<<<dim3(cdiv64(cap, kWave), 1), kWave, 0, s>>>(ws.sor_xyz, ws.vals[0], ws.vals[1], ws.geom, ws.sor_cell_first, cell_stride,
                                                           ws.sor_geom, cap, ws.sor_dist, ws.sor_left, ws.sor_left_cnt);
    int lg = cdiv64(cap, kWave * kSorLeftWaves);
    if (lg > 1024) lg = 1024;
    int zg = cdiv64((int64_t)max_cells, 256);
    if (zg > 256) zg = 256;
    k_sor_cell_z<<<dim3(zg, 1), 256, 0, s>>>(ws.sor_xyz, ws.sor_cell_first, cell_stride, ws.sor_geom, cap, ws.sor_left_cnt, ws.sor_cell_z);
    k_sor_knn_left<<<dim3(lg, 1), kSorLeftWaves * kWave, 0, s>>>(ws.sor_xyz, ws.vals[0], ws.vals[1], ws.geom, ws.sor_cell_first, cell_stride,
                                                                ws.sor_geom, cap, ws.sor_dist, ws.sor_left, ws.sor_left_cnt, ws.sor_cell_z);
    k_sor_small_tail<<<1, kSmallThreads, 0, s>>>(in, ws.sor_dist, ws.sor_geom, stddev_mul, out, n_out_dev);
}

void launch_small_voxel(Profiler* pf, hipStream_t s, const o3dr_point* in, const uint32_t* n_in_dev, uint32_t n_in,
                        const float* box6, const float leaf[3], uint32_t min_points, float z_offset, o3dr_point* out,
                        CloudCounters* cc)
{
    ProfScope ps(pf, O3DR_K_OTHER, s);
    ReprojectArgs a;
    memset(&a, 0, sizeof a);
    SmallArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.downsample = 1;
    sa.in = in;
    sa.n_in_dev = n_in_dev;
    sa.n_in = n_in;
    sa.box6 = box6;
    for (int i = 0; i < 3; ++i) sa.leaf[i] = leaf[i];
    sa.min_points = min_points;
    sa.z_offset = z_offset;
    sa.out = out;
    sa.cc = cc;
    k_small<<<1, kSmallThreads, 0, s>>>(a, sa);
}

void launch_keypoint_pass(Profiler* pf, hipStream_t s, const ReprojectArgs& a, const float* kp_xy, int n_kp,
                          o3dr_point* out, uint32_t* n_kp_out, float* mm, const int32_t* kp_off, int frames)
{
    ProfScope ps(pf, O3DR_K_OTHER, s);
    k_keypoint_pass<<<frames, 256, 0, s>>>(a, kp_xy, n_kp, kp_off, out, n_kp_out, mm);
}

void launch_reproject(Profiler* pf, hipStream_t s, const ReprojectArgs& a, int frames, o3dr_point* out,
                      uint32_t* tile_cnt, const uint32_t* n_kp, uint32_t* n_valid, float* mm,
                      uint32_t* scan_partial)
{
    if (a.n_tiles <= 0) {  // jump_pixels == 0: keypoints only
        ProfScope ps(pf, O3DR_K_OTHER, s);
        (void)hipMemcpyAsync(n_valid, n_kp, sizeof(uint32_t) * frames, hipMemcpyDeviceToDevice, s);
        return;
    }
    const dim3 grid(a.n_tiles, frames);
    {
        ProfScope ps(pf, O3DR_K_COUNT, s);
        if (a.disp_f64)
            k_reproject_count<true><<<grid, kEmitThreads, 0, s>>>(a, tile_cnt);
        else
            k_reproject_count<false><<<grid, kEmitThreads, 0, s>>>(a, tile_cnt);
    }
    {
        ProfScope ps(pf, O3DR_K_OTHER, s);
        launch_scan(s, tile_cnt, a.n_tiles, a.n_tiles, frames, n_valid, n_kp, scan_partial);
    }
    {
        ProfScope ps(pf, O3DR_K_REPROJECT, s);
        if (a.disp_f64)
            k_reproject_emit<true, false><<<grid, kEmitThreads, 0, s>>>(a, out, tile_cnt, n_kp, mm, nullptr, 0, nullptr);
        else
            k_reproject_emit<false, false><<<grid, kEmitThreads, 0, s>>>(a, out, tile_cnt, n_kp, mm, nullptr, 0, nullptr);
    }
}

static inline int xcd_grid(int64_t n) { return (int)((n + kXcds - 1) / kXcds * kXcds); }  // see xcd_chunk_item
void launch_reproject_fused(Profiler* pf, hipStream_t s, Workspace& ws, const ReprojectArgs& a, int frames, int64_t cap,
                            const float leaf[3], bool conservative_box)
{
    const dim3 grid(a.n_tiles, frames);
    // rectified-stereo Q and byte disparities: counts + a conservative box from the corners of (x, y, disparity) ranges
    // (k_reproject_count_cbox); the exact box only for frames whose conservative one trips PCL's overflow guard
    const bool cbox = a.lut != nullptr && !a.disp_f64 && conservative_box;
    const dim3 cgrid(cdiv64(a.n_tiles, kCountTiles), frames);
    {
        ProfScope ps(pf, O3DR_K_COUNT, s);
        if (cbox)
            k_reproject_count_cbox<kCboxTiles><<<dim3(cdiv64(a.n_tiles, kCboxTiles), frames), kEmitThreads, 0, s>>>(a, ws.tile_cnt, ws.mm);
        else if (a.disp_f64)
            k_reproject_bbox_count<true><<<cgrid, kEmitThreads, 0, s>>>(a, ws.tile_cnt, ws.mm, nullptr);
        else
            k_reproject_bbox_count<false><<<cgrid, kEmitThreads, 0, s>>>(a, ws.tile_cnt, ws.mm, nullptr);
    }
    {
        ProfScope ps(pf, O3DR_K_OTHER, s);
        launch_scan(s, ws.tile_cnt, a.n_tiles, a.n_tiles, frames, ws.n_valid, ws.n_kp, ws.scan_partial);
        k_voxel_geom<<<frames, 256, 0, s>>>(ws.mm, ws.mm_stride, a.n_tiles + 1, ws.n_valid, leaf[0], leaf[1], leaf[2], 0.f, ws.geom);
    }
    if (cbox) {
        ProfScope ps(pf, O3DR_K_COUNT, s);
        k_reproject_bbox_count<false><<<cgrid, kEmitThreads, 0, s>>>(a, ws.tile_cnt, ws.mm, ws.geom);
        k_voxel_geom<<<frames, 256, 0, s>>>(ws.mm, ws.mm_stride, a.n_tiles + 1, ws.n_valid, leaf[0], leaf[1], leaf[2], 0.f, ws.geom, 1);
    }
    {
        ProfScope ps(pf, O3DR_K_REPROJECT, s);
        if (a.disp_f64)
            k_reproject_emit<true, true><<<grid, kEmitThreads, 0, s>>>(a, ws.pts, ws.tile_cnt, ws.n_kp, ws.mm, ws.geom, cap, ws.keys[0]);
        else
            k_reproject_emit<false, true><<<grid, kEmitThreads, 0, s>>>(a, ws.pts, ws.tile_cnt, ws.n_kp, ws.mm, ws.geom, cap, ws.keys[0]);
    }
}

void launch_transform(Profiler* pf, hipStream_t s, const o3dr_point* in, int64_t n, const float* T16_host,
                      o3dr_point* out)
{
    if (n <= 0) return;
    Mat34 T;
    for (int i = 0; i < 12; ++i) T.m[i] = T16_host[i];
    ProfScope ps(pf, O3DR_K_OTHER, s);
    k_transform<<<cdiv64(n, kPtThreads), kPtThreads, 0, s>>>(in, n, T, out);
}

int launch_points_minmax(Profiler* pf, hipStream_t s, const o3dr_point* in, int64_t in_fstride,
                         const uint32_t* n_dev, int frames, int64_t cap, int64_t mm_stride, float* mm)
{
    if (cap <= 0) return 0;
    int nblk = cdiv64(cap, kPtThreads * 4);
    if (nblk > kMinmaxBlocks) nblk = kMinmaxBlocks;
    ProfScope ps(pf, O3DR_K_OTHER, s);
    k_points_minmax<<<dim3(nblk, frames), kPtThreads, 0, s>>>(in, in_fstride, n_dev, mm_stride, mm);
    return nblk;  // slots written per frame
}

// The voxel grid proper.  Expects ws.mm slots [0, v.mm_used) of every frame to hold bounding boxes of
// the (un-offset) inputs.
void launch_voxel_grid(Profiler* pf, hipStream_t s, Workspace& ws, const VoxelArgs& v)
{
    const int F = v.frames;
    const int64_t cap = v.cap;
    const int n_sort_tiles = cdiv64(cap, kSortTile);
    const int n_seg_tiles = cdiv64(cap, kSegTile);
    // tile-major histogram rows (hist_at in kernels/radix_sort.inc) where a frame's row fits the one-workgroup scan's LDS
    const size_t tm_lds = (size_t)n_sort_tiles * (kMaxRadix + 1) * sizeof(uint32_t);
    const int tm = tm_lds <= 48 * 1024 ? 1 : 0;
    if (!v.keys_ready) {
        ProfScope ps(pf, O3DR_K_OTHER, s);
        k_voxel_geom<<<F, 256, 0, s>>>(ws.mm, ws.mm_stride, v.mm_used, v.n_dev, v.leaf[0], v.leaf[1], v.leaf[2],
                                       v.z_offset, ws.geom);
    }
    uint32_t* n_keep = nullptr;
    // grouped records: whole-cloud calls only (one cloud; its result slots live in ws.pts)
    // (below ~a million points the extra launches cost more than the point sort saves; use_runs == 2 still forces it)
    const bool use_runs = v.use_runs && !v.passthrough && F == 1 && v.in != ws.pts && (v.use_runs > 1 || cap >= kGroupMinCloud);
    const int64_t grp_slots = use_runs ? ws.grp_slots : 0;
    const int64_t max_groups = grp_slots / kGroupCells;
    // what the sort and the run/cell kernels count (group runs or points)
    const VoxelGeom* sort_geom = use_runs ? ws.geom_runs : ws.geom;
    if (!v.passthrough && cap > 0) {
        const dim3 grid(n_sort_tiles, F);
        if (use_runs) {
            // the heads of the group runs (runs of consecutive points of one voxel group) per segment tile: recorded
            // while the cloud was appended to, or from one read of the points
            const uint8_t* hb = v.heads_in ? v.heads_in : ws.head_bits;
            {
                ProfScope ps(pf, O3DR_K_KEYGEN, s);
                if (v.heads_in)
                    k_head_counts<<<cdiv64(n_seg_tiles, 4), 256, 0, s>>>(v.heads_in, ws.geom, n_seg_tiles, ws.seg_cnt);
                else
                    k_group_heads<<<n_seg_tiles, 256, 0, s>>>(v.in, ws.geom, v.z_offset, n_seg_tiles, ws.seg_cnt, ws.head_bits);
            }
            {
                ProfScope ps(pf, O3DR_K_OTHER, s);
                launch_scan(s, ws.seg_cnt, n_seg_tiles, n_seg_tiles, F, ws.n_runs, nullptr, ws.scan_partial);
            }
            {
                ProfScope ps(pf, O3DR_K_SEGMENT, s);
                // group runs or points?  (decided per cloud on the device; use_runs == 2: group runs whenever they fit)
                k_run_geom<<<cdiv64(F, 64), 64, 0, s>>>(ws.geom, ws.n_runs, F, ws.geom_runs, v.use_runs > 1 ? 1 : 0, grp_slots);
                // (group, run id) records in buffer 1
                k_run_starts<<<dim3(cdiv64(n_seg_tiles, kStartWaves), F), kStartWaves * kWave, 0, s>>>(
                    ws.keys[0], ws.keys[1], cap, ws.geom, n_seg_tiles, ws.seg_cnt, ws.n_runs, ws.run_start, 0, ws.keys[1],
                    ws.geom_runs, v.in, v.z_offset, hb);
            }
            {
                ProfScope ps(pf, O3DR_K_KEYGEN, s);  // a cloud left to the point sort needs PCL's index per point after all
                k_voxel_keys_hist0<<<grid, kSortThreads, 0, s>>>(v.in, v.in_fstride, ws.geom, v.z_offset, cap, ws.keys[0],
                                                                 n_sort_tiles, ws.hist, ws.hist_part, ws.geom_runs, tm);
            }
        } else if (!v.keys_ready) {  // PCL's index per point and the histogram of the first radix pass
            ProfScope ps(pf, O3DR_K_KEYGEN, s);
            k_voxel_keys_hist0<<<grid, kSortThreads, 0, s>>>(v.in, v.in_fstride, ws.geom, v.z_offset, cap, ws.keys[0],
                                                             n_sort_tiles, ws.hist, ws.hist_part, nullptr, tm);
        }
        // always kMaxPasses launch groups; frames whose index needs fewer passes drop out on the device
        const int64_t hist_row = (int64_t)kMaxRadix * n_sort_tiles;

        for (int pass = 0; pass < kMaxPasses; ++pass) {
            if (!(pass == 0 && !use_runs && !v.keys_ready)) {  // (k_voxel_keys_hist0 counted the first pass's digits)
                ProfScope ps(pf, O3DR_K_SORT_HIST, s);
                k_radix_hist<<<grid, kSortThreads, 0, s>>>(ws.keys[0], ws.keys[1], cap, sort_geom, pass, n_sort_tiles,
                                                          ws.hist, ws.hist_part, tm);
            }
            {
                ProfScope ps(pf, O3DR_K_OTHER, s);
                if (tm)
                    k_scan_hist_tm<<<F, 1024, tm_lds, s>>>(ws.hist, sort_geom, pass, n_sort_tiles);
                else
                    launch_scan(s, ws.hist, hist_row, hist_row, F, nullptr, nullptr, ws.scan_partial, sort_geom, pass,
                                n_sort_tiles);
            }
            {
                ProfScope ps(pf, O3DR_K_SORT_SCATTER, s);
                k_radix_scatter_lane<<<dim3(xcd_grid((int64_t)n_sort_tiles * kScatParts), F), kScatThreads, 0, s>>>(
                    ws.keys[0], ws.vals[0], ws.keys[1], ws.vals[1], cap, sort_geom, pass, n_sort_tiles, ws.hist, ws.hist_part, tm);
            }
        }
        const dim3 sgrid(n_seg_tiles, F);
        {
            ProfScope ps(pf, O3DR_K_SEGMENT, s);
            k_run_heads<<<dim3(cdiv64(n_seg_tiles, kHeadWaves), F), kHeadWaves * kWave, 0, s>>>(ws.keys[0], ws.keys[1], cap, sort_geom, n_seg_tiles,
                                                                                       ws.seg_cnt, -1, ws.head_bits);
        }
        {
            ProfScope ps(pf, O3DR_K_OTHER, s);
            launch_scan(s, ws.seg_cnt, n_seg_tiles, n_seg_tiles, F, ws.n_vox, nullptr, ws.scan_partial);
        }
        {
            ProfScope ps(pf, O3DR_K_SEGMENT, s);
            k_run_starts<<<dim3(cdiv64(n_seg_tiles, kStartWaves), F), kStartWaves * kWave, 0, s>>>(ws.keys[0], ws.keys[1], cap, sort_geom, n_seg_tiles, ws.seg_cnt, ws.n_vox,
                                              ws.seg_start, -1, nullptr, nullptr, nullptr, 0.f, ws.head_bits);
        }
        if (v.min_points > 1) {  // (grouped clouds filter inside k_centroid_groups and drop out of these on the device)
            {
                ProfScope ps(pf, O3DR_K_SEGMENT, s);
                k_keep_count<<<sgrid, 256, 0, s>>>(ws.seg_start, cap, sort_geom, ws.n_vox, v.min_points, n_seg_tiles, ws.seg_cnt);
            }
            {
                ProfScope ps(pf, O3DR_K_OTHER, s);
                launch_scan(s, ws.seg_cnt, n_seg_tiles, n_seg_tiles, F, ws.n_out, nullptr, ws.scan_partial);
            }
            {
                ProfScope ps(pf, O3DR_K_SEGMENT, s);
                k_keep_write<<<sgrid, 256, 0, s>>>(ws.seg_start, cap, sort_geom, ws.n_vox, v.min_points, n_seg_tiles, ws.seg_cnt,
                                                  ws.keep_idx);
            }
            n_keep = ws.n_out;
        }
    }
    if (v.test_corrupt && cap > 0 && !v.passthrough)
        k_test_corrupt_payload<<<1, 1, 0, s>>>(ws.vals[0], ws.vals[1], sort_geom);
    if (cap > 0 && use_runs) {
        // one wave per voxel group; the groups' output counts are only known afterwards
        {
            ProfScope ps(pf, O3DR_K_OTHER, s);
            (void)hipMemsetAsync(ws.grp_cnt, 0, sizeof(uint32_t) * (size_t)(max_groups + 1), s);
        }
        {
            ProfScope ps(pf, O3DR_K_CENTROID_RUNS, s);
            int64_t nwg = cdiv64(max_groups < cap ? max_groups : cap, kGroupWaves);
            if (nwg > 16384) nwg = 16384;
            if (nwg < 1) nwg = 1;
            k_centroid_groups<<<(int)nwg, kGroupWaves * kWave, 0, s>>>(
                v.in, ws.keys[0], ws.keys[1], ws.vals[0], ws.vals[1], ws.seg_start, ws.run_start, ws.geom_runs, ws.geom, ws.n_vox,
                v.z_offset, v.min_points, reinterpret_cast<uint4*>(ws.pts), grp_slots, ws.grp_cnt, v.cc);
        }
        {
            ProfScope ps(pf, O3DR_K_OTHER, s);
            launch_scan(s, ws.grp_cnt, max_groups + 1, max_groups + 1, 1, ws.n_grp_out, nullptr, ws.scan_partial);
        }
    }
    {
        ProfScope ps(pf, O3DR_K_OTHER, s);
        k_frame_offsets<<<1, 256, 0, s>>>(ws.geom, ws.n_vox, n_keep, use_runs ? ws.n_grp_out : nullptr, F, v.passthrough,
                                          ws.n_out, ws.out_off, v.cc, v.stats, sort_geom);
    }
    const int nbx = cdiv64(cap, kPtThreads);
    if (cap > 0 && use_runs) {
        ProfScope ps(pf, O3DR_K_CENTROID_RUNS, s);
        int64_t nwg = cdiv64(cap, 256);
        if (nwg > 8192) nwg = 8192;
        k_group_compact<<<(int)nwg, 256, 0, s>>>(v.in, ws.geom_runs, ws.geom, ws.n_vox, ws.grp_cnt,
                                                reinterpret_cast<const uint4*>(ws.pts), ws.out_off, v.z_offset, v.out_base);
    }
    if (cap > 0) {
        ProfScope ps(pf, O3DR_K_CENTROID, s);
        float* out_mm = v.cloud_box ? ws.out_mm : nullptr;
        CloudHeads heads = v.cloud_heads;
        heads.wave_gc = ws.wave_gc;
        const uint32_t* keep = (v.min_points > 1 && !v.passthrough) ? ws.keep_idx : nullptr;
        if (use_runs)  // only for clouds k_run_geom left to the point sort: a small looping grid
            k_centroid<true><<<dim3(nbx < 4096 ? nbx : 4096, F), kPtThreads, 0, s>>>(
                v.in, v.in_fstride, ws.vals[0], ws.vals[1], cap, ws.seg_start, keep, ws.geom_runs, ws.n_out, ws.out_off,
                v.z_offset, v.passthrough, v.out_base, out_mm, nbx, v.cc, CloudHeads{nullptr, {0.f, 0.f, 0.f}, 0.f, nullptr});
        else
            k_centroid<false><<<dim3(xcd_grid(nbx), F), kPtThreads, 0, s>>>(
                v.in, v.in_fstride, ws.vals[0], ws.vals[1], cap, ws.seg_start, keep, ws.geom, ws.n_out, ws.out_off,
                v.z_offset, v.passthrough, v.out_base, out_mm, nbx, v.cc, heads);
        if (!use_runs && v.cloud_heads.flags) {  // the first point of every wave against the last one of the wave before it
            const int wpf = nbx * (kPtThreads / 64);
            k_cloud_heads_fix<<<dim3(cdiv64(wpf, 256), F), 256, 0, s>>>(ws.n_out, ws.out_off, wpf, heads);
        }
    }
    if (v.cloud_box && cap > 0 && !use_runs) {
        ProfScope ps(pf, O3DR_K_OTHER, s);
        const int nbx = cdiv64(cap, kPtThreads) * (kPtThreads / 64);
        k_cloud_bbox_fold<<<kBoxFoldBlocks, 256, 0, s>>>(ws.out_mm, nbx, F, ws.n_out, ws.out_mm_partial);
        k_cloud_bbox_merge<<<1, 384, 0, s>>>(ws.out_mm_partial, kBoxFoldBlocks, v.cloud_box);
    }
}

void launch_bilateral(Profiler* pf, hipStream_t s, const uint8_t* src, int64_t src_pitch, int64_t src_fstride, int rows,
                      int cols, int frames, int radius, int maxk, const float* tab, uint8_t* dst, int64_t dst_pitch,
                      int64_t dst_fstride)
{
    if (rows <= 0 || cols <= 0 || frames <= 0) return;
    const int tiles_x = cdiv64(cols, kBilTX), tiles_y = cdiv64(rows, kBilTY);
    const size_t lds = 1024 + (size_t)(kBilTX + 2 * radius) * (kBilTY + 2 * radius);
    ProfScope ps(pf, O3DR_K_OTHER, s);
    k_bilateral_u8<<<dim3(tiles_x * tiles_y, frames), 256, lds, s>>>(src, src_pitch, src_fstride, rows, cols, radius, maxk, tab,
                                                                    dst, dst_pitch, dst_fstride, tiles_x);
}
int bilateral_tile_width(int radius) { return kBilTX + 2 * radius; }

void launch_disp_variance(Profiler* pf, hipStream_t s, const uint8_t* disp, int64_t pitch, int64_t fstride, int rows, int cols,
                          int frames, int bb, int cs, double min_disp, unsigned long long* hist, double* var_out)
{
    ProfScope ps(pf, O3DR_K_OTHER, s);
    (void)hipMemsetAsync(hist, 0, sizeof(unsigned long long) * 256 * (size_t)frames, s);
    const int roi_rows = rows - 2 * bb;
    if (roi_rows > 0) k_disp_hist<<<dim3(roi_rows, frames), 256, 0, s>>>(disp, pitch, fstride, rows, cols, bb, cs, min_disp, hist);
    k_disp_variance<<<cdiv64(frames, 64), 64, 0, s>>>(hist, frames, rows, cols, bb, cs, var_out);
}

void launch_bbox(Profiler* pf, hipStream_t s, const float* mm, int used, float* out6)
{
    ProfScope ps(pf, O3DR_K_OTHER, s);
    k_bbox_fold<<<1, 256, 0, s>>>(mm, used, out6);
}

// Stable partition of `in` (n points, count also in v.n_dev[0]) by index slice of the voxel grid whose
// bounding box sits in ws.mm slot 0.  out = reordered points, counts_dev[n_parts], overflow_dev = 1 when
// PCL's overflow guard fires for that box (then out = in and every count is 0).
void launch_pack_header(hipStream_t s, const float* box6_dev, const CloudCounters* cc, void* hdr32_dev)
{
    k_pack_header<<<1, 1, 0, s>>>(box6_dev, cc, reinterpret_cast<CloudHeader*>(hdr32_dev));
}
void launch_count_from_cc(hipStream_t s, const CloudCounters* cc, uint32_t* n_dev) { k_count_from_cc<<<1, 1, 0, s>>>(cc, n_dev); }
void launch_set_cloud_count(hipStream_t s, CloudCounters* cc, uint64_t n) { k_set_cloud_count<<<1, 1, 0, s>>>(cc, n); }

// hdrs_dev (optional): the box in ws.mm slot 0 is first folded from n_hdrs 32-byte rank headers in HBM
// the two halves of the stable partition by index slice: counts per (part, tile) + their scan (ws.hist, ws.geom: kept for the
// second half) and the slice sizes; then the move, optionally with a shift per part (k_part_move)
void launch_partition_count(Profiler* pf, hipStream_t s, Workspace& ws, const VoxelArgs& v, int n_parts, uint64_t* counts_dev,
                            uint32_t* overflow_dev, const void* hdrs_dev, int n_hdrs)
{
    const int64_t cap = v.cap;
    const int n_sort_tiles = cdiv64(cap, kSortTile);
    const int64_t hist_row = (int64_t)kMaxRadix * n_sort_tiles;
    ProfScope ps(pf, O3DR_K_OTHER, s);
    if (hdrs_dev) k_fold_headers<<<1, 1, 0, s>>>(reinterpret_cast<const CloudHeader*>(hdrs_dev), n_hdrs, ws.mm);
    k_voxel_geom<<<1, 256, 0, s>>>(ws.mm, ws.mm_stride, 1, v.n_dev, v.leaf[0], v.leaf[1], v.leaf[2], v.z_offset, ws.geom);
    // count per (part, tile), scan; n_parts <= kMaxRadix
    k_part_plan<<<1, 1, 0, s>>>(ws.geom, n_parts);
    k_part_count<<<n_sort_tiles, kSortThreads, 0, s>>>(v.in, ws.geom, v.z_offset, n_parts, n_sort_tiles, ws.hist);
    launch_scan(s, ws.hist, hist_row, hist_row, 1, nullptr, nullptr, ws.scan_partial, ws.geom, 0, n_sort_tiles);
    k_part_counts_scanned<<<cdiv64(n_parts, 64), 64, 0, s>>>(ws.hist, ws.geom, n_parts, n_sort_tiles, counts_dev, overflow_dev);
}
void launch_partition_move(Profiler* pf, hipStream_t s, Workspace& ws, const VoxelArgs& v, int n_parts, o3dr_point* out,
                           const int64_t* part_shift_dev)
{
    ProfScope ps(pf, O3DR_K_OTHER, s);
    const int n_sort_tiles = cdiv64(v.cap, kSortTile);
    k_part_move<<<n_sort_tiles, kSortThreads, 0, s>>>(v.in, ws.geom, v.z_offset, n_parts, n_sort_tiles, ws.hist, out, part_shift_dev);
}
void launch_partition(Profiler* pf, hipStream_t s, Workspace& ws, const VoxelArgs& v, int n_parts, o3dr_point* out,
                      uint64_t* counts_dev, uint32_t* overflow_dev, const void* hdrs_dev, int n_hdrs)
{
    // (two reads and one write of the cloud)
    launch_partition_count(pf, s, ws, v, n_parts, counts_dev, overflow_dev, hdrs_dev, n_hdrs);
    launch_partition_move(pf, s, ws, v, n_parts, out, nullptr);
}

// Statistical outlier removal of a batch of clouds (see o3dr_device.h).  ws.sor_* are laid out for ws.sor_cap points
// per frame; cap (<= ws.sor_cap) sizes the grids.
int launch_sor(Profiler* pf, hipStream_t s, Workspace& ws, const o3dr_point* in, int64_t in_fstride, const uint32_t* n_dev,
               int frames, int64_t cap, int mm_used, double stddev_mul, o3dr_point* out, int64_t out_fstride,
               uint32_t* n_out_dev)
{
    ProfScope ps(pf, O3DR_K_OTHER, s);
    const int F = frames;
    const int n_sort_tiles = cdiv64(cap, kSortTile);
    const int64_t hist_row = (int64_t)kMaxRadix * n_sort_tiles;
    const size_t tm_lds = (size_t)n_sort_tiles * (kMaxRadix + 1) * sizeof(uint32_t);
    const int tm = tm_lds <= 48 * 1024 ? 1 : 0;
    const int n_tiles = cdiv64(cap, 1024);
    const int64_t cell_stride = (int64_t)ws.sor_max_cells + 1;
    // The search grid never has more cells than points / 2 (and at least 1024): the cell ids of clouds of at most `cap`
    // points need ceil(log2(max_cells) / 7) sort passes, and only those are launched (a --jump_pixels 15 frame: 2 instead of
    // 5 launch groups that find nothing to do).  The grid is a search structure: the neighbours found do not depend on it.
    uint32_t max_cells = (uint32_t)(cap / 2 > 1024 ? cap / 2 : 1024);
    if (max_cells > ws.sor_max_cells) max_cells = ws.sor_max_cells;
    int cell_bits = 1;
    while ((1u << cell_bits) < max_cells) ++cell_bits;
    const int sort_passes = (cell_bits + kMaxRadixBits - 1) / kMaxRadixBits;
    k_sor_plan<<<F, 256, 0, s>>>(ws.mm, ws.mm_stride, mm_used, n_dev, max_cells, ws.sor_geom, ws.geom);
    (void)hipMemsetAsync(ws.sor_cell_first, 0, (size_t)F * (size_t)cell_stride * 4, s);
    k_sor_cells<<<dim3(cdiv64(cap, 256), F), 256, 0, s>>>(in, in_fstride, ws.sor_geom, cap, ws.keys[0]);
    for (int pass = 0; pass < sort_passes && pass < kMaxPasses; ++pass) {
        k_radix_hist<<<dim3(n_sort_tiles, F), kSortThreads, 0, s>>>(ws.keys[0], ws.keys[1], cap, ws.geom, pass, n_sort_tiles, ws.hist,
                                                                    ws.hist_part, tm);
        if (tm)
            k_scan_hist_tm<<<F, 1024, tm_lds, s>>>(ws.hist, ws.geom, pass, n_sort_tiles);
        else
            launch_scan(s, ws.hist, hist_row, hist_row, F, nullptr, nullptr, ws.scan_partial, ws.geom, pass, n_sort_tiles);
        k_radix_scatter_lane<<<dim3(xcd_grid((int64_t)n_sort_tiles * kScatParts), F), kScatThreads, 0, s>>>(
            ws.keys[0], ws.vals[0], ws.keys[1], ws.vals[1], cap, ws.geom, pass, n_sort_tiles, ws.hist, ws.hist_part, tm);
    }
    k_sor_cell_counts<<<dim3(cdiv64(cap, 256), F), 256, 0, s>>>(ws.keys[0], ws.keys[1], ws.geom, ws.sor_geom, cap, cell_stride, ws.sor_cell_first);
    launch_scan(s, ws.sor_cell_first, cell_stride, cell_stride, F, nullptr, nullptr, ws.scan_partial);
    k_sor_gather<<<dim3(cdiv64(cap, 256), F), 256, 0, s>>>(in, in_fstride, ws.vals[0], ws.vals[1], ws.sor_geom, ws.geom, cap, ws.sor_xyz);
    (void)hipMemsetAsync(ws.sor_left_cnt, 0, (size_t)F * 4, s);
    k_sor_knn<<<dim3(cdiv64(cap, kWave), F), kWave, 0, s>>>(ws.sor_xyz, ws.vals[0], ws.vals[1], ws.geom, ws.sor_cell_first, cell_stride,
                                                           ws.sor_geom, cap, ws.sor_dist, ws.sor_left, ws.sor_left_cnt);
    {   // the queries the wave-shared search handed over (a looping grid: their number is only known on the device)
        int lg = cdiv64(cap, kWave * kSorLeftWaves);
        if (lg > 1024) lg = 1024;
        int zg = cdiv64((int64_t)ws.sor_max_cells, 256);
        if (zg > 256) zg = 256;
        k_sor_cell_z<<<dim3(zg, F), 256, 0, s>>>(ws.sor_xyz, ws.sor_cell_first, cell_stride, ws.sor_geom, cap, ws.sor_left_cnt, ws.sor_cell_z);
        k_sor_knn_left<<<dim3(lg, F), kSorLeftWaves * kWave, 0, s>>>(ws.sor_xyz, ws.vals[0], ws.vals[1], ws.geom, ws.sor_cell_first, cell_stride,
                                                                    ws.sor_geom, cap, ws.sor_dist, ws.sor_left, ws.sor_left_cnt, ws.sor_cell_z);
    }
    k_sor_partial<<<dim3(kSorStatBlocks, F), 256, 0, s>>>(ws.sor_dist, cap, ws.sor_geom, ws.sor_partial);
    k_sor_threshold<<<cdiv64(F, 64), 64, 0, s>>>(ws.sor_partial, stddev_mul, ws.sor_geom, F);
    k_sor_count<<<dim3(n_tiles, F), 256, 0, s>>>(ws.sor_dist, cap, ws.sor_geom, n_tiles, ws.tile_cnt);
    launch_scan(s, ws.tile_cnt, n_tiles, n_tiles, F, n_out_dev, nullptr, ws.scan_partial);
    k_sor_emit<<<dim3(n_tiles, F), 256, 0, s>>>(in, in_fstride, ws.sor_dist, cap, ws.sor_geom, n_tiles, ws.tile_cnt, out, out_fstride,
                                                ws.mm_stride, ws.mm);
    return n_tiles;
}

}  // namespace o3dr
