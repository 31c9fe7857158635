// o3dr_api.hip — the C ABI of include/o3dr.h on top of the kernels in o3dr_kernels.hip.
//
// Host-side responsibilities only: argument checking, HBM workspaces, staging of host buffers,
// batching of frames, the device-resident cloud_big, error codes.  There is no CPU compute path:
// without a GPU o3dr_ctx_create fails and nothing else can be called.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h>  // types and enumerators only: the functions are resolved with dlsym (no link-time dependency on RCCL)

#include "../../include/o3dr_testing.h"
#include "o3dr_device.h"
#include "o3dr_profile.h"

using namespace o3dr;

// -------------------------------------------------------------------------------------------------
// errors
// -------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* what)
{
    g_err = what;
    return code;
}
#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            char buf_[512];                                                                      \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            g_err = buf_;                                                                        \
            return O3DR_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)
#define CHK(expr)                  \
    do {                           \
        int r_ = (expr);           \
        if (r_ != O3DR_OK) return r_; \
    } while (0)

// -------------------------------------------------------------------------------------------------
// context
// -------------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct o3dr_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    o3dr_params params;
    double Q[16];
    bool has_Q = false;
    QLutEntry* q_lut = nullptr;  // device table for rectified-stereo Q (nullptr: general 4x4 product per pixel)
    bool q_lut_on = false;
    // cv::bilateralFilter tables (colour weights | space weights | tile offsets) for the last (d, sigmas) used
    DevBuf bil_tab, st_blur, st_blur_in, st_hist;
    int bil_d = 0, bil_radius = 0, bil_maxk = 0;
    double bil_sc = 0, bil_ss = 0;
    bool bil_valid = false;
    int max_batch = 256;  // frames per launch group (O3DR_BATCH_FRAMES); also bounded by a workspace budget
    int slab_shift_env = -2;  // O3DR_SLABS=0: plain pixel order in the fused batch path; O3DR_SLABS=sN: slabs of 2^N cells; else automatic
    int exact_box = 0;       // O3DR_EXACT_BOX=1: the batch path always takes the exact bounding box (k_reproject_bbox_count)
    int use_runs = 1;        // O3DR_RUNS=0: whole-cloud voxel grids sort points instead of runs; 2: always runs
    int small_path = 1;      // O3DR_SMALL=0: clouds of at most kSmallMax points take the general path too

    Workspace ws;
    size_t ws_elems = 0;   // frames*(cap+1) the per-point arrays were allocated for
    size_t ws_pts_elems = 0;
    int ws_frames = 0;
    size_t ws_emit_tiles = 0, ws_sort_tiles = 0, ws_seg_tiles = 0, ws_mm_floats = 0;
    DevBuf ws_block, ws_pts_block, ws_sor_block;
    int64_t ws_sor_cap = 0;
    int ws_sor_frames = 0;

    // accumulating cloud (pose.cpp:434 cloud_big)
    o3dr_point* cloud_big = nullptr;
    int64_t cloud_cap = 0;
    o3dr_point* cloud_alt = nullptr;  // second buffer: partition target / receive buffer of the exchange
    int64_t cloud_alt_cap = 0;
    int64_t cloud_ub = 0;          // host-side upper bound of cc_big->count
    bool cloud_n_exact = true;     // cloud_ub IS the count (set by reads, resets and adopt; cleared by calls that append a bound)
    // running bounding box of cloud_big (min xyz, max xyz; device), kept by the frame calls so that the merge and
    // o3dr_cloud_big_bbox need no pass over the cloud; invalid after appends / transforms / exchanges
    float* cloud_box = nullptr;
    bool cloud_box_valid = false;
    // cloud_big's group-run heads for the merge's grid, recorded by the frame calls while it grows (4 points per byte);
    // valid under the same conditions as the running box, and for the leaf they were recorded with
    uint32_t* cloud_heads = nullptr;
    int64_t cloud_heads_cap = 0;  // points the flag buffer covers
    bool cloud_heads_valid = false;
    float cloud_heads_leaf[3] = {0.f, 0.f, 0.f};
    float cloud_heads_zo = 0.f;
    int cloud_box_enable = 1;  // O3DR_NO_CLOUD_BOX=1: always take the bounding box with a pass over the cloud
    CloudCounters* cc_big = nullptr;   // device
    CloudCounters* cc_tmp = nullptr;   // device, for single-shot calls
    CloudCounters* cc_host = nullptr;  // pinned
    CloudCounters* cc_host_dev = nullptr;  // ... as the device sees it: a one-workgroup call (kernels/small.inc) writes its result's
                                           // size straight into it, and the host reads it after the one synchronisation
    uint8_t* small_host = nullptr;     // pinned staging for the outputs of those calls (kSmallMax points)
    uint32_t* n_host = nullptr;        // pinned scratch (4 words)
    uint8_t* misc_dev = nullptr;       // 4 KiB device scratch: bbox (6 f32) | overflow (u32) | part counts (256 u64)
    uint8_t* misc_host = nullptr;      // pinned mirror
    uint8_t* misc_host_lut = nullptr;  // pinned staging of the Q table
    SortStats* stats_dev = nullptr;    // device statistics (bench.py byte accounting)
    SortStats* stats_host = nullptr;   // pinned

    DevBuf st_disp, st_bgr, st_in, st_out, st_kp, st_kpoff, st_poses;
    DevBuf st_xchg, st_merge, st_gather;  // o3dr_merge_partitioned: headers + count matrix, the merged slice, the gathered slices
    uint8_t* xchg_host = nullptr;         // pinned mirror of st_xchg
    size_t xchg_host_cap = 0;
    // host-input streaming of o3dr_accumulate_frames: two staging sets, uploads on their own stream
    DevBuf st2_disp[2], st2_bgr[2], st2_poses[2];
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    int test_corrupt = 0;  // o3dr_test_corrupt_next_gather: consumed by the next voxel grid
    int test_fail_at = 0;  // o3dr_test_fail_at: the numbered step of the next o3dr_merge_partitioned fails on this rank
    int64_t xchg_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // o3dr_merge_partitioned_stats
    int64_t place_ub = -1;   // o3dr_cloud_big_slice_counts_dev ran for a cloud of at most this many points and place_parts slices:
    int place_parts = 0;     // the (slice, tile) table in the workspace is what o3dr_cloud_big_place_slices moves by
    int test_hooks = 0;    // O3DR_TEST_HOOKS=1 at o3dr_ctx_create: the entry points of include/o3dr_testing.h act
    int host_batch = 32;  // frames per upload while the previous batch computes (O3DR_HOST_BATCH_FRAMES)
    Profiler prof;
};

static int dev_ensure(o3dr_ctx* c, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return O3DR_OK;
    if (b.p) {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        (void)hipGetLastError();
        want = bytes;
        if (hipMalloc(&b.p, want) != hipSuccess) {
            b.p = nullptr;
            return fail(O3DR_ERR_ALLOC, "hipMalloc failed (workspace)");
        }
    }
    b.cap = want;
    return O3DR_OK;
}
static void dev_release(DevBuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// (re)carve the per-batch workspace for `frames` clouds of at most `cap` points
// need_pts: ws.pts holds frames*cap points (A1 + A2 output); grp_tmp: ws.pts holds the result slots of the grouped
// whole-cloud voxel grid (one cloud, not in ws.pts itself)
static int ws_ensure(o3dr_ctx* c, int frames, int64_t cap, bool need_pts, bool grp_tmp = false)
{
    if (cap < 1) cap = 1;
    const size_t elems = (size_t)frames * (size_t)(cap + 1);
    const size_t emit_tiles = (size_t)frames * (size_t)((cap + kEmitTile - 1) / kEmitTile);
    const size_t sort_tiles = (size_t)frames * (size_t)((cap + kSortTile - 1) / kSortTile);
    const size_t seg_tiles = (size_t)frames * (size_t)((cap + kSegTile - 1) / kSegTile);
    // bounding-box slots per frame: one per reprojection tile + the keypoint slot, or kMinmaxBlocks
    size_t mm_slots = (size_t)((cap + kEmitTile - 1) / kEmitTile) + 1;
    if (mm_slots < (size_t)kMinmaxBlocks) mm_slots = kMinmaxBlocks;
    const size_t mm_floats = (size_t)frames * mm_slots * 6;
    if (elems > c->ws_elems || frames > c->ws_frames || emit_tiles > c->ws_emit_tiles ||
        sort_tiles > c->ws_sort_tiles || seg_tiles > c->ws_seg_tiles || mm_floats > c->ws_mm_floats) {
        const size_t MM = mm_floats > c->ws_mm_floats ? mm_floats : c->ws_mm_floats;
        const size_t E = elems > c->ws_elems ? elems : c->ws_elems;
        const int F = frames > c->ws_frames ? frames : c->ws_frames;
        const size_t TE = emit_tiles > c->ws_emit_tiles ? emit_tiles : c->ws_emit_tiles;
        const size_t TS = sort_tiles > c->ws_sort_tiles ? sort_tiles : c->ws_sort_tiles;
        const size_t TG = seg_tiles > c->ws_seg_tiles ? seg_tiles : c->ws_seg_tiles;
        size_t off = 0;
        size_t o_keys0 = off; off += align256(E * 4);
        size_t o_keys1 = off; off += align256(E * 4);
        size_t o_vals0 = off; off += align256(E * 4);
        size_t o_vals1 = off; off += align256(E * 4);
        size_t o_seg = off;   off += align256(E * 4);
        size_t o_keep = off;  off += align256(E * 4);
        size_t o_rs = off;    off += align256(E * 4);
        size_t o_gc = off;    off += align256(((E > (size_t)kGroupMinSlots ? E : (size_t)kGroupMinSlots) / kGroupCells + 2) * 4);
        size_t o_tile = off;  off += align256(TE * 4);
        size_t o_hist = off;  off += align256(TS * kMaxRadix * 4);
        size_t o_histp = off; off += align256(TS * kMaxRadix * 4);
        size_t o_segc = off;  off += align256(TG * 4);
        size_t o_hb = off;    off += align256(TG * 256);
        size_t o_part = off;  off += align256(((TS * kMaxRadix + TG + TE + E) / 4096 + 8 * (size_t)F + 16) * 4);
        size_t o_mm = off;    off += align256(MM * 4);
        size_t o_nv = off;    off += align256((size_t)F * 4);
        size_t o_nk = off;    off += align256((size_t)F * 4);
        size_t o_nx = off;    off += align256((size_t)F * 4);
        size_t o_no = off;    off += align256((size_t)F * 4);
        size_t o_oo = off;    off += align256((size_t)F * 8);
        size_t o_geom = off;  off += align256((size_t)F * sizeof(VoxelGeom));
        size_t o_geomr = off; off += align256((size_t)F * sizeof(VoxelGeom));
        size_t o_nr = off;    off += align256((size_t)F * 4);
        size_t o_ng = off;    off += align256((size_t)F * 4);
        size_t o_omm = off;   off += align256((E / 64 + 8 * (size_t)F + 64) * 6 * sizeof(float));
        size_t o_ommp = off;  off += align256((size_t)kBoxFoldBlocks * 6 * sizeof(float));
        size_t o_wgc = off;   off += align256((E / 64 + 8 * (size_t)F + 64) * 6 * sizeof(int32_t));
        CHK(dev_ensure(c, c->ws_block, off));
        char* base = (char*)c->ws_block.p;
        Workspace& w = c->ws;
        w.keys[0] = (uint32_t*)(base + o_keys0);
        w.keys[1] = (uint32_t*)(base + o_keys1);
        w.vals[0] = (uint32_t*)(base + o_vals0);
        w.vals[1] = (uint32_t*)(base + o_vals1);
        w.seg_start = (uint32_t*)(base + o_seg);
        w.keep_idx = (uint32_t*)(base + o_keep);
        w.run_start = (uint32_t*)(base + o_rs);
        w.grp_cnt = (uint32_t*)(base + o_gc);
        w.n_grp_out = (uint32_t*)(base + o_ng);
        w.geom_runs = (VoxelGeom*)(base + o_geomr);
        w.n_runs = (uint32_t*)(base + o_nr);
        w.out_mm = (float*)(base + o_omm);
        w.out_mm_partial = (float*)(base + o_ommp);
        w.wave_gc = (int32_t*)(base + o_wgc);
        w.tile_cnt = (uint32_t*)(base + o_tile);
        w.hist = (uint32_t*)(base + o_hist);
        w.hist_part = (uint32_t*)(base + o_histp);
        w.seg_cnt = (uint32_t*)(base + o_segc);
        w.head_bits = (uint8_t*)(base + o_hb);
        w.scan_partial = (uint32_t*)(base + o_part);
        w.mm = (float*)(base + o_mm);
        w.n_valid = (uint32_t*)(base + o_nv);
        w.n_kp = (uint32_t*)(base + o_nk);
        w.n_vox = (uint32_t*)(base + o_nx);
        w.n_out = (uint32_t*)(base + o_no);
        w.out_off = (uint64_t*)(base + o_oo);
        w.geom = (VoxelGeom*)(base + o_geom);
        w.bytes = off;
        c->ws_elems = E;
        c->ws_frames = F;
        c->ws_emit_tiles = TE;
        c->ws_sort_tiles = TS;
        c->ws_seg_tiles = TG;
        c->ws_mm_floats = MM;
    }
    c->ws.mm_stride = (int64_t)mm_slots;
    c->ws.grp_slots = 0;
    if (need_pts || grp_tmp) {
        size_t pe = (size_t)frames * (size_t)cap;
        if (grp_tmp && pe < (size_t)kGroupMinSlots) pe = (size_t)kGroupMinSlots;
        if (grp_tmp) c->ws.grp_slots = (int64_t)pe;
        if (pe > c->ws_pts_elems) {
            CHK(dev_ensure(c, c->ws_pts_block, pe * sizeof(o3dr_point)));
            c->ws_pts_elems = pe;
        }
        c->ws.pts = (o3dr_point*)c->ws_pts_block.p;
    }
    c->ws.frames = frames;
    c->ws.cap = cap;
    return O3DR_OK;
}

// buffers of the statistical outlier removal for `frames` clouds of at most `cap` points (allocated on first use: the
// measured configs run without it)
static int sor_ensure(o3dr_ctx* c, int frames, int64_t cap)
{
    if (cap < 1) cap = 1;
    if (frames < 1) frames = 1;
    if (cap > c->ws_sor_cap || frames > c->ws_sor_frames) {
        const int64_t C = cap > c->ws_sor_cap ? cap : c->ws_sor_cap;
        const size_t F = (size_t)(frames > c->ws_sor_frames ? frames : c->ws_sor_frames);
        uint32_t max_cells = (uint32_t)(C / 2 > 1024 ? C / 2 : 1024);
        if (max_cells > (1u << 22)) max_cells = 1u << 22;
        size_t off = 0;
        size_t o_xyz = off;  off += align256(F * (size_t)C * 16);
        size_t o_pts = off;  off += align256(F * (size_t)C * 16);
        size_t o_dist = off; off += align256(F * (size_t)C * 4);
        size_t o_cf = off;   off += align256(F * ((size_t)max_cells + 1) * 4);
        size_t o_cz = off;   off += align256(F * ((size_t)max_cells + 1) * 8);
        size_t o_part = off; off += align256(F * 256 * 2 * 8);
        size_t o_geom = off; off += align256(F * sizeof(SorGeom));
        size_t o_n = off;    off += align256(F * 4);
        size_t o_left = off; off += align256(F * (size_t)C * 4);
        size_t o_lc = off;   off += align256(F * 4);
        CHK(dev_ensure(c, c->ws_sor_block, off));
        char* base = (char*)c->ws_sor_block.p;
        Workspace& w = c->ws;
        w.sor_xyz = (float4*)(base + o_xyz);
        w.sor_pts = (o3dr_point*)(base + o_pts);
        w.sor_dist = (float*)(base + o_dist);
        w.sor_cell_first = (uint32_t*)(base + o_cf);
        w.sor_cell_z = (float2*)(base + o_cz);
        w.sor_partial = (double*)(base + o_part);
        w.sor_geom = (SorGeom*)(base + o_geom);
        w.sor_n = (uint32_t*)(base + o_n);
        w.sor_left = (uint32_t*)(base + o_left);
        w.sor_left_cnt = (uint32_t*)(base + o_lc);
        w.sor_max_cells = max_cells;
        w.sor_cap = C;
        c->ws_sor_cap = C;
        c->ws_sor_frames = (int)F;
    }
    return O3DR_OK;
}
static inline bool sor_on(const o3dr_ctx* c) { return c->params.sor_enable && c->params.jump_pixels > 0; }  // :1673

extern "C" int o3dr_version(void) { return O3DR_VERSION; }
extern "C" const char* o3dr_last_error(void) { return g_err.c_str(); }

extern "C" void o3dr_default_params(o3dr_params* p)
{
    if (!p) return;
    p->min_disparity = 64;          // pose.h:93
    p->voxel_size = 0.1;            // pose.h:118
    p->bounding_box = 20;           // pose.h:94
    p->cutout_ratio = 8;            // pose.h:126
    p->jump_pixels = 10;            // pose.h:96
    p->min_points_per_voxel = 1;    // pose.h:108
    p->dont_downsample = 0;
    p->sor_enable = 1;              // pose_functions.cpp:1673-1686: always on in the reference's per-frame path
    p->blur_kernel = 1;             // pose.h:98
    p->disparity_f64 = 0;           // use_segment_labels off
}

static int cloud_box_clear(o3dr_ctx* c);

extern "C" int o3dr_ctx_create(int device_id, o3dr_ctx** out_ctx)
{
    if (!out_ctx) return fail(O3DR_ERR_INVALID_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(O3DR_ERR_NO_DEVICE, "no HIP device: libo3dr has no CPU path");
    }
    if (device_id < 0 || device_id >= n) return fail(O3DR_ERR_INVALID_ARG, "device_id out of range");
    if (hipSetDevice(device_id) != hipSuccess) return fail(O3DR_ERR_NO_DEVICE, "hipSetDevice failed");
    o3dr_ctx* c = new o3dr_ctx();
    c->device = device_id;
    o3dr_default_params(&c->params);
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(O3DR_ERR_NO_DEVICE, "hipStreamCreate failed");
    }
    c->stream = c->own_stream;
    if (hipMalloc((void**)&c->cloud_box, 6 * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&c->cc_big, sizeof(CloudCounters)) != hipSuccess ||
        hipMalloc((void**)&c->cc_tmp, sizeof(CloudCounters)) != hipSuccess ||
        hipHostMalloc((void**)&c->cc_host, 2 * sizeof(CloudCounters), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->n_host, 4 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->small_host, (size_t)kSmallMax * sizeof(o3dr_point), hipHostMallocDefault) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->cc_host_dev, c->cc_host, 0) != hipSuccess ||
        hipMalloc((void**)&c->stats_dev, sizeof(SortStats)) != hipSuccess ||
        hipMalloc((void**)&c->misc_dev, 4096) != hipSuccess ||
        hipHostMalloc((void**)&c->misc_host, 4096, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->misc_host_lut, 256 * sizeof(QLutEntry), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&c->q_lut, 256 * sizeof(QLutEntry)) != hipSuccess ||
        hipHostMalloc((void**)&c->stats_host, sizeof(SortStats), hipHostMallocDefault) != hipSuccess) {
        delete c;
        return fail(O3DR_ERR_ALLOC, "counter allocation failed");
    }
    (void)hipMemsetAsync(c->cc_big, 0, sizeof(CloudCounters), c->stream);
    (void)hipMemsetAsync(c->cc_tmp, 0, sizeof(CloudCounters), c->stream);
    (void)hipMemsetAsync(c->stats_dev, 0, sizeof(SortStats), c->stream);
    (void)hipMemsetAsync(c->misc_dev, 0, 4096, c->stream);
    // the only environment switches, all read here, once per context (tests drive them)
    const char* hb_env = getenv("O3DR_HOST_BATCH_FRAMES");
    if (hb_env && atoi(hb_env) > 0) c->host_batch = atoi(hb_env);
    const char* ru_env = getenv("O3DR_RUNS");
    if (ru_env && atoi(ru_env) == 0) c->use_runs = 0;       // whole-cloud grids always sort points
    else if (ru_env && atoi(ru_env) == 2) c->use_runs = 2;  // ... always sort runs (default: decided per cloud on the device)
    if (getenv("O3DR_NO_CLOUD_BOX")) c->cloud_box_enable = 0;
    const char* eb_env = getenv("O3DR_EXACT_BOX");
    c->exact_box = eb_env && atoi(eb_env) == 1;
    const char* sm_env = getenv("O3DR_SMALL");
    if (sm_env) c->small_path = atoi(sm_env) != 0;
    const char* th_env = getenv("O3DR_TEST_HOOKS");
    c->test_hooks = th_env && atoi(th_env) == 1;
    const char* sl_env = getenv("O3DR_SLABS");
    if (sl_env && sl_env[0] == '0') c->slab_shift_env = -1;
    else if (sl_env && sl_env[0] == 's' && atoi(sl_env + 1) >= 0 && atoi(sl_env + 1) <= 20) c->slab_shift_env = atoi(sl_env + 1);
    const char* env = getenv("O3DR_BATCH_FRAMES");
    if (env && atoi(env) > 0) c->max_batch = atoi(env) > 512 ? 512 : atoi(env);
    (void)cloud_box_clear(c);
    *out_ctx = c;
    return O3DR_OK;
}

extern "C" int o3dr_ctx_destroy(o3dr_ctx* c)
{
    if (!c) return O3DR_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    dev_release(c->ws_block);
    dev_release(c->ws_pts_block);
    dev_release(c->ws_sor_block);
    dev_release(c->st_disp);
    dev_release(c->st_bgr);
    dev_release(c->st_in);
    dev_release(c->st_out);
    dev_release(c->st_kp);
    dev_release(c->st_kpoff);
    dev_release(c->st_poses);
    dev_release(c->st_xchg);
    dev_release(c->st_merge);
    dev_release(c->st_gather);
    if (c->xchg_host) (void)hipHostFree(c->xchg_host);
    dev_release(c->bil_tab);
    dev_release(c->st_blur);
    dev_release(c->st_blur_in);
    dev_release(c->st_hist);
    if (c->cloud_big) (void)hipFree(c->cloud_big);
    if (c->cloud_alt) (void)hipFree(c->cloud_alt);
    if (c->cloud_box) (void)hipFree(c->cloud_box);
    if (c->cloud_heads) (void)hipFree(c->cloud_heads);
    if (c->cc_big) (void)hipFree(c->cc_big);
    if (c->cc_tmp) (void)hipFree(c->cc_tmp);
    if (c->cc_host) (void)hipHostFree(c->cc_host);
    if (c->n_host) (void)hipHostFree(c->n_host);
    if (c->small_host) (void)hipHostFree(c->small_host);
    if (c->stats_dev) (void)hipFree(c->stats_dev);
    if (c->misc_dev) (void)hipFree(c->misc_dev);
    if (c->misc_host) (void)hipHostFree(c->misc_host);
    if (c->misc_host_lut) (void)hipHostFree(c->misc_host_lut);
    if (c->q_lut) (void)hipFree(c->q_lut);
    if (c->stats_host) (void)hipHostFree(c->stats_host);
    for (int i = 0; i < 2; ++i) {
        dev_release(c->st2_disp[i]);
        dev_release(c->st2_bgr[i]);
        dev_release(c->st2_poses[i]);
        if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
        if (c->ev_done[i]) (void)hipEventDestroy(c->ev_done[i]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return O3DR_OK;
}

#define CTX_ENTER(c)                                                        \
    do {                                                                    \
        if (!(c)) return fail(O3DR_ERR_INVALID_ARG, "ctx is NULL");         \
        HIPCHK(hipSetDevice((c)->device));                                  \
    } while (0)

extern "C" int o3dr_ctx_set_stream(o3dr_ctx* c, void* hip_stream)
{
    CTX_ENTER(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return O3DR_OK;
}
extern "C" int o3dr_ctx_synchronize(o3dr_ctx* c)
{
    CTX_ENTER(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    return O3DR_OK;
}
extern "C" int o3dr_set_camera(o3dr_ctx* c, const double Q[16])
{
    CTX_ENTER(c);
    if (!Q) return fail(O3DR_ERR_INVALID_ARG, "Q is NULL");
    memcpy(c->Q, Q, sizeof c->Q);
    c->has_Q = true;
    // Rectified stereo Q (cv::stereoRectify): X and Y rows use one pixel coordinate each, Z and W only the
    // disparity.  Then 1./W and Z are functions of the 8-bit disparity alone: tabulate them with the very
    // operations the per-pixel path would execute (IEEE fp64 on both sides), drop the exact-zero terms.
    static const int zeros[] = {1, 2, 4, 6, 8, 9, 12, 13};
    bool sparse = true;
    for (int z : zeros) sparse = sparse && (Q[z] == 0.0);
    c->q_lut_on = false;
    if (sparse) {
        QLutEntry* h = (QLutEntry*)c->misc_host_lut;
        for (int d = 0; d < 256; ++d) {
            const double dd = (double)d;
            const double t2 = ((Q[8] * 0.0 + Q[9] * 0.0) + Q[10] * dd) + Q[11];
            const double t3 = ((Q[12] * 0.0 + Q[13] * 0.0) + Q[14] * dd) + Q[15];
            const double alpha = 1. / t3;
            h[d].alpha = alpha;
            h[d].z = (float)(t2 * alpha + 0.0);
            h[d].pad = 0.f;
        }
        HIPCHK(hipMemcpyAsync(c->q_lut, h, 256 * sizeof(QLutEntry), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->q_lut_on = true;
    }
    return O3DR_OK;
}
extern "C" int o3dr_set_params(o3dr_ctx* c, const o3dr_params* p)
{
    CTX_ENTER(c);
    if (!p) return fail(O3DR_ERR_INVALID_ARG, "params is NULL");
    if (p->bounding_box < 0 || p->cutout_ratio <= 0 || p->jump_pixels < 0 || !(p->voxel_size > 0))
        return fail(O3DR_ERR_INVALID_ARG, "params out of range");
    c->params = *p;
    return O3DR_OK;
}
extern "C" int o3dr_get_params(o3dr_ctx* c, o3dr_params* p)
{
    CTX_ENTER(c);
    if (!p) return fail(O3DR_ERR_INVALID_ARG, "params is NULL");
    *p = c->params;
    return O3DR_OK;
}

// -------------------------------------------------------------------------------------------------
// helpers
// -------------------------------------------------------------------------------------------------
struct GridShape {
    int cs, Ny, Nx;
    int64_t n;
};
// pose_functions.cpp:638 and the loop bounds of :1094-1096
static GridShape grid_shape(const o3dr_params& p, int rows, int cols)
{
    GridShape g;
    g.cs = (int)(cols / p.cutout_ratio);
    g.Ny = g.Nx = 0;
    if (p.jump_pixels > 0) {
        const int h = rows - 2 * p.bounding_box, w = cols - p.bounding_box - g.cs;
        g.Ny = h > 0 ? (h + p.jump_pixels - 1) / p.jump_pixels : 0;
        g.Nx = w > 0 ? (w + p.jump_pixels - 1) / p.jump_pixels : 0;
    }
    g.n = (int64_t)g.Ny * g.Nx;
    return g;
}

extern "C" int64_t o3dr_max_points(o3dr_ctx* c, int32_t rows, int32_t cols)
{
    if (!c) return 0;
    return grid_shape(c->params, rows, cols).n;
}

static int check_images(const o3dr_ctx* c, const uint8_t* disp, int64_t disp_pitch, const uint8_t* bgr, int64_t bgr_pitch,
                        int rows, int cols, int64_t disp_frame_stride = 0)
{
    if (!disp || !bgr) return fail(O3DR_ERR_INVALID_ARG, "image pointer is NULL");
    if (rows <= 0 || cols <= 0) return fail(O3DR_ERR_INVALID_ARG, "rows/cols must be positive");
    const int64_t esz = c->params.disparity_f64 ? 8 : 1;
    if (disp_pitch < esz * cols || bgr_pitch < 3 * (int64_t)cols) return fail(O3DR_ERR_INVALID_ARG, "pitch smaller than a row");
    if (esz == 8 && (((uintptr_t)disp | (uintptr_t)disp_pitch | (uintptr_t)disp_frame_stride) & 7))
        return fail(O3DR_ERR_INVALID_ARG, "CV_64F disparities must be 8-byte aligned (pointer, pitch, frame stride)");
    return O3DR_OK;
}

static void fill_args(o3dr_ctx* c, ReprojectArgs& a, const uint8_t* disp, int64_t disp_pitch, int64_t disp_fstride,
                      const uint8_t* bgr, int64_t bgr_pitch, int64_t bgr_fstride, int rows, int cols,
                      const GridShape& g, int64_t out_fstride)
{
    memset(&a, 0, sizeof a);
    a.disp = disp;
    a.bgr = bgr;
    a.disp_pitch = disp_pitch;
    a.bgr_pitch = bgr_pitch;
    a.disp_fstride = disp_fstride;
    a.bgr_fstride = bgr_fstride;
    a.rows = rows;
    a.cols = cols;
    a.bb = c->params.bounding_box;
    a.cs = g.cs;
    a.jump = c->params.jump_pixels;
    a.Ny = g.Ny;
    a.Nx = g.Nx;
    a.n_tiles = (int)((g.n + kEmitTile - 1) / kEmitTile);
    a.disp_f64 = c->params.disparity_f64 ? 1 : 0;
    a.vec4 = (!a.disp_f64 && a.jump == 1 && (g.Nx % 4) == 0 && (g.cs % 4) == 0 && (disp_pitch % 4) == 0 && (bgr_pitch % 4) == 0 &&
              (disp_fstride % 4) == 0 && (bgr_fstride % 4) == 0 && ((uintptr_t)disp % 4) == 0 &&
              ((uintptr_t)bgr % 4) == 0)
                 ? 1
                 : 0;
    memcpy(a.Q, c->Q, sizeof a.Q);
    a.min_disp = c->params.min_disparity;
    {   // the same comparison on integers (no fp64 convert + compare per pixel): d > m <=> d > floor(m) for integer d
        const double m = a.min_disp;
        a.min_disp_u8 = !(m == m) || m >= 255.0 ? 255 : (m < 0.0 ? -1 : (int32_t)floor(m));
    }
    a.out_fstride = out_fstride;
    a.mm_stride = c->ws.mm_stride;
    a.lut = (c->q_lut_on && !a.disp_f64) ? c->q_lut : nullptr;
}

// Thickness (log2, in cells) of the grid slabs the fused batch path sorts a frame's points into (slab_class in
// kernels/reproject.inc): about two thirds of the distance between the depth sheets of two neighbouring disparity levels
// at a nominal disparity, so that the sheets one line of pixels lands on fall into different classes.  A layout choice
// only: results do not depend on it.  -1: plain pixel order (O3DR_SLABS=0).
static int slab_shift_for(const o3dr_ctx* c, int rows, int cols, const float leaf[3])
{
    if (c->slab_shift_env >= -1) return c->slab_shift_env;
    const double* Q = c->Q;
    auto point = [&](double d, double out[3]) {
        const double v[4] = {0.5 * cols, 0.5 * rows, d, 1.0};
        double t[4];
        for (int r = 0; r < 4; ++r) t[r] = ((Q[4 * r] * v[0] + Q[4 * r + 1] * v[1]) + Q[4 * r + 2] * v[2]) + Q[4 * r + 3];
        for (int r = 0; r < 3; ++r) out[r] = t[r] / t[3];
    };
    double p0[3], p1[3];
    point(128.0, p0);
    point(129.0, p1);
    const double dist = sqrt((p1[0] - p0[0]) * (p1[0] - p0[0]) + (p1[1] - p0[1]) * (p1[1] - p0[1]) + (p1[2] - p0[2]) * (p1[2] - p0[2]));
    const double lf = leaf[0] > leaf[1] ? (leaf[0] > leaf[2] ? leaf[0] : leaf[2]) : (leaf[1] > leaf[2] ? leaf[1] : leaf[2]);
    const double cells = dist / lf / 1.5;
    if (!(cells >= 2.0)) return 0;  // (also NaN / inf from a degenerate Q)
    int sh = 0;
    while (sh < 16 && (double)(2 << sh) <= cells) ++sh;
    return sh;
}

// stage a host buffer into HBM (or pass a device pointer through)
static int stage_in(o3dr_ctx* c, DevBuf& b, const void* src, size_t bytes, int mem, const void** dev)
{
    if (mem == O3DR_MEM_DEVICE) {
        *dev = src;
        return O3DR_OK;
    }
    CHK(dev_ensure(c, b, bytes ? bytes : 1));
    if (bytes) HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    *dev = b.p;
    return O3DR_OK;
}

// read a CloudCounters back (synchronises)
static int read_counters(o3dr_ctx* c, const CloudCounters* dev, CloudCounters* host, const CloudCounters* dev2 = nullptr,
                         CloudCounters* host2 = nullptr)
{
    HIPCHK(hipMemcpyAsync(c->cc_host, dev, sizeof(CloudCounters), hipMemcpyDeviceToHost, c->stream));
    if (dev2) HIPCHK(hipMemcpyAsync(c->cc_host + 1, dev2, sizeof(CloudCounters), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *host = *c->cc_host;
    if (dev2) {
        *host2 = c->cc_host[1];
        if (dev2 == c->cc_big) {
            c->cloud_ub = (int64_t)host2->count;
            c->cloud_n_exact = true;
        }
    }
    if (dev == c->cc_big) {  // every read of the cloud's counters makes the host-side bound exact
        c->cloud_ub = (int64_t)host->count;
        c->cloud_n_exact = true;
    }
    if ((host->status | (dev2 ? host2->status : 0u)) & O3DR_STATUS_INTERNAL)
        return fail(O3DR_ERR_INTERNAL, "a device-side gather guard tripped (record or point id outside its cloud); results are invalid");
    return O3DR_OK;
}

static int zero_counters(o3dr_ctx* c, CloudCounters* dev)
{
    HIPCHK(hipMemsetAsync(dev, 0, sizeof(CloudCounters), c->stream));
    return O3DR_OK;
}

// weight tables of cv::bilateralFilter (OpenCV 3.1 smooth.cpp bilateralFilter_8u), built with the host's exp()
static int bilateral_prepare(o3dr_ctx* c, int d, double sigma_color, double sigma_space)
{
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    if (c->bil_valid && c->bil_d == d && c->bil_sc == sigma_color && c->bil_ss == sigma_space) return O3DR_OK;
    const double gauss_color_coeff = -0.5 / (sigma_color * sigma_color);
    const double gauss_space_coeff = -0.5 / (sigma_space * sigma_space);
    int radius = d <= 0 ? (int)lrint(sigma_space * 1.5) : d / 2;  // cvRound
    if (radius < 1) radius = 1;
    if (radius > kBilMaxRadius) return fail(O3DR_ERR_INVALID_ARG, "bilateral filter radius above 64 is not supported");
    const int dd = 2 * radius + 1, tw = bilateral_tile_width(radius);
    std::vector<float> tab(256 + 2 * (size_t)dd * dd);
    for (int i = 0; i < 256; ++i) tab[i] = (float)std::exp(i * i * gauss_color_coeff);
    int maxk = 0;
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            const double r = std::sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            ++maxk;
        }
    int k = 0;
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            const double r = std::sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            tab[256 + k] = (float)std::exp(r * r * gauss_space_coeff);
            const int32_t ofs = i * tw + j;
            memcpy(&tab[256 + maxk + k], &ofs, sizeof ofs);
            ++k;
        }
    HIPCHK(hipStreamSynchronize(c->stream));  // earlier launches may still read the old table
    c->bil_valid = false;
    CHK(dev_ensure(c, c->bil_tab, (256 + 2 * (size_t)maxk) * sizeof(float)));
    HIPCHK(hipMemcpy(c->bil_tab.p, tab.data(), (256 + 2 * (size_t)maxk) * sizeof(float), hipMemcpyHostToDevice));
    c->bil_d = d;
    c->bil_sc = sigma_color;
    c->bil_ss = sigma_space;
    c->bil_radius = radius;
    c->bil_maxk = maxk;
    c->bil_valid = true;
    return O3DR_OK;
}

// blur_kernel > 1 (pose_functions.cpp:1040-1047): the frames' disparity images are filtered into a scratch
// buffer the reprojection then reads.  Rewrites (disp, pitch, frame stride) in place.
static int maybe_blur(o3dr_ctx* c, const uint8_t** disp_d, int64_t* pitch, int64_t* fstride, int rows, int cols, int frames)
{
    const int bk = c->params.blur_kernel;
    if (bk <= 1) return O3DR_OK;
    if (c->params.disparity_f64)
        return fail(O3DR_ERR_INVALID_ARG, "blur_kernel > 1 needs CV_8UC1 disparities (cv::bilateralFilter rejects CV_64F)");
    CHK(bilateral_prepare(c, bk, (double)(bk * 2), (double)(bk / 2)));
    const int64_t out_pitch = cols, out_fstride = (int64_t)rows * cols;
    CHK(dev_ensure(c, c->st_blur, (size_t)out_fstride * (size_t)frames + 16));
    launch_bilateral(&c->prof, c->stream, *disp_d, *pitch, *fstride, rows, cols, frames, c->bil_radius, c->bil_maxk,
                     (const float*)c->bil_tab.p, (uint8_t*)c->st_blur.p, out_pitch, out_fstride);
    HIPCHK(hipGetLastError());
    *disp_d = (const uint8_t*)c->st_blur.p;
    *pitch = out_pitch;
    *fstride = out_fstride;
    return O3DR_OK;
}

// A1 (+A2) of one frame into `dst` (device).  n_valid ends up in ws.n_valid[0].
static int run_reproject_single(o3dr_ctx* c, const uint8_t* disp_d, int64_t disp_pitch, const uint8_t* bgr_d,
                                int64_t bgr_pitch, int rows, int cols, const GridShape& g, const float* T,
                                const float* kp_d, int n_kp, o3dr_point* dst, const float* T_dev = nullptr)
{
    int64_t disp_fstride = 0;
    CHK(maybe_blur(c, &disp_d, &disp_pitch, &disp_fstride, rows, cols, 1));
    ReprojectArgs a;
    fill_args(c, a, disp_d, disp_pitch, 0, bgr_d, bgr_pitch, 0, rows, cols, g, 0);
    if (T_dev) {  // pose already in HBM (one frame of a batched call)
        a.xf_mode = 2;
        a.poses = T_dev;
    } else if (T) {
        a.xf_mode = 1;
        for (int i = 0; i < 12; ++i) a.T[i] = T[i];
    }
    launch_minmax_init(&c->prof, c->stream, c->ws.mm, c->ws.mm_stride, a.n_tiles, c->ws.n_kp, 1);
    if (c->params.jump_pixels != 1 && n_kp > 0)
        launch_keypoint_pass(&c->prof, c->stream, a, kp_d, n_kp, dst, c->ws.n_kp, c->ws.mm);
    launch_reproject(&c->prof, c->stream, a, 1, dst, c->ws.tile_cnt, c->ws.n_kp, c->ws.n_valid, c->ws.mm,
                     c->ws.scan_partial);
    HIPCHK(hipGetLastError());
    return O3DR_OK;
}

static int read_u32(o3dr_ctx* c, const uint32_t* dev, uint32_t* host)
{
    HIPCHK(hipMemcpyAsync(c->n_host, dev, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *host = c->n_host[0];
    return O3DR_OK;
}

// common body of A1, A1+A2 and A6
static int frame_call(o3dr_ctx* c, const uint8_t* disp, int64_t disp_pitch, const uint8_t* bgr, int64_t bgr_pitch,
                      int rows, int cols, const float* T, const float* kp_xy, int n_kp, bool downsample,
                      o3dr_point* out, int64_t out_capacity, int64_t* n_out, uint32_t* status, int mem)
{
    if (n_out) *n_out = 0;
    if (status) *status = 0;
    CTX_ENTER(c);
    if (!c->has_Q) return fail(O3DR_ERR_NOT_CONFIGURED, "o3dr_set_camera has not been called");
    if (!n_out || !out) return fail(O3DR_ERR_INVALID_ARG, "out / n_out is NULL");
    if (mem != O3DR_MEM_HOST && mem != O3DR_MEM_DEVICE) return fail(O3DR_ERR_INVALID_ARG, "bad mem kind");
    CHK(check_images(c, disp, disp_pitch, bgr, bgr_pitch, rows, cols));
    if (n_kp < 0 || (n_kp > 0 && !kp_xy)) return fail(O3DR_ERR_INVALID_ARG, "bad keypoint list");
    if (c->params.jump_pixels == 1) n_kp = 0;  // :1057 keypoints are skipped when every pixel is taken
    const GridShape g = grid_shape(c->params, rows, cols);
    const int64_t cap = g.n + n_kp;
    if (mem == O3DR_MEM_DEVICE && out_capacity < cap)
        return fail(O3DR_ERR_CAPACITY, "device output must hold o3dr_max_points()+n_kp points");
    if (cap == 0) return O3DR_OK;

    const void *disp_d, *bgr_d, *kp_d = nullptr;
    CHK(stage_in(c, c->st_disp, disp, (size_t)rows * disp_pitch, mem, &disp_d));
    CHK(stage_in(c, c->st_bgr, bgr, (size_t)rows * bgr_pitch, mem, &bgr_d));
    if (n_kp > 0) CHK(stage_in(c, c->st_kp, kp_xy, (size_t)n_kp * 2 * sizeof(float), mem, &kp_d));
    CHK(ws_ensure(c, 1, cap, downsample));

    o3dr_point* final_dst = out;
    if (mem == O3DR_MEM_HOST) {
        CHK(dev_ensure(c, c->st_out, (size_t)cap * sizeof(o3dr_point)));
        final_dst = (o3dr_point*)c->st_out.p;
    }
    int64_t n_final = 0;
    uint32_t st = 0;
    // Frames of at most kSmallMax candidates - the reference's own configuration, --jump_pixels 10 .. 15 - take the whole
    // path in ONE launch of one workgroup (kernels/small.inc) instead of ~45 launches that cost their latency and nothing
    // else; with the outlier removal on, its kernels run between the two halves.  O3DR_SMALL=0 switches it off.
    const bool small = c->small_path && cap <= kSmallMax && !c->params.disparity_f64 && !c->test_corrupt;
    if (small) {
        const uint8_t* dsp = (const uint8_t*)disp_d;
        int64_t dsp_pitch = disp_pitch, dsp_fstride = 0;
        CHK(maybe_blur(c, &dsp, &dsp_pitch, &dsp_fstride, rows, cols, 1));
        ReprojectArgs a;
        fill_args(c, a, dsp, dsp_pitch, 0, (const uint8_t*)bgr_d, bgr_pitch, 0, rows, cols, g, 0);
        if (T) {
            a.xf_mode = 1;
            for (int i = 0; i < 12; ++i) a.T[i] = T[i];
        }
        float leaf[3];
        leaf[0] = leaf[1] = leaf[2] = (float)(c->params.voxel_size / 5);  // pose_functions.cpp:1698
        const bool with_sor = downsample && sor_on(c);
        // the result's size lands in pinned host memory, written by the kernel itself; host outputs come back in the same
        // wait (all `cap` slots, at most 128 KiB, through a pinned staging buffer): ONE synchronisation per call
        if (with_sor) {
            CHK(sor_ensure(c, 1, cap));
            launch_small_frame(&c->prof, c->stream, a, (const float*)kp_d, n_kp, 0, leaf, nullptr, c->ws.pts, c->cc_tmp, c->ws.n_valid, c->ws.mm);
            launch_sor_small(&c->prof, c->stream, c->ws, c->ws.pts, c->ws.n_valid, cap, 1.0, c->ws.sor_pts, c->ws.sor_n);
            launch_small_voxel(&c->prof, c->stream, c->ws.sor_pts, c->ws.sor_n, 0, nullptr, leaf, 0, 0.f, final_dst, c->cc_host_dev);
        } else {
            launch_small_frame(&c->prof, c->stream, a, (const float*)kp_d, n_kp, downsample ? 1 : 0, leaf, c->ws.pts, final_dst, c->cc_host_dev,
                               nullptr, nullptr);
        }
        HIPCHK(hipGetLastError());
        if (mem == O3DR_MEM_HOST)
            HIPCHK(hipMemcpyAsync(c->small_host, final_dst, (size_t)cap * sizeof(o3dr_point), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const CloudCounters cc = *c->cc_host;
        if (cc.status & O3DR_STATUS_INTERNAL)
            return fail(O3DR_ERR_INTERNAL, "a device-side gather guard tripped (record or point id outside its cloud); results are invalid");
        if ((int64_t)cc.count > cap) return fail(O3DR_ERR_INTERNAL, "the one-workgroup path returned more points than it was given");
        if (mem == O3DR_MEM_HOST) {
            if ((int64_t)cc.count > out_capacity) return fail(O3DR_ERR_CAPACITY, "output buffer too small");
            memcpy(out, c->small_host, (size_t)cc.count * sizeof(o3dr_point));
        }
        *n_out = (int64_t)cc.count;
        if (status) *status = cc.status;
        return O3DR_OK;
    } else if (!downsample) {
        CHK(run_reproject_single(c, (const uint8_t*)disp_d, disp_pitch, (const uint8_t*)bgr_d, bgr_pitch, rows, cols, g,
                                 T, (const float*)kp_d, n_kp, final_dst));
        uint32_t nv = 0;
        CHK(read_u32(c, c->ws.n_valid, &nv));
        n_final = nv;
    } else {
        CHK(run_reproject_single(c, (const uint8_t*)disp_d, disp_pitch, (const uint8_t*)bgr_d, bgr_pitch, rows, cols, g,
                                 T, (const float*)kp_d, n_kp, c->ws.pts));
        CHK(zero_counters(c, c->cc_tmp));
        VoxelArgs v;
        v.in = c->ws.pts;
        v.in_fstride = 0;
        v.n_dev = c->ws.n_valid;
        v.frames = 1;
        v.cap = cap;
        v.leaf[0] = v.leaf[1] = v.leaf[2] = (float)(c->params.voxel_size / 5);  // pose_functions.cpp:1698
        v.min_points = 0;
        v.z_offset = 0.f;
        v.out_base = final_dst;
        v.cc = c->cc_tmp;
        v.passthrough = 0;
        v.mm_used = (int)((g.n + kEmitTile - 1) / kEmitTile) + 1;
        v.stats = c->stats_dev;
        v.use_runs = 0;
        if (sor_on(c)) {  // pose_functions.cpp:1673-1686 in front of the per-frame voxel grid
            CHK(sor_ensure(c, 1, cap));
            v.mm_used = launch_sor(&c->prof, c->stream, c->ws, c->ws.pts, 0, c->ws.n_valid, 1, cap, v.mm_used, 1.0, c->ws.sor_pts, 0,
                                   c->ws.sor_n);
            v.in = c->ws.sor_pts;
            v.n_dev = c->ws.sor_n;
        }
        launch_voxel_grid(&c->prof, c->stream, c->ws, v);
        HIPCHK(hipGetLastError());
        CloudCounters cc;
        CHK(read_counters(c, c->cc_tmp, &cc));
        n_final = (int64_t)cc.count;
        st = cc.status;
    }
    if (mem == O3DR_MEM_HOST) {
        if (n_final > out_capacity) return fail(O3DR_ERR_CAPACITY, "output buffer too small");
        if (n_final > 0) {
            HIPCHK(hipMemcpyAsync(out, final_dst, (size_t)n_final * sizeof(o3dr_point), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
    }
    *n_out = n_final;
    if (status) *status = st;
    return O3DR_OK;
}

extern "C" int o3dr_create_single_img_pt_cloud(o3dr_ctx* c, const uint8_t* disp, int64_t disp_pitch, const uint8_t* bgr,
                                               int64_t bgr_pitch, int32_t rows, int32_t cols, const float* kp_xy,
                                               int32_t n_kp, o3dr_point* out, int64_t out_capacity, int64_t* n_out,
                                               int32_t mem)
{
    return frame_call(c, disp, disp_pitch, bgr, bgr_pitch, rows, cols, nullptr, kp_xy, n_kp, false, out, out_capacity,
                      n_out, nullptr, mem);
}

extern "C" int o3dr_reproject_transform(o3dr_ctx* c, const uint8_t* disp, int64_t disp_pitch, const uint8_t* bgr,
                                        int64_t bgr_pitch, int32_t rows, int32_t cols, const float T[16],
                                        const float* kp_xy, int32_t n_kp, o3dr_point* out, int64_t out_capacity,
                                        int64_t* n_out, int32_t mem)
{
    if (!T) {
        if (n_out) *n_out = 0;
        return fail(O3DR_ERR_INVALID_ARG, "T is NULL");
    }
    return frame_call(c, disp, disp_pitch, bgr, bgr_pitch, rows, cols, T, kp_xy, n_kp, false, out, out_capacity, n_out,
                      nullptr, mem);
}

extern "C" int o3dr_create_and_transform_pt_cloud(o3dr_ctx* c, const uint8_t* disp, int64_t disp_pitch,
                                                  const uint8_t* bgr, int64_t bgr_pitch, int32_t rows, int32_t cols,
                                                  const float T[16], const float* kp_xy, int32_t n_kp, o3dr_point* out,
                                                  int64_t out_capacity, int64_t* n_out, uint32_t* status, int32_t mem)
{
    if (!T) {
        if (n_out) *n_out = 0;
        return fail(O3DR_ERR_INVALID_ARG, "T is NULL");
    }
    const bool ds = c ? !c->params.dont_downsample : true;
    return frame_call(c, disp, disp_pitch, bgr, bgr_pitch, rows, cols, T, kp_xy, n_kp, ds, out, out_capacity, n_out,
                      status, mem);
}

extern "C" int o3dr_transform_pt_cloud(o3dr_ctx* c, const o3dr_point* in, int64_t n, const float T[16], o3dr_point* out,
                                       int32_t mem)
{
    CTX_ENTER(c);
    if (n < 0 || !T || (n > 0 && (!in || !out))) return fail(O3DR_ERR_INVALID_ARG, "bad arguments");
    if (n == 0) return O3DR_OK;
    if (mem == O3DR_MEM_DEVICE) {
        launch_transform(&c->prof, c->stream, in, n, T, out);
        HIPCHK(hipGetLastError());
        return O3DR_OK;
    }
    const void* in_d;
    CHK(stage_in(c, c->st_in, in, (size_t)n * sizeof(o3dr_point), mem, &in_d));
    launch_transform(&c->prof, c->stream, (const o3dr_point*)in_d, n, T, (o3dr_point*)c->st_in.p);  // in place
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, c->st_in.p, (size_t)n * sizeof(o3dr_point), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return O3DR_OK;
}

// one stand-alone voxel grid over a device cloud -> device destination; returns count + status
// write a host-supplied bounding box (min xyz, max xyz) into bounding-box slot 0 of frame 0
static int put_bbox(o3dr_ctx* c, const float mn[3], const float mx[3])
{
    float* h = (float*)c->misc_host;
    for (int a = 0; a < 3; ++a) h[a] = mn[a], h[3 + a] = mx[a];
    HIPCHK(hipMemcpyAsync(c->ws.mm, h, 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    return O3DR_OK;
}

static int voxel_single(o3dr_ctx* c, const o3dr_point* in_d, int64_t n_in, const float leaf[3], uint32_t min_points,
                        float z_offset, o3dr_point* out_d, int64_t* n_out, uint32_t* status,
                        const float* gmin = nullptr, const float* gmax = nullptr, bool do_sor = false,
                        const float* box_dev = nullptr, const uint8_t* heads_in = nullptr, CloudCounters* big_out = nullptr)
{
    CHK(ws_ensure(c, 1, n_in, false, c->use_runs != 0));
    if (c->small_path && n_in <= kSmallMax && !c->test_corrupt) {  // one launch of one workgroup (kernels/small.inc)
        const float* box = box_dev;
        if (gmin && gmax) {
            CHK(put_bbox(c, gmin, gmax));
            box = c->ws.mm;
        }
        if (do_sor) {  // ... after the outlier removal's five
            CHK(sor_ensure(c, 1, n_in));
            launch_set_counts(&c->prof, c->stream, c->ws.n_valid, (uint32_t)n_in, 1);
            launch_sor_small(&c->prof, c->stream, c->ws, in_d, c->ws.n_valid, n_in, 1.0, c->ws.sor_pts, c->ws.sor_n);
            launch_small_voxel(&c->prof, c->stream, c->ws.sor_pts, c->ws.sor_n, 0, box, leaf, min_points, z_offset, out_d, c->cc_tmp);
        } else {
            launch_small_voxel(&c->prof, c->stream, in_d, nullptr, (uint32_t)n_in, box, leaf, min_points, z_offset, out_d, c->cc_tmp);
        }
        HIPCHK(hipGetLastError());
        CloudCounters cc;
        if (big_out)
            CHK(read_counters(c, c->cc_tmp, &cc, c->cc_big, big_out));
        else
            CHK(read_counters(c, c->cc_tmp, &cc));
        *n_out = (int64_t)cc.count;
        if (status) *status = cc.status;
        return O3DR_OK;
    }
    launch_set_counts(&c->prof, c->stream, c->ws.n_valid, (uint32_t)n_in, 1);
    int mm_used = 1;
    if (gmin && gmax)  // grid laid over a caller-supplied (global) box instead of this cloud's own
        CHK(put_bbox(c, gmin, gmax));
    else if (box_dev)  // the cloud's own box is already known on the device
        HIPCHK(hipMemcpyAsync(c->ws.mm, box_dev, 6 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    else
        mm_used = launch_points_minmax(&c->prof, c->stream, in_d, 0, c->ws.n_valid, 1, n_in, c->ws.mm_stride, c->ws.mm);
    CHK(zero_counters(c, c->cc_tmp));
    VoxelArgs v;
    v.in = in_d;
    v.in_fstride = 0;
    v.n_dev = c->ws.n_valid;
    v.frames = 1;
    v.cap = n_in;
    v.leaf[0] = leaf[0];
    v.leaf[1] = leaf[1];
    v.leaf[2] = leaf[2];
    v.min_points = min_points;
    v.z_offset = z_offset;
    v.out_base = out_d;
    v.cc = c->cc_tmp;
    v.passthrough = 0;
    v.mm_used = mm_used;
    v.stats = c->stats_dev;
    v.use_runs = c->use_runs;
    v.heads_in = heads_in;
    v.test_corrupt = c->test_corrupt;
    c->test_corrupt = 0;
    if (do_sor) {
        CHK(sor_ensure(c, 1, n_in));
        v.mm_used = launch_sor(&c->prof, c->stream, c->ws, in_d, 0, c->ws.n_valid, 1, n_in, mm_used, 1.0, c->ws.sor_pts, 0, c->ws.sor_n);
        v.in = c->ws.sor_pts;
        v.n_dev = c->ws.sor_n;
    }
    launch_voxel_grid(&c->prof, c->stream, c->ws, v);
    HIPCHK(hipGetLastError());
    CloudCounters cc;
    if (big_out)
        CHK(read_counters(c, c->cc_tmp, &cc, c->cc_big, big_out));
    else
        CHK(read_counters(c, c->cc_tmp, &cc));
    *n_out = (int64_t)cc.count;
    if (status) *status = cc.status;
    return O3DR_OK;
}

static int voxel_grid_impl(o3dr_ctx* c, const o3dr_point* in, int64_t n_in, const float leaf[3], uint32_t min_points,
                           float z_offset, o3dr_point* out, int64_t out_capacity, int64_t* n_out, uint32_t* status,
                           int32_t mem, bool do_sor)
{
    if (n_out) *n_out = 0;
    if (status) *status = 0;
    CTX_ENTER(c);
    if (!n_out || n_in < 0 || !leaf || (n_in > 0 && (!in || !out))) return fail(O3DR_ERR_INVALID_ARG, "bad arguments");
    if (!(leaf[0] > 0) || !(leaf[1] > 0) || !(leaf[2] > 0)) return fail(O3DR_ERR_INVALID_ARG, "leaf must be positive");
    if (n_in >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "more than 2^32-1 points in one cloud");
    if (mem != O3DR_MEM_HOST && mem != O3DR_MEM_DEVICE) return fail(O3DR_ERR_INVALID_ARG, "bad mem kind");
    if (n_in == 0) return O3DR_OK;
    if (out_capacity < n_in && mem == O3DR_MEM_DEVICE)
        return fail(O3DR_ERR_CAPACITY, "device output must hold n_in points (overflow fallback returns the input)");
    const void* in_d;
    CHK(stage_in(c, c->st_in, in, (size_t)n_in * sizeof(o3dr_point), mem, &in_d));
    o3dr_point* out_d = out;
    if (mem == O3DR_MEM_HOST) {
        CHK(dev_ensure(c, c->st_out, (size_t)n_in * sizeof(o3dr_point)));
        out_d = (o3dr_point*)c->st_out.p;
    }
    int64_t m = 0;
    uint32_t st = 0;
    CHK(voxel_single(c, (const o3dr_point*)in_d, n_in, leaf, min_points, z_offset, out_d, &m, &st, nullptr, nullptr, do_sor));
    if (mem == O3DR_MEM_HOST) {
        if (m > out_capacity) return fail(O3DR_ERR_CAPACITY, "output buffer too small");
        if (m > 0) {
            HIPCHK(hipMemcpyAsync(out, out_d, (size_t)m * sizeof(o3dr_point), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
    }
    *n_out = m;
    if (status) *status = st;
    return O3DR_OK;
}

extern "C" int o3dr_voxel_grid(o3dr_ctx* c, const o3dr_point* in, int64_t n_in, const float leaf[3], uint32_t min_points,
                               float z_offset, o3dr_point* out, int64_t out_capacity, int64_t* n_out, uint32_t* status,
                               int32_t mem)
{
    return voxel_grid_impl(c, in, n_in, leaf, min_points, z_offset, out, out_capacity, n_out, status, mem, false);
}

// A3b alone: pcl::StatisticalOutlierRemoval with mean_k 50, stddev_mul 1.0 (pose_functions.cpp:1679-1684)
extern "C" int o3dr_statistical_outlier_removal(o3dr_ctx* c, const o3dr_point* in, int64_t n_in, o3dr_point* out,
                                                int64_t out_capacity, int64_t* n_out, int32_t mem)
{
    if (n_out) *n_out = 0;
    CTX_ENTER(c);
    if (!n_out || n_in < 0 || (n_in > 0 && (!in || !out))) return fail(O3DR_ERR_INVALID_ARG, "bad arguments");
    if (n_in >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "more than 2^32-1 points in one cloud");
    if (mem != O3DR_MEM_HOST && mem != O3DR_MEM_DEVICE) return fail(O3DR_ERR_INVALID_ARG, "bad mem kind");
    if (n_in == 0) return O3DR_OK;
    if (out_capacity < n_in) return fail(O3DR_ERR_CAPACITY, "output must hold n_in points");
    const void* in_d;
    CHK(stage_in(c, c->st_in, in, (size_t)n_in * sizeof(o3dr_point), mem, &in_d));
    CHK(ws_ensure(c, 1, n_in, false));
    CHK(sor_ensure(c, 1, n_in));
    launch_set_counts(&c->prof, c->stream, c->ws.n_valid, (uint32_t)n_in, 1);
    o3dr_point* dst = mem == O3DR_MEM_DEVICE ? out : c->ws.sor_pts;
    if (c->small_path && n_in <= kSmallMax) {  // 6 launches instead of ~32 (kernels/small.inc)
        launch_sor_small(&c->prof, c->stream, c->ws, (const o3dr_point*)in_d, c->ws.n_valid, n_in, 1.0, dst, c->ws.sor_n);
    } else {
        const int used = launch_points_minmax(&c->prof, c->stream, (const o3dr_point*)in_d, 0, c->ws.n_valid, 1, n_in, c->ws.mm_stride, c->ws.mm);
        launch_sor(&c->prof, c->stream, c->ws, (const o3dr_point*)in_d, 0, c->ws.n_valid, 1, n_in, used, 1.0, dst, 0, c->ws.sor_n);
    }
    HIPCHK(hipGetLastError());
    uint32_t m = 0;
    CHK(read_u32(c, c->ws.sor_n, &m));
    if (mem == O3DR_MEM_HOST && m > 0) {
        HIPCHK(hipMemcpyAsync(out, dst, (size_t)m * sizeof(o3dr_point), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *n_out = (int64_t)m;
    return O3DR_OK;
}

static void downsample_leaf(const o3dr_params& p, int combined, float leaf[3], uint32_t* min_pts, float* z_offset)
{
    if (combined) {  // pose_functions.cpp:1666,1693-1694
        leaf[0] = leaf[1] = (float)p.voxel_size;
        leaf[2] = 1000.f;
        *min_pts = p.min_points_per_voxel;
        *z_offset = 500.f;
    } else {  // :1698
        leaf[0] = leaf[1] = leaf[2] = (float)(p.voxel_size / 5);
        *min_pts = 0;
        *z_offset = 0.f;
    }
}

extern "C" int o3dr_downsample_pt_cloud(o3dr_ctx* c, const o3dr_point* in, int64_t n_in, int32_t combined,
                                        o3dr_point* out, int64_t out_capacity, int64_t* n_out, uint32_t* status,
                                        int32_t mem)
{
    if (!c) {
        if (n_out) *n_out = 0;
        return fail(O3DR_ERR_INVALID_ARG, "ctx is NULL");
    }
    float leaf[3], zo;
    uint32_t mp;
    downsample_leaf(c->params, combined, leaf, &mp, &zo);
    // statistical outlier removal iff !combinedPtCloud && jump_pixels > 0 (pose_functions.cpp:1673), when enabled
    return voxel_grid_impl(c, in, n_in, leaf, mp, zo, out, out_capacity, n_out, status, mem, !combined && sor_on(c));
}

// -------------------------------------------------------------------------------------------------
// A7: device-resident accumulation
// -------------------------------------------------------------------------------------------------
// the group-run head flags of cloud_big: one byte per 4 points (+ slack for the last wave's word), tied to cloud_cap.
// keep: carry the flags recorded so far over.  On failure the old buffer is gone and nothing is recorded.
static inline size_t heads_bytes(int64_t points) { return ((size_t)points / 4 + 64 + 3) & ~(size_t)3; }
static int heads_resize(o3dr_ctx* c, int64_t points, bool keep)
{
    if (c->cloud_heads && c->cloud_heads_cap >= points) return O3DR_OK;
    const size_t bytes = heads_bytes(points);
    uint32_t* nf = nullptr;
    if (hipMalloc((void**)&nf, bytes) != hipSuccess) {
        (void)hipGetLastError();
        nf = nullptr;
    }
    if (nf && hipMemsetAsync(nf, 0, bytes, c->stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(nf);
        nf = nullptr;
    }
    if (c->cloud_heads) {
        if (nf && keep) {
            const size_t old = heads_bytes(c->cloud_heads_cap);
            (void)hipMemcpyAsync(nf, c->cloud_heads, old < bytes ? old : bytes, hipMemcpyDeviceToDevice, c->stream);
        }
        (void)hipStreamSynchronize(c->stream);  // earlier launches may still write the old flags
        (void)hipFree(c->cloud_heads);
    }
    c->cloud_heads = nf;
    c->cloud_heads_cap = nf ? points : 0;
    return nf ? O3DR_OK : fail(O3DR_ERR_ALLOC, "hipMalloc failed (cloud_big run heads)");
}

static int cloud_reserve(o3dr_ctx* c, int64_t need)
{
    if (need <= c->cloud_cap) return O3DR_OK;
    int64_t want = c->cloud_cap * 2;
    if (want < need) want = need;
    o3dr_point* nb = nullptr;
    if (hipMalloc((void**)&nb, (size_t)want * sizeof(o3dr_point)) != hipSuccess) {
        (void)hipGetLastError();
        want = need;
        if (hipMalloc((void**)&nb, (size_t)want * sizeof(o3dr_point)) != hipSuccess)
            return fail(O3DR_ERR_ALLOC, "hipMalloc failed (cloud_big)");
    }
    if (c->cloud_big) {
        CloudCounters cc;
        CHK(read_counters(c, c->cc_big, &cc));
        if (cc.count)
            HIPCHK(hipMemcpyAsync(nb, c->cloud_big, (size_t)cc.count * sizeof(o3dr_point), hipMemcpyDeviceToDevice,
                                  c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipFree(c->cloud_big));
        c->cloud_ub = (int64_t)cc.count;
    }
    c->cloud_big = nb;
    c->cloud_cap = want;
    // the flag buffer follows the cloud's capacity (zeros where nothing was recorded yet); without it the heads are
    // simply not recorded (the merge then reads the points for them): never a reason to fail the reservation
    if (c->cloud_box_enable && heads_resize(c, want, true) != O3DR_OK) c->cloud_heads_valid = false;
    return O3DR_OK;
}

// make room for `extra` more points; tightens the host-side bound with one sync when it must
static int cloud_make_room(o3dr_ctx* c, int64_t extra)
{
    if (c->cloud_ub + extra > c->cloud_cap) {
        CloudCounters cc;
        CHK(read_counters(c, c->cc_big, &cc));
        c->cloud_ub = (int64_t)cc.count;
        if (c->cloud_ub + extra > c->cloud_cap) CHK(cloud_reserve(c, c->cloud_ub + extra));
    }
    return O3DR_OK;
}

// the alternate cloud buffer with room for `need` points (contents undefined)
static int alt_reserve(o3dr_ctx* c, int64_t need)
{
    if (need <= c->cloud_alt_cap) return O3DR_OK;
    if (c->cloud_alt) {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipFree(c->cloud_alt));
        c->cloud_alt = nullptr;
        c->cloud_alt_cap = 0;
    }
    if (hipMalloc((void**)&c->cloud_alt, (size_t)need * sizeof(o3dr_point)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(O3DR_ERR_ALLOC, "hipMalloc failed (alternate cloud buffer)");
    }
    c->cloud_alt_cap = need;
    return O3DR_OK;
}
static void swap_clouds(o3dr_ctx* c)
{
    o3dr_point* p = c->cloud_big;
    const int64_t cap = c->cloud_cap;
    c->cloud_big = c->cloud_alt;
    c->cloud_cap = c->cloud_alt_cap;
    c->cloud_alt = p;
    c->cloud_alt_cap = cap;
}

extern "C" int o3dr_cloud_big_reserve(o3dr_ctx* c, int64_t n_points)
{
    CTX_ENTER(c);
    if (n_points < 0) return fail(O3DR_ERR_INVALID_ARG, "negative size");
    return cloud_reserve(c, n_points);
}

static int cloud_box_clear(o3dr_ctx* c)
{
    static const float empty[6] = {__builtin_inff(), __builtin_inff(), __builtin_inff(),
                                   -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    HIPCHK(hipMemcpyAsync(c->cloud_box, empty, sizeof empty, hipMemcpyHostToDevice, c->stream));
    c->cloud_box_valid = c->cloud_box_enable != 0;
    // the flag buffer must cover the cloud buffer in use NOW: partition / adopt swap in the alternate buffer, whose
    // capacity the flags know nothing about (a smaller flag buffer would be written past its end by the next appends)
    if (c->cloud_box_enable && c->cloud_cap > 0 && c->cloud_heads_cap < c->cloud_cap) (void)heads_resize(c, c->cloud_cap, false);
    if (c->cloud_heads) HIPCHK(hipMemsetAsync(c->cloud_heads, 0, heads_bytes(c->cloud_heads_cap), c->stream));
    c->cloud_heads_valid = c->cloud_box_enable != 0 && c->cloud_heads && c->cloud_heads_cap >= c->cloud_cap;
    return O3DR_OK;
}

// frame calls that append to cloud_big also record where its group runs start, for the merge's grid
static void heads_for_append(o3dr_ctx* c, VoxelArgs& v)
{
    if (!c->cloud_heads_valid || !c->cloud_heads) return;
    if (c->cloud_heads_cap < c->cloud_cap) {  // (never expected: cloud_reserve and cloud_box_clear keep them together)
        c->cloud_heads_valid = false;
        return;
    }
    float leaf[3], zo;
    uint32_t mp;
    downsample_leaf(c->params, 1, leaf, &mp, &zo);
    const bool same = leaf[0] == c->cloud_heads_leaf[0] && leaf[1] == c->cloud_heads_leaf[1] && leaf[2] == c->cloud_heads_leaf[2] &&
                      zo == c->cloud_heads_zo;
    if (c->cloud_ub == 0) {  // an empty cloud takes the current leaf
        for (int a = 0; a < 3; ++a) c->cloud_heads_leaf[a] = leaf[a];
        c->cloud_heads_zo = zo;
    } else if (!same) {  // voxel_size changed while the cloud was growing: the flags describe no single grid
        c->cloud_heads_valid = false;
        return;
    }
    v.cloud_heads.flags = c->cloud_heads;
    for (int a = 0; a < 3; ++a) v.cloud_heads.inv[a] = 1.0f / leaf[a];
    v.cloud_heads.z_offset = zo;
}
// ... and the merge takes them instead of reading the cloud once more
static const uint8_t* heads_for_merge(o3dr_ctx* c, const float leaf[3], float zo)
{
    if (!c->cloud_heads_valid || !c->cloud_heads) return nullptr;
    if (leaf[0] != c->cloud_heads_leaf[0] || leaf[1] != c->cloud_heads_leaf[1] || leaf[2] != c->cloud_heads_leaf[2] ||
        zo != c->cloud_heads_zo)
        return nullptr;
    return reinterpret_cast<const uint8_t*>(c->cloud_heads);
}

extern "C" int o3dr_cloud_big_reset(o3dr_ctx* c)
{
    CTX_ENTER(c);
    CHK(zero_counters(c, c->cc_big));
    CHK(cloud_box_clear(c));
    c->cloud_ub = 0;
    c->cloud_n_exact = true;
    return O3DR_OK;
}

extern "C" int o3dr_cloud_big_size(o3dr_ctx* c, int64_t* n, uint32_t* status)
{
    if (n) *n = 0;
    CTX_ENTER(c);
    CloudCounters cc;
    CHK(read_counters(c, c->cc_big, &cc));
    c->cloud_ub = (int64_t)cc.count;
    if (n) *n = (int64_t)cc.count;
    if (status) *status = cc.status;
    return O3DR_OK;
}

extern "C" int o3dr_cloud_big_read(o3dr_ctx* c, o3dr_point* out, int64_t out_capacity, int64_t* n_out, int32_t mem)
{
    if (n_out) *n_out = 0;
    CTX_ENTER(c);
    if (!n_out) return fail(O3DR_ERR_INVALID_ARG, "n_out is NULL");
    CloudCounters cc;
    CHK(read_counters(c, c->cc_big, &cc));
    const int64_t n = (int64_t)cc.count;
    if (n > out_capacity) return fail(O3DR_ERR_CAPACITY, "output buffer too small");
    if (n > 0) {
        if (!out) return fail(O3DR_ERR_INVALID_ARG, "out is NULL");
        HIPCHK(hipMemcpyAsync(out, c->cloud_big, (size_t)n * sizeof(o3dr_point),
                              mem == O3DR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *n_out = n;
    return O3DR_OK;
}

// appends n points and bumps the device counter (passthrough voxel job with a known count)
extern "C" int o3dr_cloud_big_append(o3dr_ctx* c, const o3dr_point* pts, int64_t n, int32_t mem)
{
    CTX_ENTER(c);
    if (n < 0 || (n > 0 && !pts)) return fail(O3DR_ERR_INVALID_ARG, "bad arguments");
    if (n == 0) return O3DR_OK;
    if (n >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "more than 2^32-1 points in one append");
    CHK(cloud_make_room(c, n));
    const void* src;
    CHK(stage_in(c, c->st_in, pts, (size_t)n * sizeof(o3dr_point), mem, &src));
    CHK(ws_ensure(c, 1, 1, false));
    launch_set_counts(&c->prof, c->stream, c->ws.n_valid, (uint32_t)n, 1);
    VoxelArgs v;
    memset(&v, 0, sizeof v);
    v.in = (const o3dr_point*)src;
    v.n_dev = c->ws.n_valid;
    v.frames = 1;
    v.cap = n;
    v.leaf[0] = v.leaf[1] = v.leaf[2] = 1.f;
    v.out_base = c->cloud_big;
    v.cc = c->cc_big;
    v.passthrough = 1;
    v.mm_used = 0;
    v.stats = nullptr;
    c->cloud_box_valid = false;  // appended points are not tracked
    c->cloud_heads_valid = false;
    launch_voxel_grid(&c->prof, c->stream, c->ws, v);
    HIPCHK(hipGetLastError());
    c->cloud_ub += n;
    if (mem == O3DR_MEM_HOST) HIPCHK(hipStreamSynchronize(c->stream));
    return O3DR_OK;
}

extern "C" int o3dr_cloud_big_transform(o3dr_ctx* c, const float T[16])
{
    CTX_ENTER(c);
    if (!T) return fail(O3DR_ERR_INVALID_ARG, "T is NULL");
    CloudCounters cc;
    CHK(read_counters(c, c->cc_big, &cc));
    launch_transform(&c->prof, c->stream, c->cloud_big, (int64_t)cc.count, T, c->cloud_big);
    HIPCHK(hipGetLastError());
    c->cloud_box_valid = false;
    c->cloud_heads_valid = false;
    return O3DR_OK;
}

static int accumulate_impl(o3dr_ctx* c, const uint8_t* disp, int64_t disp_frame_stride, int64_t disp_pitch,
                           const uint8_t* bgr, int64_t bgr_frame_stride, int64_t bgr_pitch, int32_t rows, int32_t cols,
                           const float* poses, int32_t n_frames, const float* kp_xy, const int64_t* kp_offsets, int32_t mem)
{
    CTX_ENTER(c);
    if (!c->has_Q) return fail(O3DR_ERR_NOT_CONFIGURED, "o3dr_set_camera has not been called");
    if (n_frames < 0 || (n_frames > 0 && !poses)) return fail(O3DR_ERR_INVALID_ARG, "bad frame list");
    if (mem != O3DR_MEM_HOST && mem != O3DR_MEM_DEVICE) return fail(O3DR_ERR_INVALID_ARG, "bad mem kind");
    if (n_frames == 0) return O3DR_OK;
    CHK(check_images(c, disp, disp_pitch, bgr, bgr_pitch, rows, cols, disp_frame_stride));
    if (disp_frame_stride < (int64_t)rows * disp_pitch || bgr_frame_stride < (int64_t)rows * bgr_pitch)
        return fail(O3DR_ERR_INVALID_ARG, "frame stride smaller than a frame");
    const GridShape g = grid_shape(c->params, rows, cols);
    // keypoint pass (pose_functions.cpp:1057-1091): active iff jump_pixels != 1, emitted before the grid points
    int64_t kp_total = 0, kp_max = 0;
    if (kp_offsets && c->params.jump_pixels != 1) {
        for (int f = 0; f < n_frames; ++f) {
            const int64_t k = kp_offsets[f + 1] - kp_offsets[f];
            if (k < 0) return fail(O3DR_ERR_INVALID_ARG, "kp_offsets must not decrease");
            if (k > kp_max) kp_max = k;
        }
        kp_total = kp_offsets[n_frames] - kp_offsets[0];
        if (kp_total > 0 && !kp_xy) return fail(O3DR_ERR_INVALID_ARG, "kp_xy is NULL");
        if (kp_total >= (int64_t)INT32_MAX) return fail(O3DR_ERR_INVALID_ARG, "too many keypoints");
    }
    const bool use_kp = kp_total > 0;
    const int64_t cap = g.n + kp_max;  // points a frame can produce: capacity of every per-frame buffer below
    if (cap == 0) return O3DR_OK;      // jump_pixels == 0 without keypoints: nothing to add
    const float* kp_d = nullptr;
    const int32_t* kpoff_d = nullptr;
    std::vector<int32_t> kp_rel;
    if (use_kp) {
        const void* p;
        CHK(stage_in(c, c->st_kp, kp_xy + 2 * kp_offsets[0], (size_t)kp_total * 2 * sizeof(float), mem, &p));
        kp_d = (const float*)p;
        kp_rel.resize((size_t)n_frames + 1);
        for (int f = 0; f <= n_frames; ++f) kp_rel[f] = (int32_t)(kp_offsets[f] - kp_offsets[0]);
        HIPCHK(hipStreamSynchronize(c->stream));  // an earlier call's launches may still read the offsets
        CHK(dev_ensure(c, c->st_kpoff, kp_rel.size() * sizeof(int32_t)));
        HIPCHK(hipMemcpy(c->st_kpoff.p, kp_rel.data(), kp_rel.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        kpoff_d = (const int32_t*)c->st_kpoff.p;
    }
    float leaf[3], zo;
    uint32_t mp;
    downsample_leaf(c->params, 0, leaf, &mp, &zo);
    // statistical outlier removal (pose_functions.cpp:1673-1686, in front of every per-frame voxel grid): batched like
    // everything else - grid y = frame in all of its kernels, no host round trip per frame
    const bool with_sor = sor_on(c) && !c->params.dont_downsample;
    int B = n_frames < c->max_batch ? n_frames : c->max_batch;
    {   // ~56 bytes of workspace per candidate point; keep a batch under 12 GiB of HBM (of 288)
        const int64_t per_frame = (with_sor ? 56 + 50 : 56) * cap + (1 << 20);
        const int64_t fit = ((int64_t)12 << 30) / per_frame;
        if (fit < B) B = fit < 1 ? 1 : (int)fit;
    }
    // Host buffers: frames cross PCIe once.  Upload batch k+1 on a second stream while batch k
    // computes (two staging sets); smaller batches than the HBM-resident path so that they overlap
    // (clamped BEFORE the workspaces are sized: a streaming call never launches more than host_batch frames)
    const bool streaming = mem == O3DR_MEM_HOST;
    if (streaming && c->host_batch < B) B = c->host_batch;
    CHK(ws_ensure(c, B, cap, true));
    if (with_sor) CHK(sor_ensure(c, B, cap));

    if (streaming) {
        if (!c->copy_stream) HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            if (!c->ev_copied[i]) HIPCHK(hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming));
            if (!c->ev_done[i]) HIPCHK(hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
        }
        HIPCHK(hipStreamSynchronize(c->stream));  // staging sets are free, events start clean
    }
    int slot = 0;
    for (int f0 = 0; f0 < n_frames; f0 += B, slot ^= 1) {
        const int nb = (n_frames - f0) < B ? (n_frames - f0) : B;
        CHK(cloud_make_room(c, (int64_t)nb * cap));
        const void *disp_d, *bgr_d, *poses_d;
        if (streaming) {
            const size_t db = (size_t)nb * disp_frame_stride, cb = (size_t)nb * bgr_frame_stride, pb = (size_t)nb * 16 * sizeof(float);
            CHK(dev_ensure(c, c->st2_disp[slot], db));
            CHK(dev_ensure(c, c->st2_bgr[slot], cb));
            CHK(dev_ensure(c, c->st2_poses[slot], pb));
            if (f0 >= 2 * B) HIPCHK(hipStreamWaitEvent(c->copy_stream, c->ev_done[slot], 0));  // set's previous user finished
            HIPCHK(hipMemcpyAsync(c->st2_disp[slot].p, disp + (int64_t)f0 * disp_frame_stride, db, hipMemcpyHostToDevice, c->copy_stream));
            HIPCHK(hipMemcpyAsync(c->st2_bgr[slot].p, bgr + (int64_t)f0 * bgr_frame_stride, cb, hipMemcpyHostToDevice, c->copy_stream));
            HIPCHK(hipMemcpyAsync(c->st2_poses[slot].p, poses + 16 * (int64_t)f0, pb, hipMemcpyHostToDevice, c->copy_stream));
            HIPCHK(hipEventRecord(c->ev_copied[slot], c->copy_stream));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_copied[slot], 0));
            disp_d = c->st2_disp[slot].p;
            bgr_d = c->st2_bgr[slot].p;
            poses_d = c->st2_poses[slot].p;
        } else {
            disp_d = disp + (int64_t)f0 * disp_frame_stride;
            bgr_d = bgr + (int64_t)f0 * bgr_frame_stride;
            poses_d = poses + 16 * (int64_t)f0;
        }
        const uint8_t* dsp = (const uint8_t*)disp_d;
        int64_t dsp_pitch = disp_pitch, dsp_fstride = disp_frame_stride;
        CHK(maybe_blur(c, &dsp, &dsp_pitch, &dsp_fstride, rows, cols, nb));
        ReprojectArgs a;
        fill_args(c, a, dsp, dsp_pitch, dsp_fstride, (const uint8_t*)bgr_d, bgr_pitch,
                  bgr_frame_stride, rows, cols, g, cap);
        a.xf_mode = 2;
        a.poses = (const float*)poses_d;
        for (int i = 0; i < 3; ++i) a.slab_inv[i] = 1.0f / leaf[i];
        a.slab_shift = slab_shift_for(c, rows, cols, leaf);
        launch_minmax_init(&c->prof, c->stream, c->ws.mm, c->ws.mm_stride, a.n_tiles, c->ws.n_kp, nb);
        // no keypoint pass in front of the grid pass and a voxel grid behind it: index and first digit histogram are
        // produced by the pass that writes the points (the keypoint pass would need them too: it keeps the two-step form)
        const bool fused = !use_kp && !c->params.dont_downsample && a.n_tiles > 0 && !with_sor;
        if (use_kp)
            launch_keypoint_pass(&c->prof, c->stream, a, kp_d, 0, c->ws.pts, c->ws.n_kp, c->ws.mm, kpoff_d + f0, nb);
        if (fused)
            launch_reproject_fused(&c->prof, c->stream, c->ws, a, nb, cap, leaf, !c->exact_box);
        else
            launch_reproject(&c->prof, c->stream, a, nb, c->ws.pts, c->ws.tile_cnt, c->ws.n_kp, c->ws.n_valid, c->ws.mm,
                             c->ws.scan_partial);
        VoxelArgs v;
        v.keys_ready = fused ? 1 : 0;
        v.cloud_box = c->cloud_box_valid ? c->cloud_box : nullptr;
        heads_for_append(c, v);
        v.in = c->ws.pts;
        v.in_fstride = cap;
        v.n_dev = c->ws.n_valid;
        v.frames = nb;
        v.cap = cap;
        v.leaf[0] = leaf[0];
        v.leaf[1] = leaf[1];
        v.leaf[2] = leaf[2];
        v.min_points = mp;
        v.z_offset = zo;
        v.out_base = c->cloud_big;
        v.cc = c->cc_big;
        v.passthrough = c->params.dont_downsample ? 1 : 0;
        v.mm_used = a.n_tiles + 1;
        v.stats = c->stats_dev;
        v.use_runs = 0;
        if (with_sor) {
            v.mm_used = launch_sor(&c->prof, c->stream, c->ws, c->ws.pts, cap, c->ws.n_valid, nb, cap, a.n_tiles + 1, 1.0, c->ws.sor_pts,
                                   cap, c->ws.sor_n);
            v.in = c->ws.sor_pts;
            v.n_dev = c->ws.sor_n;
        }
        launch_voxel_grid(&c->prof, c->stream, c->ws, v);
        HIPCHK(hipGetLastError());
        c->cloud_ub += (int64_t)nb * cap;
        c->cloud_n_exact = false;
        if (streaming) HIPCHK(hipEventRecord(c->ev_done[slot], c->stream));
    }
    if (streaming) HIPCHK(hipStreamSynchronize(c->stream));  // the caller may reuse its host buffers
    return O3DR_OK;
}

extern "C" int o3dr_accumulate_frames(o3dr_ctx* c, const uint8_t* disp, int64_t disp_frame_stride, int64_t disp_pitch,
                                      const uint8_t* bgr, int64_t bgr_frame_stride, int64_t bgr_pitch, int32_t rows,
                                      int32_t cols, const float* poses, int32_t n_frames, int32_t mem)
{
    return accumulate_impl(c, disp, disp_frame_stride, disp_pitch, bgr, bgr_frame_stride, bgr_pitch, rows, cols, poses, n_frames,
                           nullptr, nullptr, mem);
}
extern "C" int o3dr_accumulate_frames_kp(o3dr_ctx* c, const uint8_t* disp, int64_t disp_frame_stride, int64_t disp_pitch,
                                         const uint8_t* bgr, int64_t bgr_frame_stride, int64_t bgr_pitch, int32_t rows,
                                         int32_t cols, const float* poses, int32_t n_frames, const float* kp_xy,
                                         const int64_t* kp_offsets, int32_t mem)
{
    return accumulate_impl(c, disp, disp_frame_stride, disp_pitch, bgr, bgr_frame_stride, bgr_pitch, rows, cols, poses, n_frames,
                           kp_xy, kp_offsets, mem);
}

static int finalize_impl(o3dr_ctx* c, const float* gmin, const float* gmax, o3dr_point* out, int64_t out_capacity,
                         int64_t* n_out, uint32_t* status, int32_t mem)
{
    if (n_out) *n_out = 0;
    if (status) *status = 0;
    CTX_ENTER(c);
    if (!n_out) return fail(O3DR_ERR_INVALID_ARG, "n_out is NULL");
    if (mem != O3DR_MEM_HOST && mem != O3DR_MEM_DEVICE) return fail(O3DR_ERR_INVALID_ARG, "bad mem kind");
    // the host knows the count exactly after a read, a reset or an adopt (the multi-GPU exchange): then the status bits
    // accumulated in cc_big come back with the result's count, in the one round trip the merge needs anyway
    CloudCounters cc;
    const bool known = c->cloud_n_exact && !c->params.dont_downsample && c->cloud_ub > 0;
    if (known) {
        cc.count = (uint64_t)c->cloud_ub;
        cc.status = 0;
    } else {
        CHK(read_counters(c, c->cc_big, &cc));
    }
    const int64_t n = (int64_t)cc.count;
    if (n == 0) return O3DR_OK;
    if (n >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "cloud_big exceeds 2^32-1 points");
    if (!out) return fail(O3DR_ERR_INVALID_ARG, "out is NULL");
    if (c->params.dont_downsample) {  // pose.cpp:534-537: cloud_small = cloud_big
        if (n > out_capacity) return fail(O3DR_ERR_CAPACITY, "output buffer too small");
        return o3dr_cloud_big_read(c, out, out_capacity, n_out, mem);
    }
    float leaf[3], zo;
    uint32_t mp;
    downsample_leaf(c->params, 1, leaf, &mp, &zo);
    o3dr_point* out_d = out;
    if (mem == O3DR_MEM_HOST) {
        CHK(dev_ensure(c, c->st_out, (size_t)n * sizeof(o3dr_point)));
        out_d = (o3dr_point*)c->st_out.p;
    } else if (out_capacity < n) {
        return fail(O3DR_ERR_CAPACITY, "device output must hold cloud_big (overflow fallback returns the input)");
    }
    int64_t m = 0;
    uint32_t st = 0;
    CHK(voxel_single(c, c->cloud_big, n, leaf, mp, zo, out_d, &m, &st, gmin, gmax, false,
                     c->cloud_box_valid ? c->cloud_box : nullptr, heads_for_merge(c, leaf, zo), known ? &cc : nullptr));
    if ((int64_t)cc.count != n) return fail(O3DR_ERR_INTERNAL, "cloud_big's device count differs from the host's");
    if (mem == O3DR_MEM_HOST) {
        if (m > out_capacity) return fail(O3DR_ERR_CAPACITY, "output buffer too small");
        if (m > 0) {
            HIPCHK(hipMemcpyAsync(out, out_d, (size_t)m * sizeof(o3dr_point), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
    }
    *n_out = m;
    if (status) *status = st | cc.status;
    return O3DR_OK;
}

extern "C" int o3dr_finalize(o3dr_ctx* c, o3dr_point* out, int64_t out_capacity, int64_t* n_out, uint32_t* status,
                             int32_t mem)
{
    return finalize_impl(c, nullptr, nullptr, out, out_capacity, n_out, status, mem);
}

// ---- multi-GPU merge: global box -> index-slice partition -> exchange -> local merge ----
extern "C" int o3dr_finalize_global(o3dr_ctx* c, const float gmin[3], const float gmax[3], o3dr_point* out,
                                    int64_t out_capacity, int64_t* n_out, uint32_t* status, int32_t mem)
{
    if (!gmin || !gmax) {
        if (n_out) *n_out = 0;
        return fail(O3DR_ERR_INVALID_ARG, "bounding box is NULL");
    }
    return finalize_impl(c, gmin, gmax, out, out_capacity, n_out, status, mem);
}

// Zero-copy access for the exchange: the HBM address of cloud_big (valid until the next call that
// appends, partitions or adopts), a receive buffer of the requested size, and "make what I received
// the new cloud_big".
extern "C" int o3dr_cloud_big_view(o3dr_ctx* c, void** ptr, int64_t* n)
{
    CTX_ENTER(c);
    if (!ptr || !n) return fail(O3DR_ERR_INVALID_ARG, "ptr / n is NULL");
    if (!c->cloud_n_exact) {  // (no round trip when the host already knows the size: reads, resets, adopt, assume_size)
        CloudCounters cc;
        CHK(read_counters(c, c->cc_big, &cc));
    }
    *ptr = c->cloud_big;
    *n = c->cloud_ub;
    return O3DR_OK;
}
// The caller learnt cloud_big's exact size by other means (its own header from o3dr_cloud_big_header_dev, read back
// with the all-to-all's sizes): later calls then need no round trip for it.  A wrong value is caught by the merge
// (O3DR_ERR_INTERNAL), never used to address memory beyond the cloud's capacity.
extern "C" int o3dr_cloud_big_assume_size(o3dr_ctx* c, int64_t n_points)
{
    CTX_ENTER(c);
    if (n_points < 0 || n_points > c->cloud_ub) return fail(O3DR_ERR_INVALID_ARG, "size above what the calls so far can have produced");
    c->cloud_ub = n_points;
    c->cloud_n_exact = true;
    return O3DR_OK;
}
extern "C" int o3dr_cloud_big_recv_buffer(o3dr_ctx* c, int64_t n_points, void** ptr)
{
    CTX_ENTER(c);
    if (!ptr || n_points < 0) return fail(O3DR_ERR_INVALID_ARG, "bad arguments");
    CHK(alt_reserve(c, n_points > 0 ? n_points : 1));
    *ptr = c->cloud_alt;
    return O3DR_OK;
}
extern "C" int o3dr_cloud_big_adopt(o3dr_ctx* c, int64_t n_points)
{
    CTX_ENTER(c);
    if (n_points < 0 || n_points > c->cloud_alt_cap) return fail(O3DR_ERR_INVALID_ARG, "more points than the receive buffer holds");
    if (n_points >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "cloud_big exceeds 2^32-1 points");
    // stream-ordered, no host round trip: the count is set by a one-thread kernel (the status bits accumulated so far
    // are kept); what filled the receive buffer must be ordered before this stream's next work by the caller
    launch_set_cloud_count(c->stream, c->cc_big, (uint64_t)n_points);
    HIPCHK(hipGetLastError());
    swap_clouds(c);
    c->cloud_box_valid = false;
    c->cloud_heads_valid = false;
    c->cloud_ub = n_points;
    c->cloud_n_exact = true;
    return O3DR_OK;
}

// ---- the exchange's small data on the device: header (box + count) and slice counts without a host round trip ----------
extern "C" int o3dr_cloud_big_header_dev(o3dr_ctx* c, void* hdr_dev)
{
    CTX_ENTER(c);
    if (!hdr_dev) return fail(O3DR_ERR_INVALID_ARG, "hdr_dev is NULL");
    const float* box = c->cloud_box;
    if (!c->cloud_box_valid) {
        float* tmp = (float*)c->misc_dev;
        if (c->cloud_ub == 0 || !c->cloud_big) {
            static const float empty[6] = {__builtin_inff(), __builtin_inff(), __builtin_inff(),
                                           -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            HIPCHK(hipMemcpyAsync(tmp, empty, sizeof empty, hipMemcpyHostToDevice, c->stream));
        } else {  // appends / transforms / exchanges dropped the running box: one pass over the cloud, sized by the bound
            if (c->cloud_ub >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "cloud_big exceeds 2^32-1 points");
            CHK(ws_ensure(c, 1, c->cloud_ub, false));
            launch_count_from_cc(c->stream, c->cc_big, c->ws.n_valid);
            const int used = launch_points_minmax(&c->prof, c->stream, c->cloud_big, 0, c->ws.n_valid, 1, c->cloud_ub, c->ws.mm_stride, c->ws.mm);
            launch_bbox(&c->prof, c->stream, c->ws.mm, used, tmp);
        }
        box = tmp;
    }
    launch_pack_header(c->stream, box, c->cc_big, hdr_dev);
    HIPCHK(hipGetLastError());
    return O3DR_OK;
}

extern "C" int o3dr_cloud_big_partition_dev(o3dr_ctx* c, const void* hdrs_dev, int32_t n_hdrs, int32_t n_parts, int64_t* counts_dev)
{
    CTX_ENTER(c);
    if (!hdrs_dev || !counts_dev || n_hdrs < 1 || n_parts < 1 || n_parts > kMaxRadix)
        return fail(O3DR_ERR_INVALID_ARG, "bad arguments (1 <= n_parts <= 128)");
    HIPCHK(hipMemsetAsync(counts_dev, 0, sizeof(int64_t) * ((size_t)n_parts + 1), c->stream));
    const int64_t ub = c->cloud_ub;  // kernels take the count from the device; the bound sizes grids and buffers
    if (ub == 0 || !c->cloud_big) return O3DR_OK;
    if (ub >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "cloud_big exceeds 2^32-1 points");
    CHK(ws_ensure(c, 1, ub, false));
    CHK(alt_reserve(c, ub));
    o3dr_point* nb = c->cloud_alt;
    launch_count_from_cc(c->stream, c->cc_big, c->ws.n_valid);
    float leaf[3], zo;
    uint32_t mp;
    downsample_leaf(c->params, 1, leaf, &mp, &zo);
    VoxelArgs v;
    memset(&v, 0, sizeof v);
    v.in = c->cloud_big;
    v.n_dev = c->ws.n_valid;
    v.frames = 1;
    v.cap = ub;
    v.leaf[0] = leaf[0];
    v.leaf[1] = leaf[1];
    v.leaf[2] = leaf[2];
    v.z_offset = zo;
    launch_partition(&c->prof, c->stream, c->ws, v, n_parts, nb, (uint64_t*)counts_dev, (uint32_t*)(counts_dev + n_parts), hdrs_dev, n_hdrs);
    if (hipGetLastError() != hipSuccess) return fail(O3DR_ERR_HIP, "partition launch failed");
    swap_clouds(c);  // the partitioned copy becomes cloud_big; the old buffer is kept as the alternate
    c->cloud_heads_valid = false;
    return O3DR_OK;
}

// ---- the partition in two halves: slice sizes first (nothing moves), then ONE pass that places the slices where the
// exchange wants them - [room for what the lower ranks send | this rank's own slice | room for the higher ranks' | the
// slices that leave, in rank order] - so that the all-to-all receives straight into the gaps and the own slice (95 % of the
// cloud at the benchmark shapes) is never sent to itself
static int slice_args(o3dr_ctx* c, VoxelArgs& v, int64_t ub)
{
    float leaf[3], zo;
    uint32_t mp;
    downsample_leaf(c->params, 1, leaf, &mp, &zo);
    memset(&v, 0, sizeof v);
    v.in = c->cloud_big;
    v.n_dev = c->ws.n_valid;
    v.frames = 1;
    v.cap = ub;
    v.leaf[0] = leaf[0];
    v.leaf[1] = leaf[1];
    v.leaf[2] = leaf[2];
    v.z_offset = zo;
    return O3DR_OK;
}
extern "C" int o3dr_cloud_big_slice_counts_dev(o3dr_ctx* c, const void* hdrs_dev, int32_t n_hdrs, int32_t n_parts, int64_t* counts_dev)
{
    CTX_ENTER(c);
    if (!hdrs_dev || !counts_dev || n_hdrs < 1 || n_parts < 1 || n_parts > kMaxRadix)
        return fail(O3DR_ERR_INVALID_ARG, "bad arguments (1 <= n_parts <= 128)");
    c->place_ub = -1;
    HIPCHK(hipMemsetAsync(counts_dev, 0, sizeof(int64_t) * ((size_t)n_parts + 1), c->stream));
    const int64_t ub = c->cloud_ub;  // kernels take the count from the device; the bound sizes grids and buffers
    if (ub == 0 || !c->cloud_big) {
        c->place_ub = 0;
        c->place_parts = n_parts;
        return O3DR_OK;
    }
    if (ub >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "cloud_big exceeds 2^32-1 points");
    CHK(ws_ensure(c, 1, ub, false));
    launch_count_from_cc(c->stream, c->cc_big, c->ws.n_valid);
    VoxelArgs v;
    CHK(slice_args(c, v, ub));
    launch_partition_count(&c->prof, c->stream, c->ws, v, n_parts, (uint64_t*)counts_dev, (uint32_t*)(counts_dev + n_parts), hdrs_dev, n_hdrs);
    if (hipGetLastError() != hipSuccess) return fail(O3DR_ERR_HIP, "slice count launch failed");
    c->place_ub = ub;
    c->place_parts = n_parts;
    return O3DR_OK;
}

extern "C" int o3dr_cloud_big_place_slices(o3dr_ctx* c, int32_t n_parts, int32_t own_part, const int64_t* counts, int64_t n_before,
                                           int64_t n_after, int64_t* send_offset)
{
    CTX_ENTER(c);
    if (!counts || n_parts < 1 || n_parts > kMaxRadix || own_part < 0 || own_part >= n_parts || n_before < 0 || n_after < 0)
        return fail(O3DR_ERR_INVALID_ARG, "bad arguments");
    if (c->place_ub < 0 || c->place_parts != n_parts) return fail(O3DR_ERR_INVALID_ARG, "o3dr_cloud_big_slice_counts_dev must run first, for the same slices");
    int64_t n_local = 0;
    for (int p = 0; p < n_parts; ++p) {
        if (counts[p] < 0) return fail(O3DR_ERR_INVALID_ARG, "negative slice size");
        n_local += counts[p];
    }
    if (n_local > c->place_ub) return fail(O3DR_ERR_INVALID_ARG, "slice sizes above the cloud's size");
    const int64_t own = counts[own_part], send_start = n_before + own + n_after, total = send_start + (n_local - own);
    if (send_offset) *send_offset = send_start;
    const int64_t cap_counted = c->place_ub;  // the (slice, tile) table is laid out for this many points: the move must use the same tiling
    c->place_ub = -1;                         // (the table is consumed)
    if (n_before == 0 && n_after == 0 && own == n_local) return O3DR_OK;  // nothing leaves, nothing arrives: the cloud stays as it is
    if (total >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "the exchange buffer exceeds 2^32-1 points");
    CHK(alt_reserve(c, total > 0 ? total : 1));
    // part p's records start at sum(counts[0..p)) in the plain layout; where they go instead
    int64_t* sh = (int64_t*)(c->misc_host + 2560);
    int64_t nat = 0, out_off = send_start;
    for (int p = 0; p < n_parts; ++p) {
        if (p == own_part) {
            sh[p] = n_before - nat;
        } else {
            sh[p] = out_off - nat;
            out_off += counts[p];
        }
        nat += counts[p];
    }
    HIPCHK(hipMemcpyAsync(c->misc_dev + 2560, sh, sizeof(int64_t) * (size_t)n_parts, hipMemcpyHostToDevice, c->stream));
    if (n_local > 0) {
        VoxelArgs v;
        CHK(slice_args(c, v, cap_counted));
        launch_partition_move(&c->prof, c->stream, c->ws, v, n_parts, c->cloud_alt, (const int64_t*)(c->misc_dev + 2560));
        if (hipGetLastError() != hipSuccess) return fail(O3DR_ERR_HIP, "slice placement launch failed");
        // (the pinned shift table is only written here, and every exchange ends in a stream wait - the merged slice's size -
        // before the next one can begin)
    }
    swap_clouds(c);  // the laid-out copy becomes cloud_big (its device count is stale until o3dr_cloud_big_set_size)
    c->cloud_box_valid = false;
    c->cloud_heads_valid = false;
    c->cloud_ub = total;
    c->cloud_n_exact = false;
    return O3DR_OK;
}

extern "C" int o3dr_cloud_big_set_size(o3dr_ctx* c, int64_t n_points)
{
    CTX_ENTER(c);
    if (n_points < 0 || n_points > c->cloud_cap) return fail(O3DR_ERR_INVALID_ARG, "more points than the cloud buffer holds");
    if (n_points >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "cloud_big exceeds 2^32-1 points");
    launch_set_cloud_count(c->stream, c->cc_big, (uint64_t)n_points);  // stream-ordered, like o3dr_cloud_big_adopt
    HIPCHK(hipGetLastError());
    c->cloud_box_valid = false;
    c->cloud_heads_valid = false;
    c->cloud_ub = n_points;
    c->cloud_n_exact = true;
    return O3DR_OK;
}

extern "C" int o3dr_cloud_big_raw_view(o3dr_ctx* c, void** ptr, int64_t* capacity_points)
{
    CTX_ENTER(c);
    if (!ptr || !capacity_points) return fail(O3DR_ERR_INVALID_ARG, "ptr / capacity is NULL");
    *ptr = c->cloud_big;
    *capacity_points = c->cloud_cap;
    return O3DR_OK;
}

extern "C" int o3dr_cloud_big_bbox(o3dr_ctx* c, float mn[3], float mx[3], int64_t* n_out)
{
    CTX_ENTER(c);
    if (!mn || !mx) return fail(O3DR_ERR_INVALID_ARG, "bounding box is NULL");
    CloudCounters cc;
    CHK(read_counters(c, c->cc_big, &cc));
    const int64_t n = (int64_t)cc.count;
    c->cloud_ub = n;
    if (n_out) *n_out = n;
    for (int a = 0; a < 3; ++a) mn[a] = __builtin_inff(), mx[a] = -__builtin_inff();
    if (n == 0) return O3DR_OK;
    if (n >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "cloud_big exceeds 2^32-1 points");
    if (c->cloud_box_valid) {
        HIPCHK(hipMemcpyAsync(c->misc_host, c->cloud_box, 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    } else {
        CHK(ws_ensure(c, 1, n, false));
        launch_set_counts(&c->prof, c->stream, c->ws.n_valid, (uint32_t)n, 1);
        const int used = launch_points_minmax(&c->prof, c->stream, c->cloud_big, 0, c->ws.n_valid, 1, n, c->ws.mm_stride, c->ws.mm);
        launch_bbox(&c->prof, c->stream, c->ws.mm, used, (float*)c->misc_dev);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->misc_host, c->misc_dev, 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    const float* h = (const float*)c->misc_host;
    for (int a = 0; a < 3; ++a) mn[a] = h[a], mx[a] = h[3 + a];
    return O3DR_OK;
}

extern "C" int o3dr_cloud_big_partition(o3dr_ctx* c, const float gmin[3], const float gmax[3], int32_t n_parts,
                                        int64_t* counts, uint32_t* status)
{
    if (status) *status = 0;
    CTX_ENTER(c);
    if (!gmin || !gmax || !counts || n_parts < 1 || n_parts > kMaxRadix)
        return fail(O3DR_ERR_INVALID_ARG, "bad arguments (1 <= n_parts <= 128)");
    for (int p = 0; p < n_parts; ++p) counts[p] = 0;
    CloudCounters cc;
    CHK(read_counters(c, c->cc_big, &cc));
    const int64_t n = (int64_t)cc.count;
    c->cloud_ub = n;
    if (n == 0) return O3DR_OK;
    if (n >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "cloud_big exceeds 2^32-1 points");
    CHK(ws_ensure(c, 1, n, false));
    CHK(alt_reserve(c, n));
    o3dr_point* nb = c->cloud_alt;
    launch_set_counts(&c->prof, c->stream, c->ws.n_valid, (uint32_t)n, 1);
    CHK(put_bbox(c, gmin, gmax));
    float leaf[3], zo;
    uint32_t mp;
    downsample_leaf(c->params, 1, leaf, &mp, &zo);
    VoxelArgs v;
    memset(&v, 0, sizeof v);
    v.in = c->cloud_big;
    v.n_dev = c->ws.n_valid;
    v.frames = 1;
    v.cap = n;
    v.leaf[0] = leaf[0];
    v.leaf[1] = leaf[1];
    v.leaf[2] = leaf[2];
    v.z_offset = zo;
    uint32_t* ovf_dev = (uint32_t*)(c->misc_dev + 32);
    uint64_t* cnt_dev = (uint64_t*)(c->misc_dev + 64);
    launch_partition(&c->prof, c->stream, c->ws, v, n_parts, nb, cnt_dev, ovf_dev);
    if (hipGetLastError() != hipSuccess) return fail(O3DR_ERR_HIP, "partition launch failed");
    HIPCHK(hipMemcpyAsync(c->misc_host, c->misc_dev, 64 + 8 * (size_t)n_parts, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    swap_clouds(c);  // the partitioned copy becomes cloud_big; the old buffer is kept as the alternate
    c->cloud_heads_valid = false;
    const uint32_t ovf = *(const uint32_t*)(c->misc_host + 32);
    const uint64_t* hc = (const uint64_t*)(c->misc_host + 64);
    for (int p = 0; p < n_parts; ++p) counts[p] = (int64_t)hc[p];
    if (status) *status = ovf ? O3DR_STATUS_VOXEL_OVERFLOW : 0u;
    return O3DR_OK;
}

// -------------------------------------------------------------------------------------------------
// The whole multi-GPU exchange behind one entry point, for C++ hosts that own RCCL communicators (one host thread and
// one context per GPU; the reference's merge sits in its C++ main flow, pose.cpp:527-532).  RCCL is resolved with
// dlopen/dlsym at first use: libo3dr.so itself carries no dependency on it.
// -------------------------------------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
};
static RcclApi* rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
        bool all = true;
        auto sym = [&](const char* n) {
            void* p = dlsym(api.handle, n);
            all = all && p != nullptr;
            return p;
        };
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
        api.CommUserRank = (decltype(api.CommUserRank))sym("ncclCommUserRank");
        api.ok = all;
    });
    return api.ok ? &api : nullptr;
}
#define NCCLCHK(R, expr)                                                                              \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess) {                                                                      \
            char buf_[512];                                                                           \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(r_), __FILE__, __LINE__); \
            g_err = buf_;                                                                             \
            return O3DR_ERR_HIP;                                                                      \
        }                                                                                             \
    } while (0)

extern "C" int o3dr_comm_init_all(int32_t n_devices, const int32_t* devices, void** comms_out)
{
    if (n_devices < 1 || n_devices > kMaxRadix || !comms_out) return fail(O3DR_ERR_INVALID_ARG, "bad arguments (1 <= n_devices <= 128)");
    RcclApi* R = rccl_api();
    if (!R) return fail(O3DR_ERR_HIP, "RCCL (librccl.so.1) could not be loaded");
    std::vector<int> devs((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i) devs[(size_t)i] = devices ? devices[i] : i;
    std::vector<ncclComm_t> comms((size_t)n_devices);
    NCCLCHK(R, R->CommInitAll(comms.data(), n_devices, devs.data()));
    for (int i = 0; i < n_devices; ++i) comms_out[i] = comms[(size_t)i];
    return O3DR_OK;
}
extern "C" int o3dr_comm_destroy(void* comm)
{
    if (!comm) return O3DR_OK;
    RcclApi* R = rccl_api();
    if (!R) return fail(O3DR_ERR_HIP, "RCCL (librccl.so.1) could not be loaded");
    NCCLCHK(R, R->CommDestroy((ncclComm_t)comm));
    return O3DR_OK;
}

// ---- transports of the exchange -----------------------------------------------------------------------------------------
// The protocol below only needs "all-gather a few bytes" and "move my slice segments to their owners"; RCCL provides
// both for real ranks.  The LOCAL transport (include/o3dr_testing.h) connects W contexts of ONE process - one host thread
// each, all on the same device - through device-to-device copies and a host barrier, so that the very code that runs over
// RCCL (sizes, slices, failure agreement, statistics) is exercised with W > 1 on a one-GPU box, where RCCL refuses two
// ranks on one device.
struct Transport {
    int W = 1, rank = 0;
    virtual ~Transport() {}
    // every rank contributes `bytes` at send_dev; recv_dev receives W * bytes in rank order (ordered on c->stream)
    virtual int all_gather(o3dr_ctx* c, const void* send_dev, void* recv_dev, size_t bytes) = 0;
    // send[p] points at base + send_off[p] go to rank p; recv[p] points from rank p land at base + recv_off[p] (same buffer,
    // disjoint regions).  Nothing is sent to the rank itself: its own slice is already where it belongs.
    virtual int all_to_all(o3dr_ctx* c, o3dr_point* base, const int64_t* send_off, const int64_t* send, const int64_t* recv_off,
                           const int64_t* recv) = 0;
};

struct RcclTransport : Transport {
    RcclApi* R = nullptr;
    ncclComm_t comm = nullptr;
    int all_gather(o3dr_ctx* c, const void* send_dev, void* recv_dev, size_t bytes) override
    {
        NCCLCHK(R, R->AllGather(send_dev, recv_dev, bytes, ncclUint8, comm, c->stream));
        return O3DR_OK;
    }
    int all_to_all(o3dr_ctx* c, o3dr_point* base, const int64_t* send_off, const int64_t* send, const int64_t* recv_off,
                   const int64_t* recv) override
    {
        // every peer pair has its own xGMI link.  An error inside the group must not leave it open: ncclGroupEnd is
        // always reached, the first error is reported after it.
        NCCLCHK(R, R->GroupStart());
        ncclResult_t first = ncclSuccess;
        for (int p = 0; p < W; ++p) {
            if (p == rank) continue;
            if (send[p] && first == ncclSuccess) first = R->Send(base + send_off[p], (size_t)send[p] * sizeof(o3dr_point), ncclUint8, p, comm, c->stream);
            if (recv[p] && first == ncclSuccess) first = R->Recv(base + recv_off[p], (size_t)recv[p] * sizeof(o3dr_point), ncclUint8, p, comm, c->stream);
        }
        const ncclResult_t endr = R->GroupEnd();
        NCCLCHK(R, first);
        NCCLCHK(R, endr);
        return O3DR_OK;
    }
};

struct LocalComm {  // test transport: shared by the W rank threads
    int W = 1;
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    bool broken = false;
    std::vector<const void*> ptr;
    std::vector<const int64_t*> cnt, off;
    // false: a rank did not show up within 30 s (it left the protocol: exactly what the tests look for)
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const uint64_t g = generation;
        if (++waiting == W) {
            waiting = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(lk, std::chrono::seconds(30), [&] { return generation != g || broken; })) {
            broken = true;
            cv.notify_all();
            return false;
        }
        return !broken;
    }
};
struct LocalTransport : Transport {
    LocalComm* L = nullptr;
    int all_gather(o3dr_ctx* c, const void* send_dev, void* recv_dev, size_t bytes) override
    {
        HIPCHK(hipStreamSynchronize(c->stream));
        L->ptr[(size_t)rank] = send_dev;
        if (!L->barrier()) return fail(O3DR_ERR_PEER, "local transport: a rank left the exchange (all-gather)");
        for (int r = 0; r < W; ++r)
            HIPCHK(hipMemcpyAsync((char*)recv_dev + (size_t)r * bytes, L->ptr[(size_t)r], bytes, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!L->barrier()) return fail(O3DR_ERR_PEER, "local transport: a rank left the exchange (all-gather)");
        return O3DR_OK;
    }
    int all_to_all(o3dr_ctx* c, o3dr_point* base, const int64_t* send_off, const int64_t* send, const int64_t* recv_off,
                   const int64_t* recv) override
    {
        HIPCHK(hipStreamSynchronize(c->stream));
        L->ptr[(size_t)rank] = base;
        L->cnt[(size_t)rank] = send;
        L->off[(size_t)rank] = send_off;
        if (!L->barrier()) return fail(O3DR_ERR_PEER, "local transport: a rank left the exchange (all-to-all)");
        int rc = O3DR_OK;
        for (int p = 0; p < W && rc == O3DR_OK; ++p) {
            if (p == rank) continue;
            if (L->cnt[(size_t)p][rank] != recv[p]) rc = fail(O3DR_ERR_INTERNAL, "local transport: send and receive counts differ");
            else if (recv[p] && hipMemcpyAsync(base + recv_off[p], (const o3dr_point*)L->ptr[(size_t)p] + L->off[(size_t)p][rank],
                                               (size_t)recv[p] * sizeof(o3dr_point), hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
                rc = fail(O3DR_ERR_HIP, "local transport: copy failed");
        }
        if (hipStreamSynchronize(c->stream) != hipSuccess && rc == O3DR_OK) rc = fail(O3DR_ERR_HIP, "local transport: sync failed");
        if (!L->barrier()) return fail(O3DR_ERR_PEER, "local transport: a rank left the exchange (all-to-all)");
        return rc;
    }
};

// What one rank tells the others with the slice counts (all-gather #2): W counts, then
//   [W]     status word of the partition (bit 0: PCL's overflow guard on the global box)
//   [W + 1] this rank's error so far (0, or a negative O3DR_ERR_* code: every rank then leaves the exchange together)
//   [W + 2] points its receive buffer (the alternate cloud) holds now
//   [W + 3] points its merged-slice buffer holds now
//   [W + 4] points its gather buffer holds now
// With the capacities every rank knows whether ANY rank has to grow a buffer before the all-to-all.  If none has to
// (every call after the first of a run), nothing can fail locally between the count matrix and the merge, and the
// exchange goes on without another word; else the ranks that must allocate do so and one more 8-byte all-gather
// carries the outcome, so that a failed allocation stops every rank before the all-to-all instead of leaving the
// others inside it.
static constexpr int kRowExtra = 5;

static int peer_error(int own, int first_rank, int64_t first_code, const std::string& own_msg)
{
    char buf[256];
    if (own != O3DR_OK) {
        snprintf(buf, sizeof buf, "%s [every rank left the exchange]", own_msg.c_str());
        g_err = buf;
        return own;
    }
    snprintf(buf, sizeof buf, "rank %d failed with code %lld: every rank left the exchange together", first_rank, (long long)first_code);
    g_err = buf;
    return O3DR_ERR_PEER;
}

static int merge_partitioned_impl(o3dr_ctx* c, Transport& T, int32_t gather_result, o3dr_point* out, int64_t out_capacity,
                                  int64_t* n_out, int64_t* n_total, uint32_t* status, int32_t mem)
{
    const int W = T.W, rank = T.rank;
    if (W < 1 || W > kMaxRadix || rank < 0 || rank >= W) return fail(O3DR_ERR_INVALID_ARG, "communicators of 1..128 ranks are supported");
    const size_t RW = (size_t)W + kRowExtra;
    // device scratch: own header | all headers | own row | row matrix (read back in one copy from o_hdrs on).  The one
    // allocation in front of the first collective (a few KiB); everything after it is decided by all ranks together.
    const size_t o_hdr = 0, o_hdrs = 32, o_row = o_hdrs + 32 * (size_t)W, o_mat = o_row + 8 * RW;
    const size_t total_bytes = o_mat + 8 * (size_t)W * RW;
    CHK(dev_ensure(c, c->st_xchg, total_bytes));
    if (c->xchg_host_cap < total_bytes) {
        if (c->xchg_host) (void)hipHostFree(c->xchg_host);
        c->xchg_host = nullptr;
        c->xchg_host_cap = 0;
        if (hipHostMalloc((void**)&c->xchg_host, total_bytes, hipHostMallocDefault) != hipSuccess) return fail(O3DR_ERR_ALLOC, "hipHostMalloc failed");
        c->xchg_host_cap = total_bytes;
    }
    memset(c->xchg_stats, 0, sizeof c->xchg_stats);
    char* d = (char*)c->st_xchg.p;
    struct Hdr {
        float mn[3], mx[3];
        int64_t count;  // negative: this rank's O3DR_ERR_* code
    };
    int local = O3DR_OK;  // this rank's first failure; it keeps taking part in the collectives that remain
    std::string local_msg;
    auto note = [&](int rc) {
        if (rc != O3DR_OK && local == O3DR_OK) {
            local = rc;
            local_msg = g_err;
        }
    };
    auto injected = [&](int point) {  // o3dr_test_fail_at
        if (c->test_fail_at != point) return false;
        c->test_fail_at = 0;
        (void)fail(O3DR_ERR_ALLOC, "failure injected by o3dr_test_fail_at");
        return true;
    };
    // 1. headers
    note(injected(1) ? O3DR_ERR_ALLOC : o3dr_cloud_big_header_dev(c, d + o_hdr));
    if (local != O3DR_OK) {
        Hdr* eh = (Hdr*)c->xchg_host;
        for (int a = 0; a < 3; ++a) eh->mn[a] = __builtin_inff(), eh->mx[a] = -__builtin_inff();
        eh->count = (int64_t)local;
        HIPCHK(hipMemcpyAsync(d + o_hdr, eh, 32, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));  // (the pinned word is reused below)
    }
    CHK(T.all_gather(c, d + o_hdr, d + o_hdrs, 32));
    // 2. slice sizes over the box the headers span (nothing moves yet: the slices are placed once the counts are known)
    if (local == O3DR_OK) note(injected(2) ? O3DR_ERR_ALLOC : o3dr_cloud_big_slice_counts_dev(c, d + o_hdrs, W, W, (int64_t*)(d + o_row)));
    if (local != O3DR_OK) HIPCHK(hipMemsetAsync(d + o_row, 0, 8 * ((size_t)W + 1), c->stream));
    int64_t* extra = (int64_t*)c->xchg_host;
    extra[0] = (int64_t)local;
    extra[1] = c->cloud_alt_cap;
    extra[2] = (int64_t)(c->st_merge.cap / sizeof(o3dr_point));
    extra[3] = (int64_t)(c->st_gather.cap / sizeof(o3dr_point));
    HIPCHK(hipMemcpyAsync(d + o_row + 8 * ((size_t)W + 1), extra, 8 * (kRowExtra - 1), hipMemcpyHostToDevice, c->stream));
    // 3. row matrix and the ONE read-back
    CHK(T.all_gather(c, d + o_row, d + o_mat, 8 * RW));
    HIPCHK(hipMemcpyAsync(c->xchg_host + o_hdrs, d + o_hdrs, total_bytes - o_hdrs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const Hdr* hdrs = (const Hdr*)(c->xchg_host + o_hdrs);
    const int64_t* mat = (const int64_t*)(c->xchg_host + o_mat);
    for (int r = 0; r < W; ++r) {  // a failure anywhere so far: every rank sees it here and leaves before the all-to-all
        const int64_t code = hdrs[r].count < 0 ? hdrs[r].count : mat[(size_t)r * RW + W + 1];
        if (code < 0) return peer_error(local, r, code, local_msg);
    }
    float gmin[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, gmax[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    int64_t total_pts = 0;
    for (int r = 0; r < W; ++r) {
        if (hdrs[r].count <= 0) continue;
        total_pts += hdrs[r].count;
        for (int a = 0; a < 3; ++a) {
            gmin[a] = std::min(gmin[a], hdrs[r].mn[a]);
            gmax[a] = std::max(gmax[a], hdrs[r].mx[a]);
        }
    }
    if (n_total) *n_total = total_pts;
    const int64_t n_local = hdrs[rank].count;
    bool overflow = false;
    for (int r = 0; r < W; ++r) overflow = overflow || ((mat[(size_t)r * RW + W] & 1) != 0);  // (same global box everywhere: all agree)
    // what every rank sends, receives and merges (the same arithmetic on the same matrix everywhere)
    auto sends = [&](int from, int to) { return overflow ? (from == to ? hdrs[from].count : (int64_t)0) : mat[(size_t)from * RW + to]; };
    std::vector<int64_t> send((size_t)W, 0), recv((size_t)W, 0), slice_in((size_t)W, 0);
    for (int p = 0; p < W; ++p) {
        send[(size_t)p] = sends(rank, p);
        recv[(size_t)p] = sends(p, rank);
        for (int r = 0; r < W; ++r) slice_in[(size_t)p] += sends(r, p);
    }
    int64_t n_recv = 0, max_slice = 0;
    for (int p = 0; p < W; ++p) n_recv += recv[(size_t)p], max_slice = std::max(max_slice, slice_in[(size_t)p]);
    if (total_pts == 0) return O3DR_OK;
    if (max_slice >= (int64_t)0xffffffffLL) return fail(O3DR_ERR_INVALID_ARG, "a slice exceeds 2^32-1 points");  // (every rank: same matrix)
    // an upper bound of any rank's merged slice, for the gather buffer: no more cells than points in the slice, and a
    // slice covers at most ~cells / W linear indices of the combined grid over the global box (part_of: multiply-shift)
    int64_t pad_bound = max_slice;
    if (!overflow) {
        float leaf[3], zo;
        uint32_t mp;
        downsample_leaf(c->params, 1, leaf, &mp, &zo);
        double cells = 1.0;
        const double lo[3] = {gmin[0], gmin[1], (double)gmin[2] + zo}, hi[3] = {gmax[0], gmax[1], (double)gmax[2] + zo};
        for (int a = 0; a < 3; ++a) cells *= std::floor((hi[a] - lo[a]) / (double)leaf[a]) + 3.0;
        const double per_slice = 2.0 * cells / W + 4.0;
        if (per_slice < (double)pad_bound) pad_bound = (int64_t)per_slice;
    }
    c->xchg_stats[0] = n_local;
    c->xchg_stats[1] = n_local - send[(size_t)rank];                                   // points sent to other ranks
    c->xchg_stats[2] = n_recv - recv[(size_t)rank];                                    // points received from other ranks
    c->xchg_stats[3] = c->xchg_stats[1] * (int64_t)sizeof(o3dr_point);                 // bytes sent over the links
    c->xchg_stats[4] = c->xchg_stats[2] * (int64_t)sizeof(o3dr_point);
    c->xchg_stats[5] = n_recv;                                                         // points entering this rank's merge
    c->xchg_stats[7] = total_pts;
    // Where this rank's points go: its own slice stays with it, placed once with room in front for what the lower ranks
    // send and behind for the higher ranks'; the slices that leave follow.  A rank that neither sends nor receives
    // anything keeps its cloud as it is (its recorded box and run heads stay valid).
    auto placed = [&](int r, int64_t& off_rank, int64_t& nr) {  // points rank r's exchange buffer must hold (0: nothing moves there)
        nr = 0;
        for (int p = 0; p < W; ++p) nr += sends(p, r);
        off_rank = hdrs[r].count - sends(r, r);
        return (off_rank == 0 && nr == sends(r, r)) ? (int64_t)0 : nr + off_rank;
    };
    int64_t n_off = 0, nr_me = 0;
    const int64_t need_alt = overflow ? 0 : placed(rank, n_off, nr_me);
    const bool moves = need_alt != 0;
    // does any rank have to grow a buffer before the all-to-all?  (every rank evaluates every rank: no disagreement)
    bool any_grows = false;
    for (int r = 0; r < W; ++r) {
        int64_t o_r = 0, n_r = 0;
        const int64_t need_r = overflow ? 0 : placed(r, o_r, n_r);
        const int64_t* ex = mat + (size_t)r * RW + W + 2;
        any_grows = any_grows || need_r > ex[0] || max_slice > ex[1] || (gather_result && (int64_t)W * pad_bound > ex[2]);
    }
    if (any_grows) {
        int rc = O3DR_OK;
        if (injected(3)) rc = O3DR_ERR_ALLOC;
        if (rc == O3DR_OK && need_alt > 0) rc = alt_reserve(c, need_alt);
        if (rc == O3DR_OK) rc = dev_ensure(c, c->st_merge, (size_t)(max_slice > 0 ? max_slice : 1) * sizeof(o3dr_point));
        if (rc == O3DR_OK && gather_result) rc = dev_ensure(c, c->st_gather, (size_t)W * (size_t)(pad_bound > 0 ? pad_bound : 1) * sizeof(o3dr_point));
        note(rc);
        extra[0] = (int64_t)local;
        HIPCHK(hipMemcpyAsync(d + o_row, extra, 8, hipMemcpyHostToDevice, c->stream));
        CHK(T.all_gather(c, d + o_row, d + o_mat, 8));
        HIPCHK(hipMemcpyAsync(c->xchg_host + o_mat, d + o_mat, 8 * (size_t)W, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->xchg_stats[6] = 1;  // agreement rounds of this call
        for (int r = 0; r < W; ++r)
            if (mat[r] < 0) return peer_error(local, r, mat[r], local_msg);
    }
    c->cloud_ub = n_local;  // (its own header told the host)
    c->cloud_n_exact = true;
    if (!overflow) {
        // one pass places the slices; the all-to-all then receives straight into the gaps (every peer pair has its own
        // xGMI link; segments land in source-rank order = global frame order); the own slice is never sent
        int64_t n_before = 0, n_after = 0;
        for (int p = 0; p < W; ++p) (p < rank ? n_before : n_after) += p == rank ? 0 : recv[(size_t)p];
        std::vector<int64_t> send_off((size_t)W, 0), recv_off((size_t)W, 0), send_x(send), recv_x(recv);
        send_x[(size_t)rank] = recv_x[(size_t)rank] = 0;
        if (moves) {
            int64_t send_start = 0;
            CHK(o3dr_cloud_big_place_slices(c, W, rank, send.data(), n_before, n_after, &send_start));
            int64_t so = send_start, lo = 0, hi = n_before + send[(size_t)rank];
            for (int p = 0; p < W; ++p) {
                if (p == rank) continue;
                send_off[(size_t)p] = so;
                so += send[(size_t)p];
                int64_t& ro = p < rank ? lo : hi;
                recv_off[(size_t)p] = ro;
                ro += recv[(size_t)p];
            }
        }
        CHK(T.all_to_all(c, c->cloud_big, send_off.data(), send_x.data(), recv_off.data(), recv_x.data()));
        if (moves) CHK(o3dr_cloud_big_set_size(c, n_recv));
    }
    // 4. local merge of the slice over the global box (the second host wait: its size).  A failure here travels with
    //    the merged sizes of the final gather, so that no rank waits in a collective the failed one never enters.
    int64_t m = 0;
    uint32_t st = 0;
    if (n_recv > 0) note(injected(4) ? O3DR_ERR_ALLOC : finalize_impl(c, gmin, gmax, (o3dr_point*)c->st_merge.p, max_slice, &m, &st, O3DR_MEM_DEVICE));
    if (overflow) st |= O3DR_STATUS_VOXEL_OVERFLOW;
    if (status) *status = st;
    const hipMemcpyKind kind = mem == O3DR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (!gather_result) {  // (no collective follows: a local failure is this rank's alone)
        if (local != O3DR_OK) {
            g_err = local_msg;
            return local;
        }
        if (out_capacity > 0) {
            if (m > out_capacity) return fail(O3DR_ERR_CAPACITY, "output buffer too small");
            if (m > 0) HIPCHK(hipMemcpyAsync(out, c->st_merge.p, (size_t)m * sizeof(o3dr_point), kind, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            *n_out = m;
        }
        return O3DR_OK;
    }
    // 5. final gather: sizes (or error codes), then the slices padded to the largest (rank order = ascending voxel index)
    int64_t* mh = (int64_t*)c->xchg_host;
    mh[0] = local != O3DR_OK ? (int64_t)local : m;
    HIPCHK(hipMemcpyAsync(d + o_row, mh, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    CHK(T.all_gather(c, d + o_row, d + o_mat, 8));
    HIPCHK(hipMemcpyAsync(c->xchg_host + o_mat, d + o_mat, 8 * (size_t)W, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int64_t pad = 0, merged_total = 0;
    for (int r = 0; r < W; ++r) {
        if (mat[r] < 0) return peer_error(local, r, mat[r], local_msg);
        pad = std::max(pad, mat[r]), merged_total += mat[r];
    }
    if (pad > pad_bound) return fail(O3DR_ERR_INTERNAL, "a merged slice exceeds its bound");  // (every rank: same sizes, same bound)
    if (pad > 0) CHK(T.all_gather(c, c->st_merge.p, c->st_gather.p, (size_t)pad * sizeof(o3dr_point)));
    if (out_capacity > 0) {  // (a rank that does not want the result passes no buffer; it still took part in the collectives)
        if (merged_total > out_capacity) {
            HIPCHK(hipStreamSynchronize(c->stream));
            return fail(O3DR_ERR_CAPACITY, "output buffer too small");
        }
        int64_t off = 0;
        for (int r = 0; r < W; ++r) {
            if (mat[r] > 0)
                HIPCHK(hipMemcpyAsync(out + off, (const o3dr_point*)c->st_gather.p + (size_t)r * (size_t)pad, (size_t)mat[r] * sizeof(o3dr_point), kind, c->stream));
            off += mat[r];
        }
        *n_out = merged_total;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return O3DR_OK;
}

extern "C" int o3dr_merge_partitioned(o3dr_ctx* c, void* nccl_comm, int32_t gather_result, o3dr_point* out, int64_t out_capacity,
                                      int64_t* n_out, int64_t* n_total, uint32_t* status, int32_t mem)
{
    if (n_out) *n_out = 0;
    if (n_total) *n_total = 0;
    if (status) *status = 0;
    CTX_ENTER(c);
    if (!nccl_comm || !n_out) return fail(O3DR_ERR_INVALID_ARG, "nccl_comm / n_out is NULL");
    if (mem != O3DR_MEM_HOST && mem != O3DR_MEM_DEVICE) return fail(O3DR_ERR_INVALID_ARG, "bad mem kind");
    if (out_capacity < 0 || (out_capacity > 0 && !out)) return fail(O3DR_ERR_INVALID_ARG, "out is NULL");
    RcclTransport T;
    T.R = rccl_api();
    if (!T.R) return fail(O3DR_ERR_HIP, "RCCL (librccl.so.1) could not be loaded");
    T.comm = (ncclComm_t)nccl_comm;
    NCCLCHK(T.R, T.R->CommCount(T.comm, &T.W));
    NCCLCHK(T.R, T.R->CommUserRank(T.comm, &T.rank));
    return merge_partitioned_impl(c, T, gather_result, out, out_capacity, n_out, n_total, status, mem);
}

extern "C" int o3dr_merge_partitioned_stats(o3dr_ctx* c, int64_t out[8])
{
    CTX_ENTER(c);
    if (!out) return fail(O3DR_ERR_INVALID_ARG, "out is NULL");
    memcpy(out, c->xchg_stats, sizeof c->xchg_stats);
    return O3DR_OK;
}

extern "C" int o3dr_cloud_big_capacity(o3dr_ctx* c, int64_t* cloud_points, int64_t* recv_points)
{
    CTX_ENTER(c);
    if (cloud_points) *cloud_points = c->cloud_cap;
    if (recv_points) *recv_points = c->cloud_alt_cap;
    return O3DR_OK;
}

// ---- test-only: the exchange among W contexts of one process (include/o3dr_testing.h) -----------------------------------
extern "C" int o3dr_test_local_comm_create(int32_t n_ranks, void** comm_out)
{
    if (n_ranks < 1 || n_ranks > kMaxRadix || !comm_out) return fail(O3DR_ERR_INVALID_ARG, "bad arguments (1 <= n_ranks <= 128)");
    LocalComm* L = new LocalComm;
    L->W = n_ranks;
    L->ptr.assign((size_t)n_ranks, nullptr);
    L->cnt.assign((size_t)n_ranks, nullptr);
    L->off.assign((size_t)n_ranks, nullptr);
    *comm_out = L;
    return O3DR_OK;
}
extern "C" int o3dr_test_local_comm_destroy(void* comm)
{
    delete (LocalComm*)comm;
    return O3DR_OK;
}
extern "C" int o3dr_test_merge_partitioned_local(o3dr_ctx* c, void* local_comm, int32_t rank, int32_t gather_result, o3dr_point* out,
                                                 int64_t out_capacity, int64_t* n_out, int64_t* n_total, uint32_t* status, int32_t mem)
{
    if (n_out) *n_out = 0;
    if (n_total) *n_total = 0;
    if (status) *status = 0;
    CTX_ENTER(c);
    if (!c->test_hooks) return fail(O3DR_ERR_INVALID_ARG, "test hooks are off (create the context with O3DR_TEST_HOOKS=1)");
    if (!local_comm || !n_out) return fail(O3DR_ERR_INVALID_ARG, "local_comm / n_out is NULL");
    if (mem != O3DR_MEM_HOST && mem != O3DR_MEM_DEVICE) return fail(O3DR_ERR_INVALID_ARG, "bad mem kind");
    if (out_capacity < 0 || (out_capacity > 0 && !out)) return fail(O3DR_ERR_INVALID_ARG, "out is NULL");
    LocalTransport T;
    T.L = (LocalComm*)local_comm;
    T.W = T.L->W;
    T.rank = rank;
    return merge_partitioned_impl(c, T, gather_result, out, out_capacity, n_out, n_total, status, mem);
}
extern "C" int o3dr_test_fail_at(o3dr_ctx* c, int32_t point)
{
    CTX_ENTER(c);
    if (!c->test_hooks) return fail(O3DR_ERR_INVALID_ARG, "test hooks are off (create the context with O3DR_TEST_HOOKS=1)");
    c->test_fail_at = point;
    return O3DR_OK;
}

// page-locking of caller memory (frame stacks handed to o3dr_accumulate_frames with O3DR_MEM_HOST then move by DMA)
extern "C" int o3dr_host_register(void* ptr, int64_t bytes)
{
    if (!ptr || bytes <= 0) return fail(O3DR_ERR_INVALID_ARG, "bad arguments");
    // (callers treat this as best effort: HIP's sticky last error must not outlive the failure, or the next
    // hipGetLastError() after a kernel launch would report it as that launch's)
    const hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        char buf[256];
        snprintf(buf, sizeof buf, "hipHostRegister failed: %s", hipGetErrorString(e));
        return fail(O3DR_ERR_HIP, buf);
    }
    return O3DR_OK;
}
extern "C" int o3dr_host_unregister(void* ptr)
{
    if (!ptr) return O3DR_OK;
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        char buf[256];
        snprintf(buf, sizeof buf, "hipHostUnregister failed: %s", hipGetErrorString(e));
        return fail(O3DR_ERR_HIP, buf);
    }
    return O3DR_OK;
}

// -------------------------------------------------------------------------------------------------
// measurement hooks
// -------------------------------------------------------------------------------------------------
extern "C" int o3dr_bilateral_filter_u8(o3dr_ctx* c, const uint8_t* src, int64_t src_pitch, int32_t rows, int32_t cols, int32_t d,
                                        double sigma_color, double sigma_space, uint8_t* dst, int64_t dst_pitch, int32_t mem)
{
    CTX_ENTER(c);
    if (mem != O3DR_MEM_HOST && mem != O3DR_MEM_DEVICE) return fail(O3DR_ERR_INVALID_ARG, "bad mem kind");
    if (rows < 0 || cols < 0) return fail(O3DR_ERR_INVALID_ARG, "negative image size");
    if (rows == 0 || cols == 0) return O3DR_OK;
    if (!src || !dst) return fail(O3DR_ERR_INVALID_ARG, "src / dst is NULL");
    if (src_pitch < cols || dst_pitch < cols) return fail(O3DR_ERR_INVALID_ARG, "pitch smaller than a row");
    CHK(bilateral_prepare(c, d, sigma_color, sigma_space));
    const uint8_t* src_d = src;
    uint8_t* dst_d = dst;
    if (mem == O3DR_MEM_HOST) {
        const void* p;
        CHK(stage_in(c, c->st_blur_in, src, (size_t)src_pitch * rows, mem, &p));
        src_d = (const uint8_t*)p;
        CHK(dev_ensure(c, c->st_blur, (size_t)dst_pitch * rows));
        dst_d = (uint8_t*)c->st_blur.p;
    }
    launch_bilateral(&c->prof, c->stream, src_d, src_pitch, 0, rows, cols, 1, c->bil_radius, c->bil_maxk,
                     (const float*)c->bil_tab.p, dst_d, dst_pitch, 0);
    HIPCHK(hipGetLastError());
    if (mem == O3DR_MEM_HOST) {
        // only the pixels are written back: padding bytes of the caller's rows stay untouched
        HIPCHK(hipMemcpy2DAsync(dst, (size_t)dst_pitch, dst_d, (size_t)dst_pitch, (size_t)cols, (size_t)rows,
                                hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return O3DR_OK;
}

extern "C" int o3dr_disparity_variance(o3dr_ctx* c, const uint8_t* disp, int64_t disp_pitch, int64_t disp_frame_stride,
                                       int32_t rows, int32_t cols, int32_t n_frames, double* variance_out, int32_t mem)
{
    CTX_ENTER(c);
    if (mem != O3DR_MEM_HOST && mem != O3DR_MEM_DEVICE) return fail(O3DR_ERR_INVALID_ARG, "bad mem kind");
    if (n_frames < 0 || rows <= 0 || cols <= 0) return fail(O3DR_ERR_INVALID_ARG, "bad image size / frame count");
    if (n_frames == 0) return O3DR_OK;
    if (!disp || !variance_out) return fail(O3DR_ERR_INVALID_ARG, "disp / variance_out is NULL");
    if (disp_pitch < cols) return fail(O3DR_ERR_INVALID_ARG, "pitch smaller than a row");
    if (c->params.disparity_f64) return fail(O3DR_ERR_INVALID_ARG, "o3dr_disparity_variance takes CV_8UC1 images");
    if (n_frames > 1 && disp_frame_stride < (int64_t)rows * disp_pitch)
        return fail(O3DR_ERR_INVALID_ARG, "frame stride smaller than a frame");
    const GridShape g = grid_shape(c->params, rows, cols);
    const void* disp_d;
    CHK(stage_in(c, c->st_blur_in, disp, (size_t)disp_frame_stride * (n_frames - 1) + (size_t)disp_pitch * rows, mem, &disp_d));
    const size_t hist_bytes = sizeof(unsigned long long) * 256 * (size_t)n_frames;
    CHK(dev_ensure(c, c->st_hist, hist_bytes + sizeof(double) * (size_t)n_frames));
    unsigned long long* hist = (unsigned long long*)c->st_hist.p;
    double* var_d = (double*)((char*)c->st_hist.p + hist_bytes);
    launch_disp_variance(&c->prof, c->stream, (const uint8_t*)disp_d, disp_pitch, disp_frame_stride, rows, cols, n_frames,
                         c->params.bounding_box, g.cs, c->params.min_disparity, hist, var_d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(variance_out, var_d, sizeof(double) * (size_t)n_frames, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return O3DR_OK;
}

extern "C" int o3dr_test_corrupt_next_gather(o3dr_ctx* c)
{
    CTX_ENTER(c);
    if (!c->test_hooks) return fail(O3DR_ERR_INVALID_ARG, "test hooks are off (create the context with O3DR_TEST_HOOKS=1)");
    c->test_corrupt = 1;
    return O3DR_OK;
}

extern "C" int o3dr_test_sor_distances(o3dr_ctx* c, float* out, int64_t n)
{
    CTX_ENTER(c);
    if (!c->test_hooks) return fail(O3DR_ERR_INVALID_ARG, "test hooks are off (create the context with O3DR_TEST_HOOKS=1)");
    if (!out || n < 0 || n > c->ws_sor_cap || !c->ws.sor_dist) return fail(O3DR_ERR_INVALID_ARG, "no outlier removal of that size has run");
    if (n == 0) return O3DR_OK;
    HIPCHK(hipMemcpyAsync(out, c->ws.sor_dist, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return O3DR_OK;
}

extern "C" int o3dr_profile_enable(o3dr_ctx* c, int32_t kernel_id, int32_t enable)
{
    CTX_ENTER(c);
    if (kernel_id >= O3DR_K_NUM) return fail(O3DR_ERR_INVALID_ARG, "bad kernel id");
    const uint32_t bits = kernel_id < 0 ? ((1u << O3DR_K_NUM) - 1u) : (1u << kernel_id);
    if (enable)
        c->prof.mask |= bits;
    else
        c->prof.mask &= ~bits;
    return O3DR_OK;
}
extern "C" int o3dr_profile_read(o3dr_ctx* c, int32_t kernel_id, double* total_ms, int64_t* launches)
{
    CTX_ENTER(c);
    if (kernel_id < 0 || kernel_id >= O3DR_K_NUM) return fail(O3DR_ERR_INVALID_ARG, "bad kernel id");
    HIPCHK(hipStreamSynchronize(c->stream));
    c->prof.drain();
    if (total_ms) *total_ms = c->prof.total_ms[kernel_id];
    if (launches) *launches = c->prof.launches[kernel_id];
    return O3DR_OK;
}
extern "C" int o3dr_profile_reset(o3dr_ctx* c)
{
    CTX_ENTER(c);
    HIPCHK(hipMemsetAsync(c->stats_dev, 0, sizeof(SortStats), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->prof.reset();
    return O3DR_OK;
}
extern "C" int o3dr_profile_stats(o3dr_ctx* c, int64_t out[8])
{
    CTX_ENTER(c);
    if (!out) return fail(O3DR_ERR_INVALID_ARG, "out is NULL");
    HIPCHK(hipMemcpyAsync(c->stats_host, c->stats_dev, sizeof(SortStats), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    out[0] = (int64_t)c->stats_host->sort_record_passes;
    out[1] = (int64_t)c->stats_host->voxel_points_in;
    out[2] = (int64_t)c->stats_host->voxel_points_out;
    out[3] = 0;
    out[4] = (int64_t)c->stats_host->sort_records;
    out[5] = out[6] = out[7] = 0;
    return O3DR_OK;
}
extern "C" int o3dr_device_info(o3dr_ctx* c, char* name, int32_t name_len, int32_t* cu_count, int64_t* hbm_bytes)
{
    CTX_ENTER(c);
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, c->device));
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return O3DR_OK;
}
