// o3dr_kernels.hip — hand-written gfx950 kernels of the reconstruction hot path.
//
//   K1  k_reproject_count / k_reproject_emit   A1+A2: (u,v,disparity) -> Q -> SE(3) -> ordered cloud
//   K2  k_voxel_geom / k_voxel_keys            A4 steps 1-5: PCL VoxelGrid geometry and linear index
//       k_radix_hist / k_radix_scatter         A4 step 6: stable LSD radix sort of (index, point id)
//       k_run_heads / k_run_starts / k_centroid  A4 steps 7-8: runs -> ordered fp32 centroid
//
// Everything here is HBM-bound integer/byte/fp32 work: no MFMA.  The whole translation unit is
// compiled with -ffp-contract=off because the reference arithmetic (unoptimised x86 build,
// SURVEY.md section 7 "hard parts") never fuses a multiply-add, and a 1-ulp difference in a world
// coordinate flips voxel occupancy.
//
// Reference lines each kernel follows are cited at the kernel.  Wavefront size is 64.
#include "o3dr_device.h"
#include "o3dr_profile.h"

namespace o3dr {

// =================================================================================================
// small device helpers
// =================================================================================================
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float wave_min_f32(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
// exclusive scan of one value per thread over a workgroup of NW waves; returns the exclusive
// prefix, `total` = sum over the workgroup.  `lds` holds NW+1 words.
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* lds, uint32_t& total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_u32(v);
    if (lane == 63) lds[w] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const uint32_t t = lds[i];
            lds[i] = run;
            run += t;
        }
        lds[NW] = run;
    }
    __syncthreads();
    const uint32_t r = lds[w] + incl - v;
    total = lds[NW];
    __syncthreads();
    return r;
}
// the buffer (of the ping-pong pair) a frame's records end up in after its sort passes
__device__ __forceinline__ const uint32_t* sorted_buf(const VoxelGeom& g, const uint32_t* b0, const uint32_t* b1)
{
    return ((g.passes + g.buf0) & 1u) ? b1 : b0;
}
// Run-compressed sorts carry (first point, length) of a run in the 32-bit payload: the low `bits` bits hold the
// first point (< n), the rest length - 1.  Runs are cut at multiples of the largest length that fits, which
// only makes more (shorter) records of one voxel; the stable sort keeps them in order.
__device__ __forceinline__ uint32_t run_start_bits(uint32_t n) { return n > 1u ? 32u - (uint32_t)__clz(n - 1u) : 1u; }
__device__ __forceinline__ uint32_t run_split_mask(uint32_t bits) { return bits >= 32u ? 0u : (1u << (32u - bits)) - 1u; }
__device__ __forceinline__ uint32_t run_pack(uint32_t first, uint32_t next, uint32_t bits)
{
    return bits >= 32u ? first : (first | ((next - first - 1u) << bits));
}
__device__ __forceinline__ void run_unpack(uint32_t v, uint32_t bits, uint32_t& first, uint32_t& len)
{
    if (bits >= 32u) {
        first = v;
        len = 1u;
    } else {
        first = v & ((1u << bits) - 1u);
        len = (v >> bits) + 1u;
    }
}

// workgroup min/max of per-thread (lo[3], hi[3]) -> one 6-float slot (min xyz, max xyz); a workgroup
// without points stores (+inf, -inf).  Slots are reduced by k_voxel_geom: no same-address atomics.
template <int NW>
__device__ __forceinline__ void block_minmax_store(const float lo[3], const float hi[3], bool any,
                                                   float* lds /*6*NW*/, float* __restrict__ slot6)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = wave_min_f32(any ? lo[a] : __builtin_inff());
        const float h = wave_max_f32(any ? hi[a] : -__builtin_inff());
        if (lane == 0) {
            lds[w * 6 + a] = l;
            lds[w * 6 + 3 + a] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float v = lds[a];
#pragma unroll
        for (int i = 1; i < NW; ++i) v = (a < 3) ? fminf(v, lds[i * 6 + a]) : fmaxf(v, lds[i * 6 + a]);
        slot6[a] = v;
    }
    __syncthreads();
}

// =================================================================================================
// K1 — fused reprojection + rigid transform (A1 + A2)
//
//   pose_functions.cpp:1110-1121: vec = Q*(x,y,d,1) in fp64 (rows evaluated left to right),
//       vec /= vec(3) as multiplication by 1./vec(3) (+0), (float) casts, colour R<<16|G<<8|B;
//   pose_functions.cpp:1358-1362 -> pcl::transformPointCloud, dense branch:
//       x' = ((m00*x + m01*y) + m02*z) + m03 in fp32, no fused multiply-add.
// =================================================================================================
struct Pix {
    float x, y, z;
    uint32_t rgba;
};

__device__ __forceinline__ Pix reproject_one(const double* __restrict__ Q, int x, int y, double d,
                                             uint32_t b, uint32_t g, uint32_t r, const float* m, bool xf)
{
    const double v0 = (double)x, v1 = (double)y;
    const double t0 = ((Q[0] * v0 + Q[1] * v1) + Q[2] * d) + Q[3];
    const double t1 = ((Q[4] * v0 + Q[5] * v1) + Q[6] * d) + Q[7];
    const double t2 = ((Q[8] * v0 + Q[9] * v1) + Q[10] * d) + Q[11];
    const double t3 = ((Q[12] * v0 + Q[13] * v1) + Q[14] * d) + Q[15];
    const double alpha = 1. / t3;
    const float X = (float)(t0 * alpha + 0.0);
    const float Y = (float)(t1 * alpha + 0.0);
    const float Z = (float)(t2 * alpha + 0.0);
    Pix p;
    if (xf) {
        p.x = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
        p.y = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
        p.z = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
    } else {
        p.x = X;
        p.y = Y;
        p.z = Z;
    }
    p.rgba = (r << 16) | (g << 8) | b;
    return p;
}

// disparity bytes of the (up to) 4 candidates of one lane; returns the validity mask
// disparity image element: CV_8UC1, or CV_64F with --use_segment_labels (pose_functions.cpp:1037,1102)
template <bool F64> struct DispT { using type = uint32_t; };
template <> struct DispT<true> { using type = double; };

template <bool F64>
__device__ __forceinline__ uint32_t load_lane_disparities(const ReprojectArgs& a, const uint8_t* __restrict__ disp,
                                                          int c0, int n_cand, int& x0, int& y0,
                                                          typename DispT<F64>::type d[4])
{
    uint32_t valid = 0;
    if (c0 >= n_cand) return 0;
    if (!F64 && a.vec4) {  // 4 consecutive pixels of one row, 4-byte aligned
        const int ry = c0 / a.Nx, rx = c0 - ry * a.Nx;
        y0 = a.bb + ry;
        x0 = a.cs + rx;
        const uint32_t w = *reinterpret_cast<const uint32_t*>(disp + (int64_t)y0 * a.disp_pitch + x0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d[k] = (typename DispT<F64>::type)((w >> (8 * k)) & 255u);
            if ((double)d[k] > a.min_disp) valid |= 1u << k;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + k;
            d[k] = 0;
            if (c < n_cand) {
                const int ry = c / a.Nx, rx = c - ry * a.Nx;
                const uint8_t* row = disp + (int64_t)(a.bb + ry * a.jump) * a.disp_pitch;
                const int x = a.cs + rx * a.jump;
                if (F64)
                    d[k] = (typename DispT<F64>::type)reinterpret_cast<const double*>(row)[x];
                else
                    d[k] = (typename DispT<F64>::type)row[x];
                if ((double)d[k] > a.min_disp) valid |= 1u << k;
            }
        }
    }
    return valid;
}

// pass 1: valid candidates per 1024-candidate tile (reads 1 byte per candidate)
template <bool F64>
__global__ __launch_bounds__(kEmitThreads) void k_reproject_count(ReprojectArgs a, uint32_t* __restrict__ tile_cnt)
{
    __shared__ uint32_t lds[kEmitThreads / 64];
    const int f = blockIdx.y, tile = blockIdx.x;
    const uint8_t* disp = a.disp + (int64_t)f * a.disp_fstride;
    const int n_cand = a.Ny * a.Nx;
    const int c0 = tile * kEmitTile + threadIdx.x * kEmitPerLane;
    int x0 = 0, y0 = 0;
    typename DispT<F64>::type d[4];
    const uint32_t valid = load_lane_disparities<F64>(a, disp, c0, n_cand, x0, y0, d);
    const uint32_t s = wave_sum_u32(__popc(valid));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < kEmitThreads / 64; ++i) t += lds[i];
        tile_cnt[(int64_t)f * a.n_tiles + tile] = t;
    }
}

// pass 2: recompute, transform, compact in row-major order, write 16-byte points coalesced
template <bool F64>
__global__ __launch_bounds__(kEmitThreads) void k_reproject_emit(ReprojectArgs a, o3dr_point* __restrict__ out,
                                                                 const uint32_t* __restrict__ tile_off,
                                                                 const uint32_t* __restrict__ n_kp,
                                                                 float* __restrict__ mm,
                                                                 const VoxelGeom* __restrict__ gate)
{
    // batches on the pixel-window path: only frames left to the sort-based path need the ordered cloud itself
    if (gate && gate[blockIdx.y].n == 0) return;
    __shared__ uint4 stage[kEmitTile];  // 16 KiB: the tile's points in output order
    __shared__ uint32_t scan_lds[kEmitThreads / 64 + 1];
    __shared__ float mm_lds[6 * (kEmitThreads / 64)];
    __shared__ double lut_alpha[256];  // rectified-stereo Q: 1/w and Z depend on the disparity byte only
    __shared__ float lut_z[256];
    if (a.lut) {
        lut_alpha[threadIdx.x] = a.lut[threadIdx.x].alpha;
        lut_z[threadIdx.x] = a.lut[threadIdx.x].z;
        __syncthreads();
    }
    const int f = blockIdx.y, tile = blockIdx.x;
    const uint8_t* disp = a.disp + (int64_t)f * a.disp_fstride;
    const uint8_t* bgr = a.bgr + (int64_t)f * a.bgr_fstride;
    const int n_cand = a.Ny * a.Nx;
    const int c0 = tile * kEmitTile + threadIdx.x * kEmitPerLane;

    float m[12];
    const bool xf = a.xf_mode != 0;
    if (a.xf_mode == 2) {
        const float* T = a.poses + 16 * (int64_t)f;  // wave-uniform: scalar loads
#pragma unroll
        for (int i = 0; i < 12; ++i) m[i] = T[i];
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) m[i] = a.T[i];
    }

    int x0 = 0, y0 = 0;
    typename DispT<F64>::type d[4];
    const uint32_t valid = load_lane_disparities<F64>(a, disp, c0, n_cand, x0, y0, d);

    Pix p[4];
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    if (valid) {
        uint32_t cb[4], cg[4], cr[4];
        if (!F64 && a.vec4) {  // 12 colour bytes of 4 pixels = 3 aligned dwords
            const uint32_t* q = reinterpret_cast<const uint32_t*>(bgr + (int64_t)y0 * a.bgr_pitch + 3 * (int64_t)x0);
            const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
            cb[0] = w0 & 255u;         cg[0] = (w0 >> 8) & 255u;  cr[0] = (w0 >> 16) & 255u;
            cb[1] = w0 >> 24;          cg[1] = w1 & 255u;         cr[1] = (w1 >> 8) & 255u;
            cb[2] = (w1 >> 16) & 255u; cg[2] = w1 >> 24;          cr[2] = w2 & 255u;
            cb[3] = (w2 >> 8) & 255u;  cg[3] = (w2 >> 16) & 255u; cr[3] = w2 >> 24;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (valid & (1u << k)) {
                int x, y;
                if (!F64 && a.vec4) {
                    x = x0 + k;
                    y = y0;
                } else {
                    const int c = c0 + k;
                    const int ry = c / a.Nx, rx = c - ry * a.Nx;
                    y = a.bb + ry * a.jump;
                    x = a.cs + rx * a.jump;
                    const uint8_t* px = bgr + (int64_t)y * a.bgr_pitch + 3 * (int64_t)x;
                    cb[k] = px[0];
                    cg[k] = px[1];
                    cr[k] = px[2];
                }
                bool done = false;
                if constexpr (!F64) {
                  if (a.lut) {
                    done = true;
                    // same arithmetic as reproject_one with the exact-zero terms of Q dropped (adding
                    // +-0 and multiplying by the tabulated 1./w change no bit of the result)
                    const double al = lut_alpha[d[k]];
                    const float X = (float)((a.Q[0] * (double)x + a.Q[3]) * al + 0.0);
                    const float Y = (float)((a.Q[5] * (double)y + a.Q[7]) * al + 0.0);
                    const float Z = lut_z[d[k]];
                    if (xf) {
                        p[k].x = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
                        p[k].y = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
                        p[k].z = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
                    } else {
                        p[k].x = X; p[k].y = Y; p[k].z = Z;
                    }
                    p[k].rgba = (cr[k] << 16) | (cg[k] << 8) | cb[k];
                  }
                }
                if (!done) p[k] = reproject_one(a.Q, x, y, (double)d[k], cb[k], cg[k], cr[k], m, xf);
                lo[0] = fminf(lo[0], p[k].x); hi[0] = fmaxf(hi[0], p[k].x);
                lo[1] = fminf(lo[1], p[k].y); hi[1] = fmaxf(hi[1], p[k].y);
                lo[2] = fminf(lo[2], p[k].z); hi[2] = fmaxf(hi[2], p[k].z);
            }
        }
    }
    uint32_t total;
    uint32_t pos = block_excl_scan_u32<kEmitThreads / 64>(__popc(valid), scan_lds, total);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (valid & (1u << k)) {
            stage[pos] = make_uint4(__float_as_uint(p[k].x), __float_as_uint(p[k].y), __float_as_uint(p[k].z), p[k].rgba);
            ++pos;
        }
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(out + (int64_t)f * a.out_fstride + n_kp[f] + tile_off[(int64_t)f * a.n_tiles + tile]);
    for (uint32_t i = threadIdx.x; i < total; i += kEmitThreads) dst[i] = stage[i];
    block_minmax_store<kEmitThreads / 64>(lo, hi, valid != 0, mm_lds, mm + ((int64_t)f * a.mm_stride + tile) * 6);
}

// keypoint pass (pose_functions.cpp:1057-1091): one workgroup walks the keypoints in order
// one workgroup per frame; kp_off (optional) holds the frames' ranges in kp_xy, else all n_kp belong to frame 0
__global__ __launch_bounds__(256) void k_keypoint_pass(ReprojectArgs a, const float* __restrict__ kp_xy, int n_kp,
                                                       const int32_t* __restrict__ kp_off, o3dr_point* __restrict__ out,
                                                       uint32_t* __restrict__ n_kp_out, float* __restrict__ mm)
{
    __shared__ uint32_t scan_lds[256 / 64 + 1];
    __shared__ float mm_lds[6 * 4];
    const int f = blockIdx.x;
    if (kp_off) {
        kp_xy += 2 * (int64_t)kp_off[f];
        n_kp = kp_off[f + 1] - kp_off[f];
    }
    const uint8_t* disp = a.disp + (int64_t)f * a.disp_fstride;
    const uint8_t* bgr = a.bgr + (int64_t)f * a.bgr_fstride;
    out += (int64_t)f * a.out_fstride;
    float m[12];
    const bool xf = a.xf_mode != 0;
    for (int i = 0; i < 12; ++i) m[i] = (a.xf_mode == 2) ? a.poses[16 * (int64_t)f + i] : a.T[i];
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    bool any = false;
    uint32_t base = 0;
    for (int i0 = 0; i0 < n_kp; i0 += 256) {
        const int i = i0 + threadIdx.x;
        bool ok = false;
        Pix p;
        if (i < n_kp) {
            const int x = (int)kp_xy[2 * i], y = (int)kp_xy[2 * i + 1];
            if (x >= a.cs && x < a.cols - a.bb && y >= a.bb && y < a.rows - a.bb) {
                const uint8_t* row = disp + (int64_t)y * a.disp_pitch;
                const double d = a.disp_f64 ? reinterpret_cast<const double*>(row)[x] : (double)row[x];
                if (d > a.min_disp) {
                    const uint8_t* px = bgr + (int64_t)y * a.bgr_pitch + 3 * (int64_t)x;
                    p = reproject_one(a.Q, x, y, d, px[0], px[1], px[2], m, xf);
                    ok = true;
                }
            }
        }
        uint32_t total;
        const uint32_t pos = block_excl_scan_u32<4>(ok ? 1u : 0u, scan_lds, total);
        if (ok) {
            o3dr_point q;
            q.x = p.x; q.y = p.y; q.z = p.z; q.rgba = p.rgba;
            out[base + pos] = q;
            any = true;
            lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
            lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
            lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
        }
        base += total;
    }
    if (threadIdx.x == 0) n_kp_out[f] = base;
    block_minmax_store<4>(lo, hi, any, mm_lds, mm + ((int64_t)f * a.mm_stride + a.n_tiles) * 6);  // slot after the grid tiles
}

// A2 alone: pcl::transformPointCloud on an existing cloud (also the in-place re-transform of
// cloud_big, pose.cpp:353)
struct Mat34 {
    float m[12];
};
__global__ __launch_bounds__(kPtThreads) void k_transform(const o3dr_point* __restrict__ in, int64_t n, Mat34 T,
                                                          o3dr_point* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kPtThreads + threadIdx.x;
    if (i >= n) return;
    const uint4 v = reinterpret_cast<const uint4*>(in)[i];
    const float x = __uint_as_float(v.x), y = __uint_as_float(v.y), z = __uint_as_float(v.z);
    const float* m = T.m;
    const float X = ((m[0] * x + m[1] * y) + m[2] * z) + m[3];
    const float Y = ((m[4] * x + m[5] * y) + m[6] * z) + m[7];
    const float Z = ((m[8] * x + m[9] * y) + m[10] * z) + m[11];
    reinterpret_cast<uint4*>(out)[i] = make_uint4(__float_as_uint(X), __float_as_uint(Y), __float_as_uint(Z), v.w);
}

// =================================================================================================
// small bookkeeping kernels
// =================================================================================================
// neutral bounding-box slot `slot` of every frame (the keypoint-pass slot when there is no keypoint pass)
__global__ void k_minmax_init(float* mm, int64_t mm_stride, int slot, uint32_t* n_kp, int frames)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < frames * 6) {
        const int f = i / 6, a = i % 6;
        mm[((int64_t)f * mm_stride + slot) * 6 + a] = (a < 3) ? __builtin_inff() : -__builtin_inff();
    }
    if (n_kp && i < frames) n_kp[i] = 0;
}
__global__ void k_set_counts(uint32_t* n_dev, uint32_t value, int frames)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < frames) n_dev[i] = value;
}

// Exclusive scan of `frames` independent rows of length L (row f at data + f*row_stride), in place.
// One 1024-thread workgroup per row; each thread owns a contiguous chunk.  totals[f] (optional)
// receives the row sum plus add[f] (optional).  With `geom` the row is a radix histogram of pass
// `pass`: its live length is bins(f)*n_tiles, or nothing when that frame skips the pass.
__device__ __forceinline__ int64_t scan_row_len(const VoxelGeom* __restrict__ geom, int f, int pass, int n_tiles,
                                                int64_t L)
{
    if (!geom) return L;
    const VoxelGeom g = geom[f];
    if (pass < 0)  // one entry per run record + 1 (run lengths); nothing for frames that sort points
        return (g.overflow || g.val_bits == 0u) ? 0 : (int64_t)g.n + 1;
    if (g.overflow || pass >= (int)g.passes) return 0;
    return ((int64_t)1 << g.bpp) * n_tiles;
}
__global__ __launch_bounds__(1024) void k_scan_rows(uint32_t* __restrict__ data, int64_t L_in, int64_t row_stride,
                                                    uint32_t* __restrict__ totals, const uint32_t* __restrict__ add,
                                                    const VoxelGeom* __restrict__ geom, int pass, int n_tiles,
                                                    int chunked)
{
    __shared__ uint32_t lds[1024 / 64 + 1];
    uint32_t* row = data + (int64_t)blockIdx.x * row_stride;
    int64_t L = scan_row_len(geom, blockIdx.x, pass, n_tiles, L_in);
    if (chunked) L = (L + 4095) / 4096;  // scanning the chunk sums of a chunked scan
    if (L == 0 && !totals) return;
    const int64_t chunk = (L + 1023) / 1024;
    const int64_t b = (int64_t)threadIdx.x * chunk;
    const int64_t e = (b + chunk < L) ? b + chunk : L;
    uint32_t s = 0;
    for (int64_t i = b; i < e; ++i) s += row[i];
    uint32_t total;
    uint32_t run = block_excl_scan_u32<16>(s, lds, total);
    for (int64_t i = b; i < e; ++i) {
        const uint32_t t = row[i];
        row[i] = run;
        run += t;
    }
    if (totals && threadIdx.x == 0) totals[blockIdx.x] = total + (add ? add[blockIdx.x] : 0u);
}

// Multi-workgroup form for long rows (radix histograms, big merges): chunk sums -> k_scan_rows over
// the chunk sums -> per-chunk scan with the chunk's base.  A chunk is 4096 words, 16 per lane.
constexpr int kScanChunk = 4096;
__global__ __launch_bounds__(256) void k_scan_chunk_sums(const uint32_t* __restrict__ data, int64_t L_in,
                                                         int64_t row_stride, int n_chunks,
                                                         uint32_t* __restrict__ partial,
                                                         const VoxelGeom* __restrict__ geom, int pass, int n_tiles)
{
    __shared__ uint32_t lds[4];
    const int64_t L = scan_row_len(geom, blockIdx.y, pass, n_tiles, L_in);
    if ((int64_t)blockIdx.x * kScanChunk >= L) return;
    const uint32_t* row = data + (int64_t)blockIdx.y * row_stride;
    const int64_t b = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * 16;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (b + k < L) s += row[b + k];
    s = wave_sum_u32(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * n_chunks + blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}
__global__ __launch_bounds__(256) void k_scan_chunk_apply(uint32_t* __restrict__ data, int64_t L_in, int64_t row_stride,
                                                          int n_chunks, const uint32_t* __restrict__ partial_scanned,
                                                          const VoxelGeom* __restrict__ geom, int pass, int n_tiles)
{
    __shared__ uint32_t lds[5];
    const int64_t L = scan_row_len(geom, blockIdx.y, pass, n_tiles, L_in);
    if ((int64_t)blockIdx.x * kScanChunk >= L) return;
    uint32_t* row = data + (int64_t)blockIdx.y * row_stride;
    const int64_t b = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * 16;
    uint32_t v[16];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        v[k] = (b + k < L) ? row[b + k] : 0u;
        s += v[k];
    }
    uint32_t total;
    uint32_t run = block_excl_scan_u32<4>(s, lds, total) + partial_scanned[(int64_t)blockIdx.y * n_chunks + blockIdx.x];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (b + k < L) row[b + k] = run;
        run += v[k];
    }
}

// fp32 min/max of arbitrary clouds (stand-alone voxel grid calls; the fused path gets its
// bounding box from k_reproject_emit).  The `z += 500` of the combined mode (pose_functions.cpp:1666)
// is applied to the box afterwards by k_voxel_geom: fp32 addition is monotonic.
__global__ __launch_bounds__(kPtThreads) void k_points_minmax(const o3dr_point* __restrict__ in, int64_t in_fstride,
                                                              const uint32_t* __restrict__ n_dev, int64_t mm_stride,
                                                              float* __restrict__ mm)
{
    __shared__ float mm_lds[6 * (kPtThreads / 64)];
    const int f = blockIdx.y;
    const int64_t n = n_dev[f];
    const uint4* src = reinterpret_cast<const uint4*>(in + (int64_t)f * in_fstride);
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    bool any = false;
    const int64_t stride = (int64_t)gridDim.x * (kPtThreads * 4);
    for (int64_t base = (int64_t)blockIdx.x * (kPtThreads * 4); base < n; base += stride) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = base + k * kPtThreads + threadIdx.x;
            if (i < n) {
                const uint4 v = src[i];
                const float x = __uint_as_float(v.x), y = __uint_as_float(v.y), z = __uint_as_float(v.z);
                lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
                lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
                lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
                any = true;
            }
        }
    }
    block_minmax_store<kPtThreads / 64>(lo, hi, any, mm_lds, mm + ((int64_t)f * mm_stride + blockIdx.x) * 6);
}

// =================================================================================================
// Disparity pre-passes (A1 front end): cv::bilateralFilter (pose_functions.cpp:1040-1047) and the variance
// gate (pose_functions.cpp:987-1028, pose.cpp:187-196).
// =================================================================================================
constexpr int kBilTY = 32, kBilTX = 64;  // outputs per workgroup: lane = column, 8 rows per lane

__device__ __forceinline__ int reflect101(int p, int len)  // cv::borderInterpolate(p, len, BORDER_REFLECT_101)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        p = p < 0 ? -p : 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

// OpenCV 3.1 bilateralFilter_8u, one channel, as an x86-64 (SSE3) build evaluates it: neighbours in the order of
// the space table, w = color_weight[|v - v0|] * space_weight[k] and v*w in fp32; groups of four reduced as
// (a0+a1)+(a2+a3) and added to the running sums, the remaining < 4 one by one; result cvRound(sum / wsum).
// tab: color_weight[256] | space_weight[maxk] | tile offsets dy * (kBilTX + 2 radius) + dx as int32 [maxk].
__global__ __launch_bounds__(256) void k_bilateral_u8(const uint8_t* __restrict__ src, int64_t src_pitch, int64_t src_fstride,
                                                      int rows, int cols, int radius, int maxk,
                                                      const float* __restrict__ tab, uint8_t* __restrict__ dst,
                                                      int64_t dst_pitch, int64_t dst_fstride, int tiles_x)
{
    extern __shared__ uint8_t bil_lds[];
    const int TW = kBilTX + 2 * radius, TH = kBilTY + 2 * radius;
    float* cw = reinterpret_cast<float*>(bil_lds);  // 256 floats
    uint8_t* tile = bil_lds + 1024;
    const int f = blockIdx.y;
    const int y0 = ((int)blockIdx.x / tiles_x) * kBilTY, x0 = ((int)blockIdx.x % tiles_x) * kBilTX;
    const uint8_t* sf = src + (int64_t)f * src_fstride;
    cw[threadIdx.x] = tab[threadIdx.x];
    for (int i = threadIdx.x; i < TW * TH; i += 256) {
        const int ty = i / TW, tx = i - ty * TW;
        tile[i] = sf[(int64_t)reflect101(y0 - radius + ty, rows) * src_pitch + reflect101(x0 - radius + tx, cols)];
    }
    __syncthreads();
    const float* sw = tab + 256;
    const int* ofs = reinterpret_cast<const int*>(tab + 256 + maxk);
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    constexpr int NO = kBilTY / 4;  // outputs per lane
    int e0[NO], v0[NO];
    float sum[NO], wsum[NO];
#pragma unroll
    for (int i = 0; i < NO; ++i) {
        e0[i] = (r0 + 4 * i + radius) * TW + (c + radius);
        v0[i] = tile[e0[i]];
        sum[i] = 0.f;
        wsum[i] = 0.f;
    }
    int k = 0;
    for (; k <= maxk - 4; k += 4) {
        float swk[4];
        int ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            swk[q] = sw[k + q];  // wave-uniform: scalar loads
            ok[q] = ofs[k + q];
        }
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            float w[4], vw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int val = tile[e0[i] + ok[q]];
                w[q] = cw[abs(val - v0[i])] * swk[q];
                vw[q] = w[q] * (float)val;
            }
            const float ws = (w[0] + w[1]) + (w[2] + w[3]);
            const float vs = (vw[0] + vw[1]) + (vw[2] + vw[3]);
            sum[i] += vs;
            wsum[i] += ws;
        }
    }
    for (; k < maxk; ++k) {
        const float swk = sw[k];
        const int ok = ofs[k];
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const int val = tile[e0[i] + ok];
            const float w = swk * cw[abs(val - v0[i])];
            sum[i] += (float)val * w;
            wsum[i] += w;
        }
    }
    uint8_t* df = dst + (int64_t)f * dst_fstride;
#pragma unroll
    for (int i = 0; i < NO; ++i) {
        const int y = y0 + r0 + 4 * i, x = x0 + c;
        if (y < rows && x < cols) df[(int64_t)y * dst_pitch + x] = (uint8_t)__float2int_rn(sum[i] / wsum[i]);
    }
}

// histogram of the ROI's valid disparities (d > min_disparity), one row of 256 counters per frame
__global__ __launch_bounds__(256) void k_disp_hist(const uint8_t* __restrict__ disp, int64_t pitch, int64_t fstride, int rows,
                                                   int cols, int bb, int cs, double min_disp,
                                                   unsigned long long* __restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int f = blockIdx.y;
    const int W = cols - bb - cs;
    const int y = bb + (int)blockIdx.x;
    if (y < rows - bb && W > 0) {
        const uint8_t* row = disp + (int64_t)f * fstride + (int64_t)y * pitch + cs;
        for (int x = threadIdx.x; x < W; x += 256) {
            const uint32_t d = row[x];
            if ((double)d > min_disp) atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[(int64_t)f * 256 + threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
// mean: the reference's sequential fp64 sum of integers is exact, so sum_d h[d] * d reproduces it bit for bit;
// variance: sum_d h[d] * (d - mean)^2 in ascending d instead of pixel order (differs from the sequential sum only
// by fp64 rounding, at most N * 2^-53 relative for N pixels)
__global__ void k_disp_variance(const unsigned long long* __restrict__ hist, int frames, int rows, int cols, int bb, int cs,
                                double* __restrict__ var_out)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    const unsigned long long* h = hist + (int64_t)f * 256;
    unsigned long long s = 0;
    for (int d = 0; d < 256; ++d) s += h[d] * (unsigned long long)d;
    const int roi = (rows - 2 * bb) * (cols - bb - cs);
    const double mean = (double)s / roi;
    double temp = 0;
    for (int d = 0; d < 256; ++d) temp += (double)h[d] * (((double)d - mean) * ((double)d - mean));
    var_out[f] = temp / (roi - 1);
}

// =================================================================================================
// Pixel-window voxel grouping (fused A6 path, rectified-stereo Q, small leaf).
//   Two pixels can only fall into the same (voxel_size/5) voxel if they carry the same disparity byte and
//   lie within W(d) = floor(1.01*leaf*sqrt(3) / (|Q0| * |1/w(d)|)) pixels of each other (the host checks that
//   neighbouring disparity levels are more than leaf*sqrt(3) apart in depth; o3dr_api.hip).  So voxels are
//   formed inside an LDS tile with halo: the pixel with the lowest row-major index of a voxel (its head)
//   adds up the voxel's pixels in row-major order — the same strictly sequential fp32 sums, in the same
//   order, as the sort-based path — and only the (index, centroid) records of the voxels are sorted.
// =================================================================================================
// pass 1: like k_reproject_count, plus the frame's bounding box (it is needed before any voxel index)
__global__ __launch_bounds__(kEmitThreads) void k_frame_bbox(ReprojectArgs a, uint32_t* __restrict__ tile_cnt,
                                                             float* __restrict__ mm)
{
    __shared__ uint32_t lds[kEmitThreads / 64];
    __shared__ float mm_lds[6 * (kEmitThreads / 64)];
    __shared__ double lut_alpha[256];
    __shared__ float lut_z[256];
    lut_alpha[threadIdx.x] = a.lut[threadIdx.x].alpha;
    lut_z[threadIdx.x] = a.lut[threadIdx.x].z;
    __syncthreads();
    const int f = blockIdx.y, tile = blockIdx.x;
    const uint8_t* disp = a.disp + (int64_t)f * a.disp_fstride;
    const int n_cand = a.Ny * a.Nx;
    const int c0 = tile * kEmitTile + threadIdx.x * kEmitPerLane;
    float m[12];
    const float* T = a.poses + 16 * (int64_t)f;
#pragma unroll
    for (int i = 0; i < 12; ++i) m[i] = T[i];
    int x0 = 0, y0 = 0;
    uint32_t d[4];
    const uint32_t valid = load_lane_disparities<false>(a, disp, c0, n_cand, x0, y0, d);
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (valid & (1u << k)) {
            int x, y;
            if (a.vec4) {
                x = x0 + k;
                y = y0;
            } else {
                const int c = c0 + k;
                const int ry = c / a.Nx, rx = c - ry * a.Nx;
                y = a.bb + ry * a.jump;
                x = a.cs + rx * a.jump;
            }
            const double al = lut_alpha[d[k]];
            const float X = (float)((a.Q[0] * (double)x + a.Q[3]) * al + 0.0);
            const float Y = (float)((a.Q[5] * (double)y + a.Q[7]) * al + 0.0);
            const float Z = lut_z[d[k]];
            const float px = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
            const float py = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
            const float pz = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
            lo[0] = fminf(lo[0], px); hi[0] = fmaxf(hi[0], px);
            lo[1] = fminf(lo[1], py); hi[1] = fmaxf(hi[1], py);
            lo[2] = fminf(lo[2], pz); hi[2] = fmaxf(hi[2], pz);
        }
    }
    const uint32_t sc = wave_sum_u32(__popc(valid));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = sc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < kEmitThreads / 64; ++i) t += lds[i];
        tile_cnt[(int64_t)f * a.n_tiles + tile] = t;
    }
    block_minmax_store<kEmitThreads / 64>(lo, hi, valid != 0, mm_lds, mm + ((int64_t)f * a.mm_stride + tile) * 6);
}

constexpr int kWinTY = 16, kWinTX = 64;          // core tile: 1024 candidates; lane = column, 4 rows per lane
constexpr int kWinRS = kWinTX + 2 * kWinHalo;    // LDS row stride (fixed, so window offsets are compile-time constants)
constexpr int kWinRows = kWinTY + 2 * kWinHalo;
constexpr int kWinPtRows = kWinTY + kWinHalo;    // rows that can hold pixels to be summed (tile + lower halo)
constexpr int kWinStage = (kWinRows * kWinRS + 255) / 256;  // region pixels per lane
constexpr uint32_t kNoKey = 0xffffffffu;

__device__ __forceinline__ void win_add(const uint4 p, float& sx, float& sy, float& sz, float& sr, float& sg, float& sb)
{
    sx += __uint_as_float(p.x);
    sy += __uint_as_float(p.y);
    sz += __uint_as_float(p.z) + 0.0f;
    sr += (float)((p.w >> 16) & 255u);
    sg += (float)((p.w >> 8) & 255u);
    sb += (float)(p.w & 255u);
}

// Search + sums for window radius W (block-uniform: the largest radius any core pixel of the tile needs; a compile-time
// constant for W <= 4 so that all window offsets are immediates, a runtime bound above).
// Looking further than a pixel's own radius is harmless: an equal index anywhere in the tile IS the same voxel.
template <int W>
__device__ __forceinline__ void win_heads(const uint32_t* s_key, const uint4* s_pt, int wt, uint32_t (&hk)[4], uint4 (&hc)[4],
                                          uint32_t& nh)
{
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 4 * k;
        const int e0 = (r + wt) * kWinRS + (c + wt);
        const uint32_t key0 = s_key[e0];
        // any earlier pixel (row-major) with the same index?  independent LDS reads
        bool before = false;
#pragma unroll
        for (int dx = 1; dx <= W; ++dx) before |= s_key[e0 - dx] == key0;
#pragma unroll
        for (int dy = 1; dy <= W; ++dy) {
#pragma unroll
            for (int dx = -W; dx <= W; ++dx) before |= s_key[e0 - dy * kWinRS + dx] == key0;
        }
        const bool head = key0 != kNoKey && !before;
        if (__ballot(head) == 0ull) continue;  // wave-uniform
        if (head) {
            // the voxel's pixels in row-major order: strictly sequential fp32 sums, like the sort-based path
            const uint4* pt0 = s_pt + (r * kWinRS + (c + wt));  // point rows start at the tile's first row
            float sx = 0.f, sy = 0.f, sz = 0.f, sr = 0.f, sg = 0.f, sb = 0.f;
            uint32_t np = 1;
            win_add(pt0[0], sx, sy, sz, sr, sg, sb);
            if (W > 0) {
                uint32_t mk = 0;
#pragma unroll
                for (int dx = 1; dx <= W; ++dx) mk |= (s_key[e0 + dx] == key0 ? 1u : 0u) << (dx - 1);
                while (mk) {
                    const int dx = __ffs(mk);
                    mk &= mk - 1;
                    win_add(pt0[dx], sx, sy, sz, sr, sg, sb);
                    ++np;
                }
#pragma unroll
                for (int dy = 1; dy <= W; ++dy) {
                    mk = 0;
#pragma unroll
                    for (int dx = -W; dx <= W; ++dx) mk |= (s_key[e0 + dy * kWinRS + dx] == key0 ? 1u : 0u) << (dx + W);
                    while (mk) {
                        const int bx = __ffs(mk) - 1 - W;
                        mk &= mk - 1;
                        win_add(pt0[dy * kWinRS + bx], sx, sy, sz, sr, sg, sb);
                        ++np;
                    }
                }
            }
            const float nf = (float)np;
            const float sa = 0.f;  // alpha bytes are 0 (pose_functions.cpp:1120)
            const uint32_t rgba = ((uint32_t)(sa / nf) << 24) | ((uint32_t)(sr / nf) << 16) | ((uint32_t)(sg / nf) << 8) |
                                  (uint32_t)(sb / nf);
            const uint4 cv = make_uint4(__float_as_uint(sx / nf), __float_as_uint(sy / nf), __float_as_uint(sz / nf - 0.0f), rgba);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
                if (nh == (uint32_t)sl) {
                    hk[sl] = key0;
                    hc[sl] = cv;
                }
            ++nh;
        }
    }
}

// the same with a runtime radius (radii above 4: rare, close-range pixels)
__device__ __noinline__ void win_heads_any(const uint32_t* s_key, const uint4* s_pt, int wt, uint32_t (&hk)[4], uint4 (&hc)[4],
                                           uint32_t& nh)
{
    const int W = wt;
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 4 * k;
        const int e0 = (r + wt) * kWinRS + (c + wt);
        const uint32_t key0 = s_key[e0];
        bool before = false;
        for (int dx = 1; dx <= W; ++dx) before |= s_key[e0 - dx] == key0;
        for (int dy = 1; dy <= W; ++dy)
            for (int dx = -W; dx <= W; ++dx) before |= s_key[e0 - dy * kWinRS + dx] == key0;
        const bool head = key0 != kNoKey && !before;
        if (__ballot(head) == 0ull) continue;
        if (head) {
            const uint4* pt0 = s_pt + (r * kWinRS + (c + wt));
            float sx = 0.f, sy = 0.f, sz = 0.f, sr = 0.f, sg = 0.f, sb = 0.f;
            uint32_t np = 1;
            win_add(pt0[0], sx, sy, sz, sr, sg, sb);
            for (int dx = 1; dx <= W; ++dx)
                if (s_key[e0 + dx] == key0) {
                    win_add(pt0[dx], sx, sy, sz, sr, sg, sb);
                    ++np;
                }
            for (int dy = 1; dy <= W; ++dy)
                for (int dx = -W; dx <= W; ++dx)
                    if (s_key[e0 + dy * kWinRS + dx] == key0) {
                        win_add(pt0[dy * kWinRS + dx], sx, sy, sz, sr, sg, sb);
                        ++np;
                    }
            const float nf = (float)np;
            const float sa = 0.f;
            const uint32_t rgba = ((uint32_t)(sa / nf) << 24) | ((uint32_t)(sr / nf) << 16) | ((uint32_t)(sg / nf) << 8) |
                                  (uint32_t)(sb / nf);
            const uint4 cv = make_uint4(__float_as_uint(sx / nf), __float_as_uint(sy / nf), __float_as_uint(sz / nf - 0.0f), rgba);
            for (int sl = 0; sl < 4; ++sl)
                if (nh == (uint32_t)sl) {
                    hk[sl] = key0;
                    hc[sl] = cv;
                }
            ++nh;
        }
    }
}

// One workgroup per (strip of kWinTY candidate rows, frame): tables, pose and row terms are set up once, then the
// strip's tiles are walked left to right.
__global__ __launch_bounds__(256) void k_window_group(ReprojectArgs a, const float* __restrict__ wbase,
                                                      const float* __restrict__ win_c, int tiles_x,
                                                      const VoxelGeom* __restrict__ geom,
                                                      const VoxelGeom* __restrict__ geom_gen, int64_t cap,
                                                      uint32_t* __restrict__ keys_out, o3dr_point* __restrict__ cent_out,
                                                      uint32_t* __restrict__ n_heads)
{
    __shared__ uint4 s_pt[kWinPtRows * kWinRS];    // world point + (R << 16 | G << 8 | B) of the tile and lower halo
    __shared__ uint32_t s_key[kWinRows * kWinRS];  // voxel index of every region pixel (kNoKey: outside / invalid)
    __shared__ double lut_alpha[256];
    __shared__ float lut_z[256];
    __shared__ float s_bu[256], s_bv[256];
    __shared__ double s_xt[kWinRS], s_yt[kWinRows];  // Q0*x + Q3 per region column, Q5*y + Q7 per region row
    __shared__ uint32_t scan_lds[5];
    __shared__ uint32_t base_lds;
    __shared__ int wt_lds[4];
    const int f = blockIdx.y;
    const VoxelGeom g = geom[f];
    if (g.n == 0 || geom_gen[f].n != 0) return;  // empty, or left to the sort-based path by k_window_plan
    const int ty0 = (int)blockIdx.x * kWinTY;
    const uint8_t* disp = a.disp + (int64_t)f * a.disp_fstride;
    const uint8_t* bgr = a.bgr + (int64_t)f * a.bgr_fstride;
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    // ---- once per strip
    s_bu[threadIdx.x] = wbase[threadIdx.x];
    s_bv[threadIdx.x] = wbase[256 + threadIdx.x];
    lut_alpha[threadIdx.x] = a.lut[threadIdx.x].alpha;
    lut_z[threadIdx.x] = a.lut[threadIdx.x].z;
    // the same fp64 operations as the per-pixel form (Q5 * y + Q7, then * 1/w), shared by a row
    if (threadIdx.x < kWinRows)
        s_yt[threadIdx.x] = a.Q[5] * (double)(a.bb + (ty0 - kWinHalo + (int)threadIdx.x) * a.jump) + a.Q[7];
    float m[12];
    const float* T = a.poses + 16 * (int64_t)f;
#pragma unroll
    for (int i = 0; i < 12; ++i) m[i] = T[i];
    const float cu = win_c[2 * f], cv = win_c[2 * f + 1];
    const bool wide = a.bb >= 1;  // 4-byte colour loads may touch the first byte of the next pixel: needs a margin
    uint32_t dnext[4];  // core disparities of the next tile (loaded one tile ahead)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ry = ty0 + r0 + 4 * k;
        dnext[k] = (ry < a.Ny && c < a.Nx) ? disp[(int64_t)(a.bb + ry * a.jump) * a.disp_pitch + (a.cs + c * a.jump)] : 0u;
    }
    for (int tile = 0; tile < tiles_x; ++tile) {
        const int tx0 = tile * kWinTX;
        // ---- radius this tile needs: the largest one over its own (core) pixels
        if (threadIdx.x < kWinRS)
            s_xt[threadIdx.x] = a.Q[0] * (double)(a.cs + (tx0 - kWinHalo + (int)threadIdx.x) * a.jump) + a.Q[3];
        if (tile == 0) __syncthreads();  // tables
        int wmax = -1;
        {
            const int rx = tx0 + c;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ry = ty0 + r0 + 4 * k;
                if (ry < a.Ny && rx < a.Nx && (double)dnext[k] > a.min_disp) {
                    const float fu = cu * s_bu[dnext[k]], fv = cv * s_bv[dnext[k]];
                    wmax = max(wmax, max((int)(fu + fu * 1e-6f), (int)(fv + fv * 1e-6f)));
                }
            }
            const int rxn = rx + kWinTX;  // prefetch the next tile's core disparities
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ry = ty0 + r0 + 4 * k;
                dnext[k] = (tile + 1 < tiles_x && ry < a.Ny && rxn < a.Nx)
                               ? disp[(int64_t)(a.bb + ry * a.jump) * a.disp_pitch + (a.cs + rxn * a.jump)] : 0u;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
        if ((threadIdx.x & 63) == 0) wt_lds[threadIdx.x >> 6] = wmax;
        __syncthreads();
        const int wt = min(max(max(wt_lds[0], wt_lds[1]), max(wt_lds[2], wt_lds[3])), kWinHalo);
        if (wt < 0) {  // no valid pixel in the tile (block-uniform)
            __syncthreads();
            continue;
        }
        // ---- stage the region (tile + halo of wt): voxel index and point of every pixel.
        //      All loads of a lane's pixels are issued before any is used: one memory round trip.
        const int RXt = kWinTX + 2 * wt, RYt = kWinTY + 2 * wt;
        const int n_region = RXt * RYt;
        const uint32_t inv_rx = ((1u << 20) + (uint32_t)RXt - 1u) / (uint32_t)RXt;  // i / RXt == (i * inv_rx) >> 20 for i < 2560
        {
            uint32_t dv[kWinStage], col[kWinStage];
            uint32_t in_mask = 0;
#pragma unroll
            for (int it = 0; it < kWinStage; ++it) {
                const int i = threadIdx.x + 256 * it;
                const int ey = (int)(((uint32_t)i * inv_rx) >> 20), ex = i - ey * RXt;
                const int ry = ty0 - wt + ey, rx = tx0 - wt + ex;
                dv[it] = col[it] = 0u;
                if (i < n_region && ry >= 0 && ry < a.Ny && rx >= 0 && rx < a.Nx) {
                    in_mask |= 1u << it;
                    const int y = a.bb + ry * a.jump, x = a.cs + rx * a.jump;
                    const uint8_t* px = bgr + (int64_t)y * a.bgr_pitch + 3 * (int64_t)x;
                    dv[it] = disp[(int64_t)y * a.disp_pitch + x];
                    if (wide) {
                        uint32_t v;
                        __builtin_memcpy(&v, px, 4);  // unaligned dword load: B | G << 8 | R << 16 | (next B) << 24
                        col[it] = v & 0xffffffu;
                    } else
                        col[it] = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
                }
            }
#pragma unroll
            for (int it = 0; it < kWinStage; ++it) {
                const int i = threadIdx.x + 256 * it;
                if (i < n_region) {
                    const int ey = (int)(((uint32_t)i * inv_rx) >> 20), ex = i - ey * RXt;
                    uint32_t key = kNoKey;
                    if (((in_mask >> it) & 1u) && (double)dv[it] > a.min_disp) {
                        const double al = lut_alpha[dv[it]];
                        const float X = (float)(s_xt[ex + kWinHalo - wt] * al + 0.0);
                        const float Y = (float)(s_yt[ey + kWinHalo - wt] * al + 0.0);
                        const float Z = lut_z[dv[it]];
                        const float wx = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
                        const float wy = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
                        const float wz = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
                        const int32_t i0 = (int32_t)floorf(wx * g.inv[0]) - g.min_b[0];
                        const int32_t i1 = (int32_t)floorf(wy * g.inv[1]) - g.min_b[1];
                        const int32_t i2 = (int32_t)floorf(wz * g.inv[2]) - g.min_b[2];
                        key = (uint32_t)i0 + (uint32_t)i1 * g.mul1 + (uint32_t)i2 * g.mul2;
                        if (ey >= wt)
                            s_pt[(ey - wt) * kWinRS + ex] = make_uint4(__float_as_uint(wx), __float_as_uint(wy),
                                                                       __float_as_uint(wz), col[it]);
                    }
                    s_key[ey * kWinRS + ex] = key;
                }
            }
        }
        __syncthreads();
        // ---- heads (lowest row-major pixel of a voxel) add up their voxel
        uint32_t hk[4];
        uint4 hc[4];
        uint32_t nh = 0;
        switch (wt) {
            case 0: win_heads<0>(s_key, s_pt, wt, hk, hc, nh); break;
            case 1: win_heads<1>(s_key, s_pt, wt, hk, hc, nh); break;
            case 2: win_heads<2>(s_key, s_pt, wt, hk, hc, nh); break;
            case 3: win_heads<3>(s_key, s_pt, wt, hk, hc, nh); break;
            case 4: win_heads<4>(s_key, s_pt, wt, hk, hc, nh); break;
            default: win_heads_any(s_key, s_pt, wt, hk, hc, nh); break;
        }
        // ---- append the tile's voxels to the frame's record list (any order: the indices are unique and get sorted)
        uint32_t total;
        const uint32_t pos = block_excl_scan_u32<4>(nh, scan_lds, total);
        if (threadIdx.x == 0) base_lds = total ? atomicAdd(&n_heads[f], total) : 0u;
        __syncthreads();
        const int64_t o = (int64_t)f * cap + base_lds + pos;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((uint32_t)k < nh) {
                keys_out[o + k] = hk[k];
                reinterpret_cast<uint4*>(cent_out)[o + k] = hc[k];
            }
        }
        __syncthreads();  // base_lds, s_key, s_pt are free for the next tile
    }
}

// which frames of the batch take the window path, and with which window.  Two pixels whose computed world points
// share a voxel differ by less than leaf (+ 2 rounding errors) along every world axis, hence along camera X by at
// most sum_i |R^-1[0][i]| times that.  For a pose within 1 % of orthonormal the sum is below |R00|+|R10|+|R20| + 0.031;
// with the rounding budget (err <= 0.025 leaf, checked here) the bound is (1.06 L1 + 0.05) leaf =: c_u leaf, same for Y.
// The host table holds leaf / (pixel footprint at disparity d); the radius is floor(c * table[d]).
// geom_gen is what the sort-based kernels see (n = 0: nothing to do), n_heads the record counters of the window kernel.
__global__ void k_window_plan(const VoxelGeom* __restrict__ geom, const float* __restrict__ poses, int frames, float rho_max,
                              float err_budget, VoxelGeom* __restrict__ geom_gen, uint32_t* __restrict__ n_heads,
                              float* __restrict__ win_c)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    VoxelGeom g = geom[f];
    const float* T = poses + 16 * (int64_t)f;
    // || R^T R - I ||_F <= 0.01
    float dev = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const float d = T[i] * T[j] + T[4 + i] * T[4 + j] + T[8 + i] * T[8 + j] - (i == j ? 1.f : 0.f);
            dev += d * d;
        }
    const float tn = sqrtf(T[3] * T[3] + T[7] * T[7] + T[11] * T[11]);
    // |error of a computed world coordinate| <= 8 half-ulps of (range + |t|): 3 products, 3 sums, rounded inputs
    const float err = 4.76837158e-7f * (rho_max + tn);  // 2^-21
    const float cu = 1.06f * (fabsf(T[0]) + fabsf(T[4]) + fabsf(T[8])) + 0.05f;
    const float cv = 1.06f * (fabsf(T[1]) + fabsf(T[5]) + fabsf(T[9])) + 0.05f;
    // NaN poses fail the comparisons; kWinCMax is the factor the host sized the halo and the depth-level test with
    const bool window = g.n != 0 && !g.overflow && dev <= 1e-4f && err <= err_budget && cu <= kWinCMax && cv <= kWinCMax;
    if (window) g.n = 0;
    geom_gen[f] = g;
    n_heads[f] = 0;
    win_c[2 * f] = cu;
    win_c[2 * f + 1] = cv;
}
// geometry the sort sees: window frames sort their voxel records (buffer 1 first; bit 1 of buf0 marks them)
__global__ void k_window_sort_geom(const VoxelGeom* __restrict__ geom, const VoxelGeom* __restrict__ geom_gen,
                                   const uint32_t* __restrict__ n_heads, int frames, VoxelGeom* __restrict__ geom_sort)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    VoxelGeom g = geom[f];
    if (g.n != 0 && geom_gen[f].n == 0) {
        g.n = n_heads[f];
        g.buf0 = 3;
    }
    geom_sort[f] = g;
}

// sorted voxel records of the window frames -> output
__global__ __launch_bounds__(kPtThreads) void k_gather_heads(const o3dr_point* __restrict__ pts, int64_t cap,
                                                             const uint32_t* __restrict__ ids0,
                                                             const uint32_t* __restrict__ ids1,
                                                             const VoxelGeom* __restrict__ geom_runs,
                                                             const uint32_t* __restrict__ n_out,
                                                             const uint64_t* __restrict__ out_off,
                                                             o3dr_point* __restrict__ out_base)
{
    const int f = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * kPtThreads + threadIdx.x;
    const VoxelGeom g = geom_runs[f];
    if (!(g.buf0 & 2u) || j >= n_out[f]) return;
    const uint4* src = reinterpret_cast<const uint4*>(pts + (int64_t)f * cap);
    uint4* dst = reinterpret_cast<uint4*>(out_base + out_off[f]);
    dst[j] = src[(sorted_buf(g, ids0, ids1) + (int64_t)f * cap)[j]];
}

// =================================================================================================
// K2a — PCL VoxelGrid geometry and per-point linear index
//   [PCL 1.8 filters/impl/voxel_grid.hpp applyFilter; called from pose_functions.cpp:1689-1700]
// =================================================================================================
__global__ __launch_bounds__(256) void k_voxel_geom(const float* __restrict__ mm, int64_t mm_stride, int mm_used,
                                                    const uint32_t* __restrict__ n_dev, float leaf0, float leaf1,
                                                    float leaf2, float z_offset, VoxelGeom* __restrict__ geom)
{
    __shared__ float red[6 * 4];
    const int f = blockIdx.x;
    // getMinMax3D: fold the per-workgroup boxes of this frame (min/max are order independent)
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    const float* row = mm + (int64_t)f * mm_stride * 6;
    for (int sidx = threadIdx.x; sidx < mm_used; sidx += 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], row[sidx * 6 + a]);
            hi[a] = fmaxf(hi[a], row[sidx * 6 + 3 + a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = wave_min_f32(lo[a]), h = wave_max_f32(hi[a]);
        if ((threadIdx.x & 63) == 0) {
            red[(threadIdx.x >> 6) * 6 + a] = l;
            red[(threadIdx.x >> 6) * 6 + 3 + a] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    VoxelGeom g;
    const float leaf[3] = {leaf0, leaf1, leaf2};
    g.n = n_dev[f];
    g.overflow = 0;
    g.passes = 0;
    g.bpp = 8;
    g.buf0 = 0;
    g.val_bits = 0;
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) {
        g.inv[a] = 1.0f / leaf[a];  // inverse_leaf_size_ = Array4f::Ones() / leaf_size_
        mn[a] = fminf(fminf(red[a], red[6 + a]), fminf(red[12 + a], red[18 + a]));
        mx[a] = fmaxf(fmaxf(red[3 + a], red[9 + a]), fmaxf(red[15 + a], red[21 + a]));
    }
    // the bounding box was taken before `z += 500` (pose_functions.cpp:1666); fp32 add is monotonic
    mn[2] = mn[2] + z_offset;
    mx[2] = mx[2] + z_offset;
    if (g.n == 0) {
        for (int a = 0; a < 3; ++a) g.min_b[a] = 0, g.div_b[a] = 1;
        g.mul1 = g.mul2 = 1;
        geom[f] = g;
        return;
    }
    // int64_t dx = static_cast<int64_t>((max_p[0]-min_p[0])*inverse_leaf_size_[0]) + 1; ...
    int64_t d[3];
    for (int a = 0; a < 3; ++a) d[a] = (int64_t)((mx[a] - mn[a]) * g.inv[a]) + 1;
    g.overflow = (d[0] * d[1] * d[2]) > (int64_t)INT32_MAX ? 1u : 0u;
    for (int a = 0; a < 3; ++a) {
        g.min_b[a] = (int32_t)floorf(mn[a] * g.inv[a]);
        const int32_t max_b = (int32_t)floorf(mx[a] * g.inv[a]);
        g.div_b[a] = max_b - g.min_b[a] + 1;
    }
    g.mul1 = (uint32_t)g.div_b[0];
    g.mul2 = (uint32_t)g.div_b[0] * (uint32_t)g.div_b[1];
    if (!g.overflow) {
        // sort plan: the linear index is < div_b.x*div_b.y*div_b.z, so only that many bits are sorted,
        // in the fewest passes of at most kMaxRadixBits bits (a wrapped 32-bit index sorts all 32 bits)
        const uint64_t cells = (uint64_t)(uint32_t)g.div_b[0] * (uint64_t)(uint32_t)g.div_b[1] * (uint64_t)(uint32_t)g.div_b[2];
        uint32_t nbits = 32;
        if (cells <= (1ull << 32)) nbits = cells > 1 ? 64u - (uint32_t)__clzll((long long)(cells - 1)) : 1u;
        if (nbits < 1) nbits = 1;
        g.passes = (nbits + kMaxRadixBits - 1) / kMaxRadixBits;
        g.bpp = (nbits + g.passes - 1) / g.passes;
    }
    geom[f] = g;
}

// idx = ijk0*divb_mul[0] + ijk1*divb_mul[1] + ijk2*divb_mul[2] with
// ijk = int(floor(p*inverse_leaf) - float(min_b));  floor and subtraction are exact here, so the
// difference is taken in integers.
__global__ __launch_bounds__(kPtThreads) void k_voxel_keys(const o3dr_point* __restrict__ in, int64_t in_fstride,
                                                           const VoxelGeom* __restrict__ geom, float z_offset,
                                                           int64_t cap, uint32_t* __restrict__ keys)
{
    const int f = blockIdx.y;
    const VoxelGeom g = geom[f];
    if (g.overflow) return;
    const uint4* src = reinterpret_cast<const uint4*>(in + (int64_t)f * in_fstride);
    uint32_t* dst = keys + (int64_t)f * cap;
    const int64_t base = (int64_t)blockIdx.x * (kPtThreads * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * kPtThreads + threadIdx.x;
        if (i < g.n) {
            const uint4 v = src[i];
            const float x = __uint_as_float(v.x), y = __uint_as_float(v.y), z = __uint_as_float(v.z) + z_offset;
            const int32_t i0 = (int32_t)floorf(x * g.inv[0]) - g.min_b[0];
            const int32_t i1 = (int32_t)floorf(y * g.inv[1]) - g.min_b[1];
            const int32_t i2 = (int32_t)floorf(z * g.inv[2]) - g.min_b[2];
            dst[i] = (uint32_t)i0 + (uint32_t)i1 * g.mul1 + (uint32_t)i2 * g.mul2;
        }
    }
}

// k_voxel_keys + the head count of k_run_heads(buf_sel 0) in one read of the points (run-compressed calls):
// a point starts a run iff its index differs from its predecessor's (or the packed length would overflow).
// One workgroup = one segment tile (kSegTile points, 4 sub-rows of 256 consecutive points).
__device__ __forceinline__ uint32_t voxel_key_of(const uint4 v, const VoxelGeom& g, float z_offset)
{
    const float x = __uint_as_float(v.x), y = __uint_as_float(v.y), z = __uint_as_float(v.z) + z_offset;
    const int32_t i0 = (int32_t)floorf(x * g.inv[0]) - g.min_b[0];
    const int32_t i1 = (int32_t)floorf(y * g.inv[1]) - g.min_b[1];
    const int32_t i2 = (int32_t)floorf(z * g.inv[2]) - g.min_b[2];
    return (uint32_t)i0 + (uint32_t)i1 * g.mul1 + (uint32_t)i2 * g.mul2;
}
__global__ __launch_bounds__(256) void k_voxel_keys_heads(const o3dr_point* __restrict__ in, int64_t in_fstride,
                                                          const VoxelGeom* __restrict__ geom, float z_offset, int64_t cap,
                                                          uint32_t* __restrict__ keys, int n_tiles,
                                                          uint32_t* __restrict__ seg_cnt)
{
    static_assert(kSegTile == 4 * 256, "one workgroup per segment tile");
    __shared__ uint32_t lds[4];
    const int f = blockIdx.y, tile = blockIdx.x;
    const VoxelGeom g = geom[f];
    if (g.overflow) return;
    const uint4* src = reinterpret_cast<const uint4*>(in + (int64_t)f * in_fstride);
    uint32_t* dst = keys + (int64_t)f * cap;
    const uint32_t split = run_split_mask(run_start_bits(g.n));
    const int64_t base = (int64_t)tile * kSegTile;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        uint32_t key = 0;
        if (i < g.n) {
            key = voxel_key_of(src[i], g, z_offset);
            dst[i] = key;
        }
        uint32_t prev = __shfl_up(key, 1, 64);
        if ((threadIdx.x & 63) == 0 && i > 0 && i < g.n) prev = voxel_key_of(src[i - 1], g, z_offset);  // previous wave's point
        if (i < g.n) c += (((uint32_t)i & split) == 0u || key != prev) ? 1u : 0u;
    }
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) seg_cnt[(int64_t)f * n_tiles + tile] = lds[0] + lds[1] + lds[2] + lds[3];
}

// Single-pass variant: the same index computation, plus per-workgroup digit histograms for EVERY pass
// of the frame's sort plan (the digits do not depend on record order), written as partial tables
// (no global atomics) and folded by k_digit_starts.
constexpr int kKeyThreads = 512;
__global__ __launch_bounds__(kKeyThreads) void k_voxel_keys_hist(const o3dr_point* __restrict__ in, int64_t in_fstride,
                                                                 const VoxelGeom* __restrict__ geom, float z_offset,
                                                                 int64_t cap, uint32_t* __restrict__ keys, int nblk,
                                                                 uint32_t* __restrict__ partial)
{
    __shared__ uint32_t h[kMaxPasses * kMaxRadix];
    const int f = blockIdx.y, blk = blockIdx.x;
    const VoxelGeom g = geom[f];
    if (g.overflow) return;
    for (int i = threadIdx.x; i < kMaxPasses * kMaxRadix; i += kKeyThreads) h[i] = 0;
    __syncthreads();
    const uint4* src = reinterpret_cast<const uint4*>(in + (int64_t)f * in_fstride);
    uint32_t* dst = keys + (int64_t)f * cap;
    const int passes = (int)g.passes, bpp = (int)g.bpp;
    const uint32_t dmask = (1u << bpp) - 1u;
    // contiguous slice of the frame per workgroup, in chunks of 4 x 512 points
    const int64_t per = ((((int64_t)g.n + nblk - 1) / nblk) + kKeyThreads * 4 - 1) / (kKeyThreads * 4) * (kKeyThreads * 4);
    const int64_t b0 = (int64_t)blk * per;
    const int64_t b1 = (b0 + per < (int64_t)g.n) ? b0 + per : (int64_t)g.n;
    const int lane = threadIdx.x & 63;
    for (int64_t base = b0; base < b1; base += kKeyThreads * 4) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = base + k * kKeyThreads + threadIdx.x;
            if (i < b1) v[k] = src[i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = base + k * kKeyThreads + threadIdx.x;
            const bool ok = i < b1;
            uint32_t key = 0;
            if (ok) {
                const float x = __uint_as_float(v[k].x), y = __uint_as_float(v[k].y), z = __uint_as_float(v[k].z) + z_offset;
                const int32_t i0 = (int32_t)floorf(x * g.inv[0]) - g.min_b[0];
                const int32_t i1 = (int32_t)floorf(y * g.inv[1]) - g.min_b[1];
                const int32_t i2 = (int32_t)floorf(z * g.inv[2]) - g.min_b[2];
                key = (uint32_t)i0 + (uint32_t)i1 * g.mul1 + (uint32_t)i2 * g.mul2;
                dst[i] = key;
            }
            const uint64_t okb = __ballot(ok);
            if (okb == 0) continue;
            const int leader = __ffsll((long long)okb) - 1;
            for (int p = 0; p < passes; ++p) {
                const uint32_t dgt = (key >> (p * bpp)) & dmask;
                const uint32_t first = __shfl(dgt, leader, 64);
                if (__ballot(ok && dgt != first) == 0) {  // whole wave in one bin (high digits): one add
                    if (lane == leader) atomicAdd(&h[p * kMaxRadix + first], (uint32_t)__popcll(okb));
                } else if (ok) {
                    atomicAdd(&h[p * kMaxRadix + dgt], 1u);
                }
            }
        }
    }
    __syncthreads();
    uint32_t* out = partial + ((int64_t)f * nblk + blk) * (kMaxPasses * kMaxRadix);
    const int bins = 1 << bpp;
    for (int i = threadIdx.x; i < passes * kMaxRadix; i += kKeyThreads)
        if ((i & (kMaxRadix - 1)) < bins) out[i] = h[i];
}

// per frame and pass: exclusive scan over the digits of the summed partial histograms; resets the tickets
__global__ __launch_bounds__(kMaxRadix) void k_digit_starts(const uint32_t* __restrict__ partial, int nblk,
                                                            const VoxelGeom* __restrict__ geom,
                                                            uint32_t* __restrict__ digit_start,
                                                            uint32_t* __restrict__ tickets)
{
    __shared__ uint32_t scan_lds[kMaxRadix / 64 + 1];
    const int f = blockIdx.x;
    const VoxelGeom g = geom[f];
    if (threadIdx.x < kMaxPasses) tickets[f * kMaxPasses + threadIdx.x] = 0;
    if (g.overflow) return;
    const int bins = 1 << g.bpp;
    for (int p = 0; p < (int)g.passes; ++p) {
        uint32_t v = 0;
        if ((int)threadIdx.x < bins)
            for (int b = 0; b < nblk; ++b) v += partial[(((int64_t)f * nblk + b) * kMaxPasses + p) * kMaxRadix + threadIdx.x];
        uint32_t total;
        const uint32_t excl = block_excl_scan_u32<kMaxRadix / 64>(v, scan_lds, total);
        if ((int)threadIdx.x < bins) digit_start[((int64_t)f * kMaxPasses + p) * kMaxRadix + threadIdx.x] = excl;
    }
}

// =================================================================================================
// K2b — stable LSD radix sort of (voxel index, point id).
//   PCL sorts with std::sort (order inside a voxel unspecified); the canonical order here is the
//   stable one: points of a voxel stay in ascending input order.  Each frame sorts only the bits its
//   index can take, in g.passes passes of g.bpp <= 10 bits (k_voxel_geom); pass p reads buffer p&1
//   and writes the other, so a frame's sorted records end in buffer g.passes&1.  A workgroup owns
//   8192 consecutive records, a wave 1024 of them, visited in 16 rounds of 64 lanes so that
//   (round, lane) order is input order.  Ranks come from wave ballots (match-any on the digit) plus
//   per-wave LDS counters: no atomics, hence stable and deterministic.
// =================================================================================================
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(const uint32_t* __restrict__ keys0,
                                                             const uint32_t* __restrict__ keys1, int64_t cap,
                                                             const VoxelGeom* __restrict__ geom, int pass,
                                                             int n_tiles, uint32_t* __restrict__ hist)
{
    // private tables per wave AND per lane quarter (4 x 8 x 128 counters): an LDS add only collides
    // with the 15 other lanes of its quarter, and those spread over 128 bins
    constexpr int kSub = 4;
    __shared__ uint32_t h[kSub * kSortWaves * kMaxRadix];
    const int f = blockIdx.y, tile = blockIdx.x;
    const VoxelGeom g = geom[f];
    if (g.overflow || pass >= (int)g.passes) return;
    const uint32_t n = g.n;
    const int bins = 1 << g.bpp, shift = pass * (int)g.bpp;
    const uint32_t dmask = (uint32_t)bins - 1u;
    for (int i = threadIdx.x; i < kSub * kSortWaves * kMaxRadix; i += kSortThreads) h[i] = 0;
    __syncthreads();
    const uint32_t* src = (((pass + g.buf0) & 1) ? keys1 : keys0) + (int64_t)f * cap;
    const int64_t base = (int64_t)tile * kSortTile;
    uint32_t* hw = h + (((threadIdx.x >> 6) * kSub) + (threadIdx.x & (kSub - 1))) * kMaxRadix;
    if (base < n) {
        // a histogram does not care which lane sees which record: 16-byte loads, 4 per lane
        const bool vec = (((int64_t)f * cap) & 3) == 0;  // tile bases are multiples of 8192
#pragma unroll
        for (int r = 0; r < kSortRounds / 4; ++r) {
            const int64_t i = base + ((int64_t)r * kSortThreads + threadIdx.x) * 4;
            if (vec && i + 3 < n) {
                const uint4 v = *reinterpret_cast<const uint4*>(src + i);
                atomicAdd(&hw[(v.x >> shift) & dmask], 1u);
                atomicAdd(&hw[(v.y >> shift) & dmask], 1u);
                atomicAdd(&hw[(v.z >> shift) & dmask], 1u);
                atomicAdd(&hw[(v.w >> shift) & dmask], 1u);
            } else {
                for (int k = 0; k < 4; ++k)
                    if (i + k < n) atomicAdd(&hw[(src[i + k] >> shift) & dmask], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* dst = hist + (int64_t)f * kMaxRadix * n_tiles;
    for (int dgt = threadIdx.x; dgt < bins; dgt += kSortThreads) {
        uint32_t t = 0;
#pragma unroll
        for (int ww = 0; ww < kSub * kSortWaves; ++ww) t += h[ww * kMaxRadix + dgt];
        dst[(int64_t)dgt * n_tiles + tile] = t;
    }
}

// k_voxel_keys + k_radix_hist of pass 0 in one read of the points (per-frame grids of the batched path): one
// workgroup = one sort tile; the indices are written for the scatter, their lowest digit is counted on the way.
__global__ __launch_bounds__(kSortThreads) void k_voxel_keys_hist0(const o3dr_point* __restrict__ in, int64_t in_fstride,
                                                                   const VoxelGeom* __restrict__ geom, float z_offset,
                                                                   int64_t cap, uint32_t* __restrict__ keys, int n_tiles,
                                                                   uint32_t* __restrict__ hist)
{
    constexpr int kSub = 4;
    __shared__ uint32_t h[kSub * kSortWaves * kMaxRadix];
    const int f = blockIdx.y, tile = blockIdx.x;
    const VoxelGeom g = geom[f];
    if (g.overflow) return;
    const uint32_t n = g.n;
    const int64_t base = (int64_t)tile * kSortTile;  // (tiles past the end still write their zeros: the scan reads them)
    const int bins = 1 << g.bpp;
    const uint32_t dmask = (uint32_t)bins - 1u;
    for (int i = threadIdx.x; i < kSub * kSortWaves * kMaxRadix; i += kSortThreads) h[i] = 0;
    __syncthreads();
    const uint4* src = reinterpret_cast<const uint4*>(in + (int64_t)f * in_fstride);
    uint32_t* dst = keys + (int64_t)f * cap;
    uint32_t* hw = h + (((threadIdx.x >> 6) * kSub) + (threadIdx.x & (kSub - 1))) * kMaxRadix;
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t i = base + (int64_t)r * kSortThreads + threadIdx.x;
        if (i < n) {
            const uint32_t key = voxel_key_of(src[i], g, z_offset);
            dst[i] = key;
            atomicAdd(&hw[key & dmask], 1u);
        }
    }
    __syncthreads();
    uint32_t* hd = hist + (int64_t)f * kMaxRadix * n_tiles;
    for (int dgt = threadIdx.x; dgt < bins; dgt += kSortThreads) {
        uint32_t t = 0;
#pragma unroll
        for (int ww = 0; ww < kSub * kSortWaves; ++ww) t += h[ww * kMaxRadix + dgt];
        hd[(int64_t)dgt * n_tiles + tile] = t;
    }
}

// Chained-scan ("look-back") state of the single-pass variant: one 64-bit word per (tile, digit),
//   [63:42] epoch of the launch that wrote it   [41:40] 1 = tile's own count, 2 = inclusive prefix
//   [39:0]  value.
// The word is written with ONE agent-scope 8-byte store and polled with agent-scope relaxed loads
// (MI355X_MICROARCH.md, visibility: "8-B agent atomics both sides"); the epoch makes stale words
// of earlier launches unreadable without clearing the array.  Tiles take a ticket in start order, so
// every tile a workgroup waits for has started before it: no wait can deadlock.
constexpr uint64_t kLbLocal = 1ull << 40, kLbIncl = 2ull << 40, kLbValueMask = (1ull << 40) - 1ull;
constexpr uint32_t kLbSpinLimit = 1u << 22;

template <bool kLookback>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(uint32_t* __restrict__ keys0, uint32_t* __restrict__ vals0,
                                                                uint32_t* __restrict__ keys1, uint32_t* __restrict__ vals1,
                                                                int64_t cap, const VoxelGeom* __restrict__ geom,
                                                                int pass, int n_tiles,
                                                                const uint32_t* __restrict__ hist_scanned,
                                                                const uint32_t* __restrict__ digit_start,
                                                                uint64_t* __restrict__ lb_state,
                                                                uint32_t* __restrict__ tickets, uint32_t epoch,
                                                                uint32_t* __restrict__ error_flag,
                                                                const uint32_t* __restrict__ run_start)
{
    __shared__ uint32_t wave_cnt[kSortWaves * kMaxRadix];  // per-wave digit counts -> exclusive wave prefixes
    __shared__ uint32_t local_base[kMaxRadix];             // start of each digit inside the tile's sorted order
    __shared__ uint32_t delta[kMaxRadix];                  // global start of (digit, tile) - local_base
    __shared__ uint32_t stage[kSortTile];                  // the tile's keys, then ids, in sorted order
    __shared__ uint32_t scan_lds[kSortWaves + 1];
    __shared__ uint32_t ticket_lds;
    const int f = blockIdx.y;
    const VoxelGeom g = geom[f];
    if (g.overflow || pass >= (int)g.passes) return;
    int tile = blockIdx.x;
    if (kLookback) {  // tiles are numbered in the order their workgroups start
        if (threadIdx.x == 0) ticket_lds = atomicAdd(&tickets[f * kMaxPasses + pass], 1u);
        __syncthreads();
        tile = (int)ticket_lds;
    }
    const uint32_t n = g.n;
    const int64_t base = (int64_t)tile * kSortTile;
    if (base >= n) return;
    const uint32_t cnt = (n - base < (uint32_t)kSortTile) ? (uint32_t)(n - base) : (uint32_t)kSortTile;
    const int bpp = (int)g.bpp, bins = 1 << bpp, shift = pass * bpp;
    const uint32_t dmask = (uint32_t)bins - 1u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int par = (pass + (int)g.buf0) & 1;  // which of the two buffers this pass reads
    const uint32_t* kin = (par ? keys1 : keys0) + (int64_t)f * cap;
    const uint32_t* vin = (par ? vals1 : vals0) + (int64_t)f * cap;
    uint32_t* kout = (par ? keys0 : keys1) + (int64_t)f * cap;
    uint32_t* vout = (par ? vals0 : vals1) + (int64_t)f * cap;

    for (int i = threadIdx.x; i < kSortWaves * kMaxRadix; i += kSortThreads) wave_cnt[i] = 0;
    __syncthreads();

    // ---- 1. load 16 records per lane; rank every record among the wave's earlier same-digit records
    uint32_t key[kSortRounds], val[kSortRounds], rank[kSortRounds];
    const int64_t wbase = base + (int64_t)w * kSortWaveItems;
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t i = wbase + r * kWave + lane;
        const bool ok = i < n;
        key[r] = ok ? kin[i] : 0xffffffffu;
        val[r] = (pass == 0) ? (uint32_t)i : (ok ? vin[i] : 0u);
        if (pass == 0 && g.val_bits != 0u && ok) {  // records are runs: payload = (first point, length)
            const uint32_t* rs = run_start + (int64_t)f * (cap + 1);
            val[r] = run_pack(rs[i], rs[i + 1], g.val_bits);
        }
    }
    volatile uint32_t* wc = wave_cnt + w * kMaxRadix;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t i = wbase + r * kWave + lane;
        const bool ok = i < n;
        const uint32_t dgt = (key[r] >> shift) & dmask;
        uint64_t peers = __ballot(ok);  // lanes of this round holding the same digit
#pragma unroll
        for (int b = 0; b < kMaxRadixBits; ++b) {
            if (b < bpp) {
                const bool bit = (dgt >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
        }
        uint32_t prior = 0;
        if (ok) prior = wc[dgt];
        rank[r] = prior + (uint32_t)__popcll(peers & lt_mask);
        __builtin_amdgcn_wave_barrier();
        if (ok && (peers & lt_mask) == 0) wc[dgt] = prior + (uint32_t)__popcll(peers);  // lowest peer updates
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // ---- 2. two adjacent digits per thread: wave prefixes, tile totals, local starts, global deltas
    uint32_t tot[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int dg = 2 * threadIdx.x + j;
        if (dg < bins) {
#pragma unroll
            for (int ww = 0; ww < kSortWaves; ++ww) {
                const uint32_t t = wave_cnt[ww * kMaxRadix + dg];
                wave_cnt[ww * kMaxRadix + dg] = tot[j];
                tot[j] += t;
            }
        }
    }
    uint32_t tile_total;
    uint32_t lb = block_excl_scan_u32<kSortWaves>(tot[0] + tot[1], scan_lds, tile_total);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int dg = 2 * threadIdx.x + j;
        if (dg < bins) {
            uint32_t gbase;
            if (kLookback) {
                // publish this tile's count, fold the predecessors' words back to the nearest inclusive
                // prefix, publish the inclusive prefix
                uint64_t* st = lb_state + ((int64_t)f * n_tiles) * kMaxRadix;
                const uint64_t tag = (uint64_t)epoch << 42;
                uint64_t excl = 0;
                if (tile > 0) {
                    __hip_atomic_store(&st[(int64_t)tile * kMaxRadix + dg], tag | kLbLocal | (uint64_t)tot[j], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    for (int t = tile - 1; t >= 0; --t) {
                        uint64_t wv;
                        uint32_t spins = 0;
                        for (;;) {
                            wv = __hip_atomic_load(&st[(int64_t)t * kMaxRadix + dg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((wv >> 42) == (uint64_t)epoch && (wv & (kLbLocal | kLbIncl))) break;
                            // never expected; fail loudly instead of hanging the GPU
                            if (++spins > kLbSpinLimit ||
                                ((spins & 1023u) == 0 && __hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                                atomicOr(error_flag, 1u);
                                wv = kLbIncl;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(2);
                        }
                        excl += wv & kLbValueMask;
                        if (wv & kLbIncl) break;
                    }
                }
                __hip_atomic_store(&st[(int64_t)tile * kMaxRadix + dg], tag | kLbIncl | (excl + (uint64_t)tot[j]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                gbase = digit_start[((int64_t)f * kMaxPasses + pass) * kMaxRadix + dg] + (uint32_t)excl;
            } else {
                gbase = hist_scanned[(int64_t)f * kMaxRadix * n_tiles + (int64_t)dg * n_tiles + tile];
            }
            local_base[dg] = lb;
            delta[dg] = gbase - lb;
        }
        lb += tot[j];
    }
    __syncthreads();

    // ---- 3. keys: into LDS at their sorted position, out to HBM in runs of equal digit (coalesced)
    uint32_t pos[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t i = wbase + r * kWave + lane;
        if (i < n) {
            const uint32_t dgt = (key[r] >> shift) & dmask;
            pos[r] = local_base[dgt] + wave_cnt[w * kMaxRadix + dgt] + rank[r];
            stage[pos[r]] = key[r];
        }
    }
    __syncthreads();
    uint32_t dst[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t p = r * kSortThreads + threadIdx.x;
        if (p < cnt) {
            const uint32_t k = stage[p];
            dst[r] = p + delta[(k >> shift) & dmask];
            kout[dst[r]] = k;
        }
    }
    __syncthreads();
    // ---- 4. ids the same way, reusing the positions
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t i = wbase + r * kWave + lane;
        if (i < n) stage[pos[r]] = val[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t p = r * kSortThreads + threadIdx.x;
        if (p < cnt) vout[dst[r]] = stage[p];
    }
}

// Lane-counting variant of the stable scatter (classic, per-tile histogram offsets).  A lane owns 16
// CONSECUTIVE records, so input order = (lane, slot) order and ranking needs no cross-lane matching:
//   A. every lane adds 1 to its own byte of cnt[digit][lane/4] (four lanes share a dword) with a
//      returning LDS add: the old byte is the record's rank among the lane's earlier records;
//   B. per digit row (one lane per row): running sum over the 16 dwords; each dword is repacked to
//      [31:20] = records of lower lane-quads, [19:0] = the four lanes' counts (5 bits each);
//   C. rank in wave = quad prefix + counts of lower lanes in the quad + own rank.
// ~300 VALU per 1024 records instead of ~1000 for ballot matching (k_radix_scatter), which PMC showed
// to be VALU-issue-bound.  Digits are at most 7 bits wide (kMaxRadix = 128).
constexpr int kCntStride = 17;  // dwords per digit row: 16 lane-quads + 1 pad (conflict-free row walks)
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter_lane(uint32_t* __restrict__ keys0, uint32_t* __restrict__ vals0,
                                                                     uint32_t* __restrict__ keys1, uint32_t* __restrict__ vals1,
                                                                     int64_t cap, const VoxelGeom* __restrict__ geom, int pass,
                                                                     int n_tiles, const uint32_t* __restrict__ hist_scanned,
                                                                     const uint32_t* __restrict__ run_start)
{
    constexpr int kCntWords = kMaxRadix * kCntStride;  // per wave
    constexpr int kRankWords = kSortWaves * kCntWords;
    __shared__ __attribute__((aligned(16))) uint32_t smem[kRankWords > kSortTile ? kRankWords : kSortTile];
    __shared__ uint32_t wave_tot[kSortWaves * kMaxRadix];  // per-wave digit totals -> exclusive wave prefixes
    __shared__ uint32_t local_base[kMaxRadix];
    __shared__ uint32_t delta[kMaxRadix];
    __shared__ uint32_t scan_lds[kSortWaves + 1];
    uint32_t* stage = smem;  // overlays the counters once every record knows its position
    const int f = blockIdx.y, tile = blockIdx.x;
    const VoxelGeom g = geom[f];
    if (g.overflow || pass >= (int)g.passes) return;
    const uint32_t n = g.n;
    const int64_t base = (int64_t)tile * kSortTile;
    if (base >= n) return;
    const uint32_t cnt = (n - base < (uint32_t)kSortTile) ? (uint32_t)(n - base) : (uint32_t)kSortTile;
    const int bpp = (int)g.bpp, bins = 1 << bpp, shift = pass * bpp;
    const uint32_t dmask = (uint32_t)bins - 1u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int par = (pass + (int)g.buf0) & 1;  // which of the two buffers this pass reads
    const uint32_t* kin = (par ? keys1 : keys0) + (int64_t)f * cap;
    const uint32_t* vin = (par ? vals1 : vals0) + (int64_t)f * cap;
    uint32_t* kout = (par ? keys0 : keys1) + (int64_t)f * cap;
    uint32_t* vout = (par ? vals0 : vals1) + (int64_t)f * cap;

    for (int i = threadIdx.x; i < kRankWords; i += kSortThreads) smem[i] = 0;
    __syncthreads();

    // ---- load: 16 consecutive records per lane (four 16-byte loads when the frame base allows it)
    uint32_t key[kSortRounds], val[kSortRounds];
    const int64_t lbase = base + (int64_t)w * kSortWaveItems + (int64_t)lane * kSortRounds;
    const bool vec = ((((int64_t)f * cap) & 3) == 0) && (lbase + kSortRounds <= (int64_t)n);
    if (vec) {
#pragma unroll
        for (int q = 0; q < kSortRounds / 4; ++q) {
            const uint4 k4 = *reinterpret_cast<const uint4*>(kin + lbase + 4 * q);
            key[4 * q] = k4.x; key[4 * q + 1] = k4.y; key[4 * q + 2] = k4.z; key[4 * q + 3] = k4.w;
        }
        if (pass != 0) {
#pragma unroll
            for (int q = 0; q < kSortRounds / 4; ++q) {
                const uint4 v4 = *reinterpret_cast<const uint4*>(vin + lbase + 4 * q);
                val[4 * q] = v4.x; val[4 * q + 1] = v4.y; val[4 * q + 2] = v4.z; val[4 * q + 3] = v4.w;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) {
            const int64_t i = lbase + r;
            const bool ok = i < n;
            key[r] = ok ? kin[i] : 0xffffffffu;
            val[r] = (pass != 0 && ok) ? vin[i] : 0u;
        }
    }
    if (pass == 0) {
        if (g.val_bits != 0u) {  // records are runs: payload = (first point, length) from the run starts
            const uint32_t* rs = run_start + (int64_t)f * (cap + 1);
            uint32_t nxt = lbase < (int64_t)n ? rs[lbase] : 0u;
#pragma unroll
            for (int r = 0; r < kSortRounds; ++r) {
                const uint32_t first = nxt;
                nxt = (lbase + r < (int64_t)n) ? rs[lbase + r + 1] : 0u;
                val[r] = (lbase + r < (int64_t)n) ? run_pack(first, nxt, g.val_bits) : 0u;
            }
        } else {
#pragma unroll
            for (int r = 0; r < kSortRounds; ++r) val[r] = (uint32_t)(lbase + r);
        }
    }

    // ---- A. own-lane ranks
    uint32_t* cw = smem + w * kCntWords;
    const int quad = lane >> 2, sub = lane & 3;
    uint32_t own[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        own[r] = 0;
        if (lbase + r < (int64_t)n) {
            const uint32_t dgt = (key[r] >> shift) & dmask;
            const uint32_t old = atomicAdd(&cw[dgt * kCntStride + quad], 1u << (8 * sub));
            own[r] = (old >> (8 * sub)) & 255u;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- B. per digit row: quad prefixes, repack, wave total (lanes 0..63 take rows lane and lane+64)
#pragma unroll
    for (int h = 0; h < kMaxRadix / 64; ++h) {
        const int row = lane + 64 * h;
        if (row < bins) {
            uint32_t run = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t v = cw[row * kCntStride + j];
                const uint32_t c0 = v & 255u, c1 = (v >> 8) & 255u, c2 = (v >> 16) & 255u, c3 = v >> 24;
                cw[row * kCntStride + j] = (run << 20) | c0 | (c1 << 5) | (c2 << 10) | (c3 << 15);
                run += c0 + c1 + c2 + c3;
            }
            wave_tot[w * kMaxRadix + row] = run;
        }
    }
    __syncthreads();

    // ---- 2. one digit per thread: wave prefixes, tile total, local start, global delta
    uint32_t tot = 0;
    const int dg = threadIdx.x;
    if (dg < bins) {
#pragma unroll
        for (int ww = 0; ww < kSortWaves; ++ww) {
            const uint32_t t = wave_tot[ww * kMaxRadix + dg];
            wave_tot[ww * kMaxRadix + dg] = tot;
            tot += t;
        }
    }
    uint32_t tile_total;
    const uint32_t lb = block_excl_scan_u32<kSortWaves>(tot, scan_lds, tile_total);
    if (dg < bins) {
        local_base[dg] = lb;
        delta[dg] = hist_scanned[(int64_t)f * kMaxRadix * n_tiles + (int64_t)dg * n_tiles + tile] - lb;
    }
    __syncthreads();

    // ---- C. positions inside the tile's sorted order
    uint32_t pos[kSortRounds];
    const uint32_t below = (1u << (5 * sub)) - 1u;  // count fields of the lower lanes of the quad
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        if (lbase + r < (int64_t)n) {
            const uint32_t dgt = (key[r] >> shift) & dmask;
            const uint32_t v = cw[dgt * kCntStride + quad];
            const uint32_t lowf = v & below;
            const uint32_t in_quad = (lowf & 31u) + ((lowf >> 5) & 31u) + ((lowf >> 10) & 31u);
            pos[r] = local_base[dgt] + wave_tot[w * kMaxRadix + dgt] + (v >> 20) + in_quad + own[r];
        }
    }
    __syncthreads();  // the staging buffer overlays the counters: everyone is done reading them
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r)
        if (lbase + r < (int64_t)n) stage[pos[r]] = key[r];
    __syncthreads();
    uint32_t dst[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t p = r * kSortThreads + threadIdx.x;
        if (p < cnt) {
            const uint32_t k = stage[p];
            dst[r] = p + delta[(k >> shift) & dmask];
            kout[dst[r]] = k;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r)
        if (lbase + r < (int64_t)n) stage[pos[r]] = val[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t p = r * kSortThreads + threadIdx.x;
        if (p < cnt) vout[dst[r]] = stage[p];
    }
}

// =================================================================================================
// K2c — runs of equal index -> one output point each
//   third/fourth pass of VoxelGrid::applyFilter + CentroidPoint<PointXYZRGB>
//   [PCL 1.8 common/impl/accumulators.hpp: fp32 sums, xyz / n, uint32_t(channel / n)]
// =================================================================================================

// buf_sel < 0: the frame's sorted buffer; 0/1: that buffer as is (runs of an UNSORTED sequence)
// four consecutive records of a segment tile per lane (one 16-byte load when the frame base allows it) and the key
// just before them: lane order = record order, so one workgroup scan per tile ranks the run heads
__device__ __forceinline__ void load_seg_keys(const uint32_t* __restrict__ k, int64_t i0, uint32_t n, bool vec, uint32_t kk[4],
                                              uint32_t& prev)
{
    if (vec && i0 + 3 < (int64_t)n) {
        const uint4 v = *reinterpret_cast<const uint4*>(k + i0);
        kk[0] = v.x; kk[1] = v.y; kk[2] = v.z; kk[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) kk[q] = (i0 + q < (int64_t)n) ? k[i0 + q] : 0u;
    }
    prev = __shfl_up(kk[3], 1, 64);
    if ((threadIdx.x & 63) == 0) prev = (i0 > 0 && i0 < (int64_t)n) ? k[i0 - 1] : 0u;  // previous wave's / tile's last key
}

__global__ __launch_bounds__(256) void k_run_heads(const uint32_t* __restrict__ keys0, const uint32_t* __restrict__ keys1,
                                                   int64_t cap, const VoxelGeom* __restrict__ geom, int n_tiles,
                                                   uint32_t* __restrict__ seg_cnt, int buf_sel)
{
    static_assert(kSegTile == 4 * 256, "four records per lane");
    __shared__ uint32_t lds[4];
    const int f = blockIdx.y, tile = blockIdx.x;
    const VoxelGeom g = geom[f];
    if (g.overflow) return;
    const uint32_t n = g.n;
    const uint32_t* k = (buf_sel < 0 ? sorted_buf(g, keys0, keys1) : (buf_sel ? keys1 : keys0)) + (int64_t)f * cap;
    // forming runs of points (buf_sel 0): also cut where the packed length would overflow
    const uint32_t split = buf_sel == 0 ? run_split_mask(run_start_bits(n)) : 0xffffffffu;
    uint32_t c = 0;
    const int64_t i0 = (int64_t)tile * kSegTile + threadIdx.x * 4;
    if ((int64_t)tile * kSegTile < n) {
        uint32_t kk[4], prev;
        load_seg_keys(k, i0, n, ((((int64_t)f * cap) & 3) == 0), kk, prev);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t i = i0 + q;
            if (i < n) c += (((uint32_t)i & split) == 0u || kk[q] != (q ? kk[q - 1] : prev)) ? 1u : 0u;
        }
    }
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) seg_cnt[(int64_t)f * n_tiles + tile] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ __launch_bounds__(256) void k_run_starts(const uint32_t* __restrict__ keys0, const uint32_t* __restrict__ keys1,
                                                    int64_t cap, const VoxelGeom* __restrict__ geom, int n_tiles,
                                                    const uint32_t* __restrict__ seg_off,
                                                    const uint32_t* __restrict__ n_vox,
                                                    uint32_t* __restrict__ seg_start, int buf_sel,
                                                    uint32_t* __restrict__ head_keys_out,
                                                    const VoxelGeom* __restrict__ mode)
{
    __shared__ uint32_t scan_lds[5];
    const int f = blockIdx.y, tile = blockIdx.x;
    const VoxelGeom g = geom[f];
    if (g.overflow) return;
    if (mode && mode[f].val_bits == 0u) return;  // this cloud sorts its points, not runs of them (k_run_geom)
    const uint32_t n = g.n;
    const int64_t base = (int64_t)tile * kSegTile;
    if (base >= n) return;
    const uint32_t* k = (buf_sel < 0 ? sorted_buf(g, keys0, keys1) : (buf_sel ? keys1 : keys0)) + (int64_t)f * cap;
    uint32_t* ss = seg_start + (int64_t)f * (cap + 1);
    if (tile == 0 && threadIdx.x == 0) ss[n_vox[f]] = n;  // sentinel: end of the last run
    const uint32_t off = seg_off[(int64_t)f * n_tiles + tile];
    const uint32_t split = buf_sel == 0 ? run_split_mask(run_start_bits(n)) : 0xffffffffu;
    const int64_t i0 = base + threadIdx.x * 4;
    uint32_t kk[4], prev;
    load_seg_keys(k, i0, n, ((((int64_t)f * cap) & 3) == 0), kk, prev);
    bool head[4];
    uint32_t cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = i0 + q;
        head[q] = (i < n) && (((uint32_t)i & split) == 0u || kk[q] != (q ? kk[q - 1] : prev));
        cnt += head[q] ? 1u : 0u;
    }
    uint32_t total;
    uint32_t pos = off + block_excl_scan_u32<4>(cnt, scan_lds, total);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (head[q]) {
            ss[pos] = (uint32_t)(i0 + q);
            if (head_keys_out) head_keys_out[(int64_t)f * cap + pos] = kk[q];  // run key = index of its first point
            ++pos;
        }
    }
}

// min_points_per_voxel > 1 (combined merge, pose_functions.cpp:1693): runs with fewer points are dropped
__global__ __launch_bounds__(256) void k_keep_count(const uint32_t* __restrict__ seg_start, int64_t cap,
                                                    const VoxelGeom* __restrict__ geom,
                                                    const uint32_t* __restrict__ n_vox, uint32_t min_points,
                                                    int n_tiles, uint32_t* __restrict__ seg_cnt)
{
    __shared__ uint32_t lds[4];
    const int f = blockIdx.y, tile = blockIdx.x;
    if (geom[f].overflow || geom[f].val_bits != 0u) return;  // (run records: k_keep_count_runs)
    const uint32_t nv = n_vox[f];
    const uint32_t* ss = seg_start + (int64_t)f * (cap + 1);
    uint32_t c = 0;
    const int64_t base = (int64_t)tile * kSegTile;
    if (base < nv) {
        for (int j = 0; j < kSegTile / 256; ++j) {
            const int64_t o = base + j * 256 + threadIdx.x;
            if (o < nv) c += (ss[o + 1] - ss[o] >= min_points) ? 1u : 0u;
        }
    }
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) seg_cnt[(int64_t)f * n_tiles + tile] = lds[0] + lds[1] + lds[2] + lds[3];
}
__global__ __launch_bounds__(256) void k_keep_write(const uint32_t* __restrict__ seg_start, int64_t cap,
                                                    const VoxelGeom* __restrict__ geom,
                                                    const uint32_t* __restrict__ n_vox, uint32_t min_points,
                                                    int n_tiles, const uint32_t* __restrict__ seg_off,
                                                    uint32_t* __restrict__ keep_idx)
{
    __shared__ uint32_t scan_lds[5];
    const int f = blockIdx.y, tile = blockIdx.x;
    if (geom[f].overflow || geom[f].val_bits != 0u) return;
    const uint32_t nv = n_vox[f];
    const int64_t base = (int64_t)tile * kSegTile;
    if (base >= nv) return;
    const uint32_t* ss = seg_start + (int64_t)f * (cap + 1);
    uint32_t* ki = keep_idx + (int64_t)f * cap;
    uint32_t off = seg_off[(int64_t)f * n_tiles + tile];
    for (int j = 0; j < kSegTile / 256; ++j) {
        const int64_t o = base + j * 256 + threadIdx.x;
        const bool keep = (o < nv) && (ss[o + 1] - ss[o] >= min_points);
        uint32_t total;
        const uint32_t pos = block_excl_scan_u32<4>(keep ? 1u : 0u, scan_lds, total);
        if (keep) ki[off + pos] = (uint32_t)o;
        off += total;
    }
}

// per-frame output counts -> absolute offsets; advances the cloud counter (one 256-thread workgroup,
// frames in chunks of 256; a batch's total stays far below 2^32 points)
__global__ __launch_bounds__(256) void k_frame_offsets(const VoxelGeom* __restrict__ geom, const uint32_t* __restrict__ n_vox,
                                                       const uint32_t* __restrict__ n_keep, int frames, int passthrough,
                                                       uint32_t* __restrict__ n_out, uint64_t* __restrict__ out_off,
                                                       CloudCounters* __restrict__ cc, SortStats* __restrict__ stats,
                                                       const VoxelGeom* __restrict__ sort_geom)
{
    __shared__ uint32_t scan_lds[5];
    __shared__ unsigned long long acc[6];  // record-passes, points in, points out, status, window frames, records
    if (threadIdx.x < 6) acc[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = cc->count;
    uint64_t carry = 0;
    for (int f0 = 0; f0 < frames; f0 += 256) {
        const int f = f0 + threadIdx.x;
        uint32_t m = 0;
        if (f < frames) {
            const VoxelGeom g = geom[f];
            if (passthrough) {
                m = g.n;
            } else if (g.overflow) {
                m = g.n;
                atomicOr(&acc[3], (unsigned long long)O3DR_STATUS_VOXEL_OVERFLOW);
            } else {
                m = n_keep ? n_keep[f] : n_vox[f];
                if (sort_geom[f].buf0 & 2u) {  // window frame: one record per voxel
                    m = sort_geom[f].n;
                    atomicAdd(&acc[4], 1ull);
                }
                atomicAdd(&acc[0], (unsigned long long)sort_geom[f].n * g.passes);  // records actually sorted
                atomicAdd(&acc[5], (unsigned long long)sort_geom[f].n);
                atomicAdd(&acc[1], (unsigned long long)g.n);
                atomicAdd(&acc[2], (unsigned long long)m);
            }
        }
        uint32_t total;
        const uint32_t excl = block_excl_scan_u32<4>(m, scan_lds, total);
        if (f < frames) {
            n_out[f] = m;
            out_off[f] = base + carry + excl;
        }
        carry += total;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cc->count = base + carry;
        cc->status |= (uint32_t)acc[3];
        if (stats) {
            stats->sort_record_passes += acc[0];
            stats->voxel_points_in += acc[1];
            stats->voxel_points_out += acc[2];
            stats->window_frames += acc[4];
            stats->sort_records += acc[5];
        }
    }
}

// outputs [256 bx, 256 bx + 256) of frame f
__device__ __forceinline__ void centroid_block(int64_t bx, int f, const VoxelGeom& g, uint32_t n_o,
                                               const o3dr_point* __restrict__ in, int64_t in_fstride,
                                               const uint32_t* __restrict__ vals0, const uint32_t* __restrict__ vals1,
                                               int64_t cap, const uint32_t* __restrict__ seg_start,
                                               const uint32_t* __restrict__ keep_idx, const uint64_t* __restrict__ out_off,
                                               float z_offset, int passthrough, o3dr_point* __restrict__ out_base,
                                               float* __restrict__ out_mm, int nbx)
{
    const int64_t o = bx * kPtThreads + threadIdx.x;
    const bool active = o < (int64_t)n_o;
    if (__ballot(active) == 0ull) return;  // whole wave idle (k_cloud_bbox_fold skips its slot by the same test)
    uint4 res = make_uint4(0, 0, 0, 0);
    if (active) {
        const uint4* src = reinterpret_cast<const uint4*>(in + (int64_t)f * in_fstride);
        if (passthrough) {
            res = src[o];
        } else if (g.overflow) {  // output = input; the caller's z += 500 / z -= 500 still happen around it
            res = src[o];
            res.z = __float_as_uint((__uint_as_float(res.z) + z_offset) - z_offset);
        } else {
            const uint32_t* ss = seg_start + (int64_t)f * (cap + 1);
            const uint32_t* pid = sorted_buf(g, vals0, vals1) + (int64_t)f * cap;
            const uint32_t v = keep_idx ? keep_idx[(int64_t)f * cap + o] : (uint32_t)o;
            const uint32_t b = ss[v], e = ss[v + 1];
            float sx = 0.f, sy = 0.f, sz = 0.f, sr = 0.f, sg = 0.f, sb = 0.f, sa = 0.f;
            // the sums are strictly sequential (input order); the loads are not: 8 gathers in flight
            for (uint32_t li = b; li < e; li += 8) {
                uint32_t id[8];
                uint4 p[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) id[k] = (li + k < e) ? pid[li + k] : 0xffffffffu;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (id[k] != 0xffffffffu) p[k] = src[id[k]];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (id[k] != 0xffffffffu) {
                        sx += __uint_as_float(p[k].x);
                        sy += __uint_as_float(p[k].y);
                        sz += __uint_as_float(p[k].z) + z_offset;
                        sr += (float)((p[k].w >> 16) & 255u);
                        sg += (float)((p[k].w >> 8) & 255u);
                        sb += (float)(p[k].w & 255u);
                        sa += (float)(p[k].w >> 24);
                    }
                }
            }
            const float nf = (float)(e - b);
            const float cx = sx / nf, cy = sy / nf, cz = sz / nf - z_offset;
            const uint32_t rgba = ((uint32_t)(sa / nf) << 24) | ((uint32_t)(sr / nf) << 16) | ((uint32_t)(sg / nf) << 8) |
                                  (uint32_t)(sb / nf);
            res = make_uint4(__float_as_uint(cx), __float_as_uint(cy), __float_as_uint(cz), rgba);
        }
        reinterpret_cast<uint4*>(out_base + out_off[f])[o] = res;
    }
    if (out_mm) {  // bounding box of what this wave appended (folded into cloud_big's box by k_cloud_bbox_fold)
        float* slot = out_mm + (((int64_t)f * nbx + bx) * (kPtThreads / 64) + (threadIdx.x >> 6)) * 6;
        const float c3[3] = {__uint_as_float(res.x), __uint_as_float(res.y), __uint_as_float(res.z)};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float l = wave_min_f32(active ? c3[a] : __builtin_inff());
            const float h = wave_max_f32(active ? c3[a] : -__builtin_inff());
            if ((threadIdx.x & 63) == 0) {
                slot[a] = l;
                slot[3 + a] = h;
            }
        }
    }
}

// kStride: the grid is smaller than the outputs need and workgroups loop (used where the kernel is launched only in
// case a cloud took the other sort variant, so that a launch that finds nothing to do costs a few thousand workgroups)
template <bool kStride>
__global__ __launch_bounds__(kPtThreads) void k_centroid(const o3dr_point* __restrict__ in, int64_t in_fstride,
                                                         const uint32_t* __restrict__ vals0,
                                                         const uint32_t* __restrict__ vals1, int64_t cap,
                                                         const uint32_t* __restrict__ seg_start,
                                                         const uint32_t* __restrict__ keep_idx,
                                                         const VoxelGeom* __restrict__ geom,
                                                         const uint32_t* __restrict__ n_out,
                                                         const uint64_t* __restrict__ out_off, float z_offset,
                                                         int passthrough, o3dr_point* __restrict__ out_base,
                                                         float* __restrict__ out_mm, int nbx)
{
    const int f = blockIdx.y;
    const VoxelGeom g = geom[f];
    // g.n == 0: frame taken by the pixel-window path; val_bits != 0: records are runs (k_centroid_runs writes it)
    if (!passthrough && (g.n == 0 || g.val_bits != 0u)) return;
    const uint32_t n_o = n_out[f];
    if (kStride) {
        for (int64_t bx = blockIdx.x; bx * kPtThreads < (int64_t)n_o; bx += gridDim.x)
            centroid_block(bx, f, g, n_o, in, in_fstride, vals0, vals1, cap, seg_start, keep_idx, out_off, z_offset,
                           passthrough, out_base, out_mm, nbx);
    } else {
        centroid_block(blockIdx.x, f, g, n_o, in, in_fstride, vals0, vals1, cap, seg_start, keep_idx, out_off, z_offset,
                       passthrough, out_base, out_mm, nbx);
    }
}

// running bounding box of cloud_big: fold the slots k_centroid wrote for a batch (only workgroups that had output)
__global__ __launch_bounds__(256) void k_cloud_bbox_fold(const float* __restrict__ slots, int nbx, int frames,
                                                         const uint32_t* __restrict__ n_out, float* __restrict__ partial)
{
    __shared__ float mm_lds[6 * 4];
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    const int64_t total = (int64_t)nbx * frames;  // nbx = wave slots per frame
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int f = (int)(i / nbx), bx = (int)(i - (int64_t)f * nbx);
        if ((int64_t)bx * 64 < (int64_t)n_out[f]) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = fminf(lo[a], slots[i * 6 + a]);
                hi[a] = fmaxf(hi[a], slots[i * 6 + 3 + a]);
            }
        }
    }
    block_minmax_store<4>(lo, hi, true, mm_lds, partial + (int64_t)blockIdx.x * 6);
}
__global__ __launch_bounds__(384) void k_cloud_bbox_merge(const float* __restrict__ partial, int n_partial, float* __restrict__ box6)
{
    const int a = threadIdx.x >> 6, lane = threadIdx.x & 63;  // one wave per component (min x,y,z, max x,y,z)
    float v = a < 3 ? __builtin_inff() : -__builtin_inff();
    for (int i = lane; i < n_partial; i += 64) v = a < 3 ? fminf(v, partial[i * 6 + a]) : fmaxf(v, partial[i * 6 + a]);
    v = a < 3 ? wave_min_f32(v) : wave_max_f32(v);
    if (lane == 0) box6[a] = a < 3 ? fminf(box6[a], v) : fmaxf(box6[a], v);
}

// =================================================================================================
// Run-compressed voxel grid (used for whole-cloud calls: the combined merge, o3dr_voxel_grid).
// Inputs of those calls are mostly concatenations of clouds that are already in voxel order, so
// consecutive points very often share a voxel.  A maximal block of consecutive points with the same
// index is a RUN; sorting runs (index, run id) instead of points is the same stable order with several
// times fewer records, and a voxel's points are then read as a few contiguous blocks instead of one
// gather per point.  Results are bit-identical to the per-point path.
// =================================================================================================
__global__ void k_run_geom(const VoxelGeom* __restrict__ geom, const uint32_t* __restrict__ n_runs, int frames,
                           VoxelGeom* __restrict__ geom_runs, int force_runs)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    VoxelGeom g = geom[f];
    // Sorting runs pays off when they are long enough (a concatenation of clouds already in voxel order: ~3 points
    // per run); on raw pixel-order points (~1.4 per run) the extra passes cost more than they save.  Decided here,
    // on the device: val_bits != 0 marks "records are runs", everything downstream keys off it.
    const uint32_t nr = n_runs[f];
    if (g.overflow || (uint64_t)nr * 2u <= (uint64_t)g.n || force_runs) {
        g.val_bits = run_start_bits(g.n);  // payload = (first point, length) of the run
        g.n = g.overflow ? 0u : nr;
        g.buf0 = 1;  // the run keys are gathered into buffer 1
    }
    geom_runs[f] = g;
}
// lengths of the runs in sorted order (then scanned in place): points of voxel v = pref[end] - pref[start]
__global__ __launch_bounds__(256) void k_run_lengths(const uint32_t* __restrict__ ids0, const uint32_t* __restrict__ ids1,
                                                     const uint32_t* __restrict__ run_start, int64_t cap,
                                                     const VoxelGeom* __restrict__ geom_runs, uint32_t* __restrict__ len_out)
{
    const int f = blockIdx.y;
    const VoxelGeom g = geom_runs[f];
    if (g.overflow || g.val_bits == 0u) return;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j > g.n) return;
    uint32_t* out = len_out + (int64_t)f * (cap + 1);
    if (j == g.n) {  // one extra slot so that the exclusive scan yields the grand total at index n
        out[j] = 0;
        return;
    }
    uint32_t first, len;
    run_unpack((sorted_buf(g, ids0, ids1) + (int64_t)f * cap)[j], g.val_bits, first, len);
    out[j] = len;
}
__global__ __launch_bounds__(256) void k_keep_count_runs(const uint32_t* __restrict__ seg_start,
                                                         const uint32_t* __restrict__ len_pref, int64_t cap,
                                                         const VoxelGeom* __restrict__ geom_runs,
                                                         const uint32_t* __restrict__ n_vox, uint32_t min_points,
                                                         int n_tiles, uint32_t* __restrict__ seg_cnt)
{
    __shared__ uint32_t lds[4];
    const int f = blockIdx.y, tile = blockIdx.x;
    const VoxelGeom g = geom_runs[f];
    if (g.overflow || g.val_bits == 0u) return;
    const uint32_t nv = n_vox[f];
    const uint32_t* ss = seg_start + (int64_t)f * (cap + 1);
    const uint32_t* lp = len_pref + (int64_t)f * (cap + 1);
    uint32_t c = 0;
    const int64_t base = (int64_t)tile * kSegTile;
    if (base < nv) {
        for (int j = 0; j < kSegTile / 256; ++j) {
            const int64_t o = base + j * 256 + threadIdx.x;
            if (o < nv) c += (lp[ss[o + 1]] - lp[ss[o]] >= min_points) ? 1u : 0u;
        }
    }
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) seg_cnt[(int64_t)f * n_tiles + tile] = lds[0] + lds[1] + lds[2] + lds[3];
}
__global__ __launch_bounds__(256) void k_keep_write_runs(const uint32_t* __restrict__ seg_start,
                                                         const uint32_t* __restrict__ len_pref, int64_t cap,
                                                         const VoxelGeom* __restrict__ geom_runs,
                                                         const uint32_t* __restrict__ n_vox, uint32_t min_points,
                                                         int n_tiles, const uint32_t* __restrict__ seg_off,
                                                         uint32_t* __restrict__ keep_idx)
{
    __shared__ uint32_t scan_lds[5];
    const int f = blockIdx.y, tile = blockIdx.x;
    const VoxelGeom g = geom_runs[f];
    if (g.overflow || g.val_bits == 0u) return;
    const uint32_t nv = n_vox[f];
    const int64_t base = (int64_t)tile * kSegTile;
    if (base >= nv) return;
    const uint32_t* ss = seg_start + (int64_t)f * (cap + 1);
    const uint32_t* lp = len_pref + (int64_t)f * (cap + 1);
    uint32_t* ki = keep_idx + (int64_t)f * cap;
    uint32_t off = seg_off[(int64_t)f * n_tiles + tile];
    for (int j = 0; j < kSegTile / 256; ++j) {
        const int64_t o = base + j * 256 + threadIdx.x;
        const bool keep = (o < nv) && (lp[ss[o + 1]] - lp[ss[o]] >= min_points);
        uint32_t total;
        const uint32_t pos = block_excl_scan_u32<4>(keep ? 1u : 0u, scan_lds, total);
        if (keep) ki[off + pos] = (uint32_t)o;
        off += total;
    }
}

// centroid of voxel o: its runs in sorted (= input) order, each run a contiguous block of points
__global__ __launch_bounds__(kPtThreads) void k_centroid_runs(const o3dr_point* __restrict__ in, int64_t in_fstride,
                                                              const uint32_t* __restrict__ ids0,
                                                              const uint32_t* __restrict__ ids1, int64_t cap,
                                                              const uint32_t* __restrict__ seg_start,
                                                              const uint32_t* __restrict__ run_start,
                                                              const uint32_t* __restrict__ keep_idx,
                                                              const VoxelGeom* __restrict__ geom_runs,
                                                              const uint32_t* __restrict__ n_out,
                                                              const uint64_t* __restrict__ out_off, float z_offset,
                                                              o3dr_point* __restrict__ out_base)
{
    const int f = blockIdx.y;
    const VoxelGeom g = geom_runs[f];
    if (g.val_bits == 0u) return;  // this cloud sorted its points: k_centroid writes it
    const uint4* src = reinterpret_cast<const uint4*>(in + (int64_t)f * in_fstride);
    uint4* dst = reinterpret_cast<uint4*>(out_base + out_off[f]);
    const int64_t o = (int64_t)blockIdx.x * kPtThreads + threadIdx.x;
    if (o >= n_out[f]) return;
    if (g.overflow) {  // output = input; the caller's z += 500 / z -= 500 still happen around it
        uint4 v = src[o];
        v.z = __float_as_uint((__uint_as_float(v.z) + z_offset) - z_offset);
        dst[o] = v;
        return;
    }
    const uint32_t* ss = seg_start + (int64_t)f * (cap + 1);
    const uint32_t* rv = sorted_buf(g, ids0, ids1) + (int64_t)f * cap;  // sorted payloads: (first point, length)
    const uint32_t v = keep_idx ? keep_idx[(int64_t)f * cap + o] : (uint32_t)o;
    const uint32_t jb = ss[v], je = ss[v + 1];
    const uint32_t bits = g.val_bits;
    float sx = 0.f, sy = 0.f, sz = 0.f, sr = 0.f, sg = 0.f, sb = 0.f, sa = 0.f;
    uint32_t n_pts = 0;
    // The sums are strictly sequential; the loads are not: the next 4 payloads and the first 4 points of each of
    // this group's 4 runs are in flight before the first add (runs are short, a few points each).
    uint32_t nv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) nv[k] = (jb + k < je) ? rv[jb + k] : 0u;
    for (uint32_t j = jb; j < je; j += 4) {
        uint32_t b[4], e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t first, len;
            run_unpack(nv[k], bits, first, len);
            b[k] = first;
            e[k] = (j + k < je) ? first + len : first;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) nv[k] = (j + 4 + k < je) ? rv[j + 4 + k] : 0u;
        uint4 p[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (b[k] + q < e[k]) p[k][q] = src[b[k] + q];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (b[k] + q < e[k]) {
                    sx += __uint_as_float(p[k][q].x);
                    sy += __uint_as_float(p[k][q].y);
                    sz += __uint_as_float(p[k][q].z) + z_offset;
                    sr += (float)((p[k][q].w >> 16) & 255u);
                    sg += (float)((p[k][q].w >> 8) & 255u);
                    sb += (float)(p[k][q].w & 255u);
                    sa += (float)(p[k][q].w >> 24);
                }
            }
            for (uint32_t i = b[k] + 4; i < e[k]; i += 4) {  // the rest of a long run
                uint4 t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (i + q < e[k]) t[q] = src[i + q];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (i + q < e[k]) {
                        sx += __uint_as_float(t[q].x);
                        sy += __uint_as_float(t[q].y);
                        sz += __uint_as_float(t[q].z) + z_offset;
                        sr += (float)((t[q].w >> 16) & 255u);
                        sg += (float)((t[q].w >> 8) & 255u);
                        sb += (float)(t[q].w & 255u);
                        sa += (float)(t[q].w >> 24);
                    }
                }
            }
            n_pts += e[k] - b[k];
        }
    }
    const float nf = (float)n_pts;
    const float cx = sx / nf, cy = sy / nf, cz = sz / nf - z_offset;
    const uint32_t rgba = ((uint32_t)(sa / nf) << 24) | ((uint32_t)(sr / nf) << 16) | ((uint32_t)(sg / nf) << 8) |
                          (uint32_t)(sb / nf);
    dst[o] = make_uint4(__float_as_uint(cx), __float_as_uint(cy), __float_as_uint(cz), rgba);
}

// =================================================================================================
// Multi-GPU merge support: the combined voxel grid is laid over the GLOBAL
// bounding box of all ranks' clouds; its linear index range is cut into n_parts contiguous slices
// and every point goes to the rank owning its slice.  Stable, so a slice's points stay in global
// (rank, frame, index) order and the merged cells are bit-identical to a single-GPU run.
// =================================================================================================
__global__ __launch_bounds__(256) void k_bbox_fold(const float* __restrict__ mm, int used, float* __restrict__ out6)
{
    __shared__ float red[6 * 4];
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int sidx = threadIdx.x; sidx < used; sidx += 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], mm[sidx * 6 + a]);
            hi[a] = fmaxf(hi[a], mm[sidx * 6 + 3 + a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = wave_min_f32(lo[a]), h = wave_max_f32(hi[a]);
        if ((threadIdx.x & 63) == 0) {
            red[(threadIdx.x >> 6) * 6 + a] = l;
            red[(threadIdx.x >> 6) * 6 + 3 + a] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        out6[a] = fminf(fminf(red[a], red[6 + a]), fminf(red[12 + a], red[18 + a]));
        out6[3 + a] = fmaxf(fmaxf(red[3 + a], red[9 + a]), fmaxf(red[15 + a], red[21 + a]));
    }
}

// linear voxel index -> owning part: slice p covers indices [p*cells/n_parts, (p+1)*cells/n_parts)
__global__ __launch_bounds__(256) void k_part_ids(uint32_t* __restrict__ keys, const VoxelGeom* __restrict__ geom,
                                                  int n_parts)
{
    const VoxelGeom g = geom[0];
    if (g.overflow) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.n) return;
    const uint64_t cells = (uint64_t)(uint32_t)g.div_b[0] * (uint64_t)(uint32_t)g.div_b[1] * (uint64_t)(uint32_t)g.div_b[2];
    uint64_t part = (uint64_t)keys[i] * (uint64_t)n_parts / cells;
    if (part >= (uint64_t)n_parts) part = n_parts - 1;
    keys[i] = (uint32_t)part;
}
__global__ void k_part_plan(VoxelGeom* geom, int n_parts)
{
    if (geom[0].overflow) return;
    uint32_t bits = 1;
    while ((1u << bits) < (uint32_t)n_parts) ++bits;
    geom[0].passes = 1;
    geom[0].bpp = bits;
}
__global__ __launch_bounds__(256) void k_gather_points(const o3dr_point* __restrict__ in, const uint32_t* __restrict__ ids,
                                                       const VoxelGeom* __restrict__ geom, o3dr_point* __restrict__ out)
{
    const VoxelGeom g = geom[0];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.n) return;
    const uint4* src = reinterpret_cast<const uint4*>(in);
    reinterpret_cast<uint4*>(out)[i] = g.overflow ? src[i] : src[ids[i]];
}
// counts[p] = records of part p in the sorted part-id array (binary search per part)
__global__ void k_part_counts(const uint32_t* __restrict__ sorted_parts, const VoxelGeom* __restrict__ geom, int n_parts,
                              uint64_t* __restrict__ counts, uint32_t* __restrict__ overflow)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const VoxelGeom g = geom[0];
    if (p == 0) *overflow = g.overflow;
    if (p >= n_parts) return;
    if (g.overflow) {
        counts[p] = 0;
        return;
    }
    auto lower = [&](uint32_t v) {
        uint32_t lo = 0, hi = g.n;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (sorted_parts[mid] < v) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    counts[p] = (uint64_t)(lower((uint32_t)p + 1) - lower((uint32_t)p));
}

// =================================================================================================
// A3b — pcl::StatisticalOutlierRemoval<PointXYZRGB> (pose_functions.cpp:1673-1686: mean_k 50, 1 sigma)
//   [PCL 1.8 filters/impl/statistical_outlier_removal.hpp on KdTreeFLANN / flann::L2_Simple<float>]
//   d2 = ((0 + dx*dx) + dy*dy) + dz*dz in fp32; the 51 smallest d2 per point (the point itself
//   included), mean of sqrt over the 50 non-first ones in fp64, global mean/stddev in fp64, keep iff
//   !(dist > mean + 1*stddev).  Exact k-NN: a uniform XY grid (cells sorted with the radix sort above),
//   ring expansion, conservative stop (no unvisited column can hold a point closer than 0.999*r*h).
// =================================================================================================
__global__ __launch_bounds__(256) void k_sor_plan(const float* __restrict__ mm, int mm_used,
                                                  const uint32_t* __restrict__ n_dev, uint32_t max_cells,
                                                  SorGeom* __restrict__ sg, VoxelGeom* __restrict__ geom)
{
    __shared__ float red[6 * 4];
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int sidx = threadIdx.x; sidx < mm_used; sidx += 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], mm[sidx * 6 + a]);
            hi[a] = fmaxf(hi[a], mm[sidx * 6 + 3 + a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = wave_min_f32(lo[a]), h = wave_max_f32(hi[a]);
        if ((threadIdx.x & 63) == 0) {
            red[(threadIdx.x >> 6) * 6 + a] = l;
            red[(threadIdx.x >> 6) * 6 + 3 + a] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    SorGeom g;
    g.n = n_dev[0];
    g.active = g.n > (uint32_t)kSorMeanK ? 1u : 0u;  // fewer points: the reference reads past its list; pass through
    g.mnx = fminf(fminf(red[0], red[6]), fminf(red[12], red[18]));
    g.mny = fminf(fminf(red[1], red[7]), fminf(red[13], red[19]));
    const float mxx = fmaxf(fmaxf(red[3], red[9]), fmaxf(red[15], red[21]));
    const float mxy = fmaxf(fmaxf(red[4], red[10]), fmaxf(red[16], red[22]));
    double ex = (double)mxx - (double)g.mnx, ey = (double)mxy - (double)g.mny;
    if (!(ex > 1e-9)) ex = 1e-9;
    if (!(ey > 1e-9)) ey = 1e-9;
    double h = sqrt(8.0 * ex * ey / (double)(g.n ? g.n : 1u));  // ~8 points per column
    if (h < 1e-6) h = 1e-6;
    int64_t gx = (int64_t)(ex / h) + 1, gy = (int64_t)(ey / h) + 1;
    while (gx * gy > (int64_t)max_cells) {
        h *= 1.25;
        gx = (int64_t)(ex / h) + 1;
        gy = (int64_t)(ey / h) + 1;
    }
    g.h = (float)h;
    g.inv_h = (float)(1.0 / h);
    g.gx = (int)gx;
    g.gy = (int)gy;
    g.threshold = 0.0;
    *sg = g;
    // sort plan for the cell ids
    VoxelGeom v;
    for (int a = 0; a < 3; ++a) v.inv[a] = 1.f, v.min_b[a] = 0, v.div_b[a] = 1;
    v.mul1 = v.mul2 = 1;
    v.n = g.n;
    v.buf0 = 0;
    v.val_bits = 0;
    v.overflow = g.active ? 0u : 1u;  // inactive: every sort kernel returns at once
    const uint64_t cells = (uint64_t)gx * (uint64_t)gy;
    uint32_t nbits = cells > 1 ? 64u - (uint32_t)__clzll((long long)(cells - 1)) : 1u;
    v.passes = (nbits + kMaxRadixBits - 1) / kMaxRadixBits;
    v.bpp = (nbits + v.passes - 1) / v.passes;
    geom[0] = v;
}

__device__ __forceinline__ int sor_cell(const SorGeom& g, float x, float y, int& cx, int& cy)
{
    cx = (int)((x - g.mnx) * g.inv_h);
    cy = (int)((y - g.mny) * g.inv_h);
    cx = cx < 0 ? 0 : (cx >= g.gx ? g.gx - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= g.gy ? g.gy - 1 : cy);
    return cy * g.gx + cx;
}

__global__ __launch_bounds__(256) void k_sor_cells(const o3dr_point* __restrict__ in, const SorGeom* __restrict__ sg,
                                                   uint32_t* __restrict__ keys)
{
    const SorGeom g = *sg;
    if (!g.active) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.n) return;
    const uint4 v = reinterpret_cast<const uint4*>(in)[i];
    int cx, cy;
    keys[i] = (uint32_t)sor_cell(g, __uint_as_float(v.x), __uint_as_float(v.y), cx, cy);
}

// after the sort: coordinates in cell order (coalesced candidate reads) and [start,end) of every cell
__global__ __launch_bounds__(256) void k_sor_cell_table(const o3dr_point* __restrict__ in, const uint32_t* __restrict__ keys0,
                                                        const uint32_t* __restrict__ keys1, const uint32_t* __restrict__ ids0,
                                                        const uint32_t* __restrict__ ids1, const SorGeom* __restrict__ sg,
                                                        const VoxelGeom* __restrict__ geom, float4* __restrict__ sxyz,
                                                        uint32_t* __restrict__ cell_start, uint32_t* __restrict__ cell_end)
{
    const SorGeom g = *sg;
    if (!g.active) return;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= g.n) return;
    const VoxelGeom vg = geom[0];
    const uint32_t* k = sorted_buf(vg, keys0, keys1);
    const uint32_t* id = sorted_buf(vg, ids0, ids1);
    const uint4 v = reinterpret_cast<const uint4*>(in)[id[j]];
    sxyz[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), 0.f);
    const uint32_t c = k[j];
    if (j == 0 || k[j - 1] != c) cell_start[c] = (uint32_t)j;
    if (j + 1 == g.n || k[j + 1] != c) cell_end[c] = (uint32_t)j + 1u;
}

constexpr int kSorThreads = 128;
// The 51 smallest squared distances of a query live in a binary MAX-heap, one LDS column per lane
// (conflict-free): a candidate below the current maximum replaces the root and sifts down (<= 6 levels)
// instead of shifting an ordered list.  At the end the heap is sorted in place (heap sort), because the
// reference adds the distances in ascending order and the fp64 sum is order sensitive in its last bits.
__device__ __forceinline__ void sor_sift_down(float (*heap)[kSorThreads], int t, int n, int i, float v)
{
    for (;;) {
        int c = 2 * i + 1;
        if (c >= n) break;
        float cv = heap[c][t];
        if (c + 1 < n) {
            const float rv = heap[c + 1][t];
            if (rv > cv) {
                cv = rv;
                ++c;
            }
        }
        if (!(cv > v)) break;
        heap[i][t] = cv;
        i = c;
    }
    heap[i][t] = v;
}

// Queries are taken in CELL order (thread j = j-th point of the cell-sorted array): the lanes of a wave sit
// in the same or adjacent columns, walk the same rings and read the same candidates (one broadcast load
// per candidate instead of 64 scattered ones); the result goes back to the point's original index.
__global__ __launch_bounds__(kSorThreads) void k_sor_knn(const float4* __restrict__ sxyz, const uint32_t* __restrict__ ids0,
                                                         const uint32_t* __restrict__ ids1,
                                                         const VoxelGeom* __restrict__ geom,
                                                         const uint32_t* __restrict__ cell_start,
                                                         const uint32_t* __restrict__ cell_end,
                                                         const SorGeom* __restrict__ sg, float* __restrict__ dist)
{
    constexpr int K = kSorMeanK + 1;
    __shared__ float heap[K][kSorThreads];
    const SorGeom g = *sg;
    if (!g.active) return;
    const int64_t i = (int64_t)blockIdx.x * kSorThreads + threadIdx.x;
    if (i >= g.n) return;
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k) heap[k][t] = __builtin_huge_valf();  // all-equal values form a valid heap
    const float4 qv = sxyz[i];
    const float qx = qv.x, qy = qv.y, qz = qv.z;
    const uint32_t out_index = sorted_buf(geom[0], ids0, ids1)[i];
    int cx, cy;
    sor_cell(g, qx, qy, cx, cy);
    float worst = __builtin_huge_valf();  // heap root = 51st smallest so far
    const int rmax = g.gx > g.gy ? g.gx : g.gy;
    for (int r = 0; r <= rmax; ++r) {
        for (int yy = cy - r; yy <= cy + r; ++yy) {
            if (yy < 0 || yy >= g.gy) continue;
            const bool edge_row = (yy == cy - r) || (yy == cy + r);
            const int step = edge_row ? 1 : (2 * r > 0 ? 2 * r : 1);
            for (int xx = cx - r; xx <= cx + r; xx += step) {
                if (xx < 0 || xx >= g.gx) continue;
                const int c = yy * g.gx + xx;
                const uint32_t s1 = cell_end[c];
                for (uint32_t sidx = cell_start[c]; sidx < s1; ++sidx) {
                    const float4 p = sxyz[sidx];
                    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
                    const float d = ((0.0f + dx * dx) + dy * dy) + dz * dz;
                    if (d < worst) {  // replaces the current maximum
                        sor_sift_down(heap, t, K, 0, d);
                        worst = heap[0][t];
                    }
                }
            }
        }
        const double bound = 0.999 * (double)r * (double)g.h;
        if (worst < __builtin_huge_valf() && (double)worst <= bound * bound) break;
    }
    // heap sort: ascending order in place
    for (int n = K - 1; n > 0; --n) {
        const float top = heap[0][t];
        const float last = heap[n][t];
        heap[n][t] = top;
        sor_sift_down(heap, t, n, 0, last);
    }
    double dist_sum = 0.0;
    for (int k = 1; k < K; ++k) dist_sum += sqrt((double)heap[k][t]);
    dist[out_index] = (float)(dist_sum / (double)kSorMeanK);
}

// sum and sum of squares (float product like PCL, fp64 sums): fixed-shape two-level reduction
__global__ __launch_bounds__(256) void k_sor_partial(const float* __restrict__ dist, const SorGeom* __restrict__ sg,
                                                     double* __restrict__ partial /*[blocks][2]*/)
{
    __shared__ double red[2 * 4];
    const SorGeom g = *sg;
    double s = 0.0, q = 0.0;
    if (g.active) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < g.n; i += (int64_t)gridDim.x * 256) {
            const float d = dist[i];
            s += (double)d;
            q += (double)(d * d);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[(threadIdx.x >> 6) * 2] = s;
        red[(threadIdx.x >> 6) * 2 + 1] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (red[0] + red[2]) + (red[4] + red[6]);
        partial[2 * blockIdx.x + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}
__global__ void k_sor_threshold(const double* __restrict__ partial, int blocks, double stddev_mul, SorGeom* __restrict__ sg)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double sum = 0.0, sq = 0.0;
    for (int b = 0; b < blocks; ++b) {
        sum += partial[2 * b];
        sq += partial[2 * b + 1];
    }
    const double n = (double)sg->n;
    const double mean = sum / n;
    const double variance = (sq - sum * sum / n) / (n - 1.0);
    sg->threshold = mean + stddev_mul * sqrt(variance);
}

// ordered compaction of the inliers (count per 1024-point tile, scan, emit) + their bounding boxes
__global__ __launch_bounds__(256) void k_sor_count(const float* __restrict__ dist, const SorGeom* __restrict__ sg,
                                                   uint32_t* __restrict__ tile_cnt)
{
    __shared__ uint32_t lds[4];
    const SorGeom g = *sg;
    uint32_t c = 0;
    const int64_t base = (int64_t)blockIdx.x * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < g.n) c += (!g.active || !((double)dist[i] > g.threshold)) ? 1u : 0u;
    }
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}
__global__ __launch_bounds__(256) void k_sor_emit(const o3dr_point* __restrict__ in, const float* __restrict__ dist,
                                                  const SorGeom* __restrict__ sg, const uint32_t* __restrict__ tile_off,
                                                  o3dr_point* __restrict__ out, int64_t mm_stride, float* __restrict__ mm)
{
    __shared__ uint32_t scan_lds[5];
    __shared__ float mm_lds[6 * 4];
    const SorGeom g = *sg;
    const int64_t base = (int64_t)blockIdx.x * 1024;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    bool any = false;
    uint32_t off = tile_off[blockIdx.x];
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const bool keep = (i < g.n) && (!g.active || !((double)dist[i] > g.threshold));
        uint32_t total;
        const uint32_t pos = block_excl_scan_u32<4>(keep ? 1u : 0u, scan_lds, total);
        if (keep) {
            const uint4 v = reinterpret_cast<const uint4*>(in)[i];
            reinterpret_cast<uint4*>(out)[off + pos] = v;
            const float x = __uint_as_float(v.x), y = __uint_as_float(v.y), z = __uint_as_float(v.z);
            lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
            lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
            lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
            any = true;
        }
        off += total;
    }
    block_minmax_store<4>(lo, hi, any, mm_lds, mm + (int64_t)blockIdx.x * 6);
}

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
            k_reproject_emit<true><<<grid, kEmitThreads, 0, s>>>(a, out, tile_cnt, n_kp, mm, nullptr);
        else
            k_reproject_emit<false><<<grid, kEmitThreads, 0, s>>>(a, out, tile_cnt, n_kp, mm, nullptr);
    }
}

void launch_frame_bbox(Profiler* pf, hipStream_t s, const ReprojectArgs& a, int frames, uint32_t* tile_cnt,
                       const uint32_t* n_kp, uint32_t* n_valid, float* mm, uint32_t* scan_partial)
{
    const dim3 grid(a.n_tiles, frames);
    {
        ProfScope ps(pf, O3DR_K_COUNT, s);
        k_frame_bbox<<<grid, kEmitThreads, 0, s>>>(a, tile_cnt, mm);
    }
    {
        ProfScope ps(pf, O3DR_K_OTHER, s);
        launch_scan(s, tile_cnt, a.n_tiles, a.n_tiles, frames, n_valid, n_kp, scan_partial);
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
    {
        ProfScope ps(pf, O3DR_K_OTHER, s);
        k_voxel_geom<<<F, 256, 0, s>>>(ws.mm, ws.mm_stride, v.mm_used, v.n_dev, v.leaf[0], v.leaf[1], v.leaf[2],
                                       v.z_offset, ws.geom);
    }
    uint32_t* n_keep = nullptr;
    // run compression: whole-cloud calls only, and not together with the look-back variant
    const bool use_runs = v.use_runs && !ws.single_pass && !v.passthrough;
    const WindowPlan* w = (v.window && !use_runs && !ws.single_pass && !v.passthrough && cap > 0) ? v.window : nullptr;
    // what the sort and the run/cell kernels count (runs / voxel records of window frames / points)
    const VoxelGeom* sort_geom = (use_runs || w) ? ws.geom_runs : ws.geom;
    // what the per-point kernels of the sort-based path see: frames on the window path have n = 0 there
    const VoxelGeom* gen_geom = w ? ws.geom_gen : ws.geom;
    const VoxelGeom* seg_geom = w ? ws.geom_gen : sort_geom;
    // plain point sort (per-frame grids of a batch): the index kernel also counts the first pass's digits
    const bool fuse_hist0 = !use_runs && !w && !ws.single_pass && !getenv("O3DR_NO_FUSE_HIST0");
    if (w) {
        {
            ProfScope ps(pf, O3DR_K_OTHER, s);
            k_window_plan<<<cdiv64(F, 64), 64, 0, s>>>(ws.geom, w->a.poses, F, w->rho_max, w->err_budget, ws.geom_gen, ws.n_runs,
                                                       ws.win_c);
        }
        {
            ProfScope ps(pf, O3DR_K_REPROJECT, s);
            k_reproject_emit<false><<<dim3(w->a.n_tiles, F), kEmitThreads, 0, s>>>(w->a, const_cast<o3dr_point*>(v.in), ws.tile_cnt,
                                                                             w->n_kp, ws.mm, ws.geom_gen);
        }
        {
            ProfScope ps(pf, O3DR_K_WINDOW, s);
            const int tiles_x = cdiv64(w->a.Nx, kWinTX), tiles_y = cdiv64(w->a.Ny, kWinTY);
            k_window_group<<<dim3(tiles_y, F), 256, 0, s>>>(w->a, w->wbase, ws.win_c, tiles_x, ws.geom, ws.geom_gen, cap,
                                                                      ws.keys[1], const_cast<o3dr_point*>(v.in), ws.n_runs);
        }
        {
            ProfScope ps(pf, O3DR_K_OTHER, s);
            k_window_sort_geom<<<cdiv64(F, 64), 64, 0, s>>>(ws.geom, ws.geom_gen, ws.n_runs, F, ws.geom_runs);
        }
    }
    if (!v.passthrough && cap > 0) {
        const dim3 grid(n_sort_tiles, F);
        if (ws.single_pass) {
            // keys + all digit histograms in one read of the points; then one look-back scatter per pass
            int nblk = 2048 / F;
            if (nblk > 256) nblk = 256;
            const int max_blk = cdiv64(cap, kKeyThreads * 4);
            if (nblk > max_blk) nblk = max_blk;
            if (nblk < 1) nblk = 1;
            {
                ProfScope ps(pf, O3DR_K_KEYGEN, s);
                k_voxel_keys_hist<<<dim3(nblk, F), kKeyThreads, 0, s>>>(v.in, v.in_fstride, ws.geom, v.z_offset, cap,
                                                                       ws.keys[0], nblk, ws.partial_hist);
            }
            {
                ProfScope ps(pf, O3DR_K_OTHER, s);
                k_digit_starts<<<F, kMaxRadix, 0, s>>>(ws.partial_hist, nblk, ws.geom, ws.digit_start, ws.tickets);
                if (ws.epoch > 0x3ffff0u - 8u) {  // epoch field about to wrap: clear the words, restart
                    (void)hipMemsetAsync(ws.lb_state, 0, ws.lb_bytes, s);
                    ws.epoch = 0;
                }
            }
            for (int pass = 0; pass < kMaxPasses; ++pass) {
                ProfScope ps(pf, O3DR_K_SORT_SCATTER, s);
                k_radix_scatter<true><<<grid, kSortThreads, 0, s>>>(ws.keys[0], ws.vals[0], ws.keys[1], ws.vals[1], cap,
                                                                   ws.geom, pass, n_sort_tiles, nullptr, ws.digit_start,
                                                                   ws.lb_state, ws.tickets, ++ws.epoch, ws.error_flag, nullptr);
            }
        } else {
            {
                ProfScope ps(pf, O3DR_K_KEYGEN, s);
                if (use_runs)  // indices and, in the same read, how many runs of equal indices start in every tile
                    k_voxel_keys_heads<<<dim3(n_seg_tiles, F), 256, 0, s>>>(v.in, v.in_fstride, ws.geom, v.z_offset, cap,
                                                                           ws.keys[0], n_seg_tiles, ws.seg_cnt);
                else if (fuse_hist0)  // ... and the histogram of the first radix pass
                    k_voxel_keys_hist0<<<grid, kSortThreads, 0, s>>>(v.in, v.in_fstride, gen_geom, v.z_offset, cap, ws.keys[0],
                                                                     n_sort_tiles, ws.hist);
                else
                    k_voxel_keys<<<dim3(cdiv64(cap, kPtThreads * 4), F), kPtThreads, 0, s>>>(v.in, v.in_fstride, gen_geom,
                                                                                            v.z_offset, cap, ws.keys[0]);
            }
            if (use_runs) {
                // runs of consecutive equal indices -> (run key, run id) records in buffer 1
                const dim3 rgrid(n_seg_tiles, F);
                {
                    ProfScope ps(pf, O3DR_K_OTHER, s);
                    launch_scan(s, ws.seg_cnt, n_seg_tiles, n_seg_tiles, F, ws.n_runs, nullptr, ws.scan_partial);
                }
                {
                    ProfScope ps(pf, O3DR_K_SEGMENT, s);
                    // runs or points?  (decided per cloud on the device; O3DR_RUNS=2 forces runs)
                    k_run_geom<<<cdiv64(F, 64), 64, 0, s>>>(ws.geom, ws.n_runs, F, ws.geom_runs, v.use_runs > 1 ? 1 : 0);
                    k_run_starts<<<rgrid, 256, 0, s>>>(ws.keys[0], ws.keys[1], cap, ws.geom, n_seg_tiles, ws.seg_cnt, ws.n_runs,
                                                      ws.run_start, 0, ws.keys[1], ws.geom_runs);  // run keys -> buffer 1
                }
            }
            // always kMaxPasses launch groups; frames whose index needs fewer passes drop out on the device
            const int64_t hist_row = (int64_t)kMaxRadix * n_sort_tiles;
            for (int pass = 0; pass < kMaxPasses; ++pass) {
                if (!(pass == 0 && fuse_hist0)) {
                    ProfScope ps(pf, O3DR_K_SORT_HIST, s);
                    k_radix_hist<<<grid, kSortThreads, 0, s>>>(ws.keys[0], ws.keys[1], cap, sort_geom, pass, n_sort_tiles,
                                                              ws.hist);
                }
                {
                    ProfScope ps(pf, O3DR_K_OTHER, s);
                    launch_scan(s, ws.hist, hist_row, hist_row, F, nullptr, nullptr, ws.scan_partial, sort_geom, pass,
                                n_sort_tiles);
                }
                {
                    ProfScope ps(pf, O3DR_K_SORT_SCATTER, s);
                    if (ws.scatter_ballot)
                        k_radix_scatter<false><<<grid, kSortThreads, 0, s>>>(ws.keys[0], ws.vals[0], ws.keys[1], ws.vals[1], cap,
                                                                            sort_geom, pass, n_sort_tiles, ws.hist, nullptr, nullptr,
                                                                            nullptr, 0u, nullptr, ws.run_start);
                    else
                        k_radix_scatter_lane<<<grid, kSortThreads, 0, s>>>(ws.keys[0], ws.vals[0], ws.keys[1], ws.vals[1], cap,
                                                                          sort_geom, pass, n_sort_tiles, ws.hist, ws.run_start);
                }
            }
        }
        const dim3 sgrid(n_seg_tiles, F);
        {
            ProfScope ps(pf, O3DR_K_SEGMENT, s);
            k_run_heads<<<sgrid, 256, 0, s>>>(ws.keys[0], ws.keys[1], cap, seg_geom, n_seg_tiles, ws.seg_cnt, -1);
        }
        {
            ProfScope ps(pf, O3DR_K_OTHER, s);
            launch_scan(s, ws.seg_cnt, n_seg_tiles, n_seg_tiles, F, ws.n_vox, nullptr, ws.scan_partial);
        }
        {
            ProfScope ps(pf, O3DR_K_SEGMENT, s);
            k_run_starts<<<sgrid, 256, 0, s>>>(ws.keys[0], ws.keys[1], cap, seg_geom, n_seg_tiles, ws.seg_cnt, ws.n_vox,
                                              ws.seg_start, -1, nullptr, nullptr);
        }
        if (v.min_points > 1) {
            {
                ProfScope ps(pf, O3DR_K_SEGMENT, s);
                if (use_runs) {
                    // points per voxel from a prefix sum over the sorted runs' lengths (one pass + one scan instead of
                    // walking every voxel's runs twice); clouds that sort points skip these on the device
                    k_run_lengths<<<dim3(cdiv64(cap + 1, 256), F), 256, 0, s>>>(ws.vals[0], ws.vals[1], ws.run_start, cap,
                                                                             ws.geom_runs, ws.run_len);
                    launch_scan(s, ws.run_len, cap + 1, cap + 1, F, nullptr, nullptr, ws.scan_partial, ws.geom_runs, -1, 0);
                    k_keep_count_runs<<<sgrid, 256, 0, s>>>(ws.seg_start, ws.run_len, cap, ws.geom_runs, ws.n_vox, v.min_points,
                                                           n_seg_tiles, ws.seg_cnt);
                }
                k_keep_count<<<sgrid, 256, 0, s>>>(ws.seg_start, cap, seg_geom, ws.n_vox, v.min_points, n_seg_tiles, ws.seg_cnt);
            }
            {
                ProfScope ps(pf, O3DR_K_OTHER, s);
                launch_scan(s, ws.seg_cnt, n_seg_tiles, n_seg_tiles, F, ws.n_out, nullptr, ws.scan_partial);
            }
            {
                ProfScope ps(pf, O3DR_K_SEGMENT, s);
                if (use_runs)
                    k_keep_write_runs<<<sgrid, 256, 0, s>>>(ws.seg_start, ws.run_len, cap, ws.geom_runs, ws.n_vox, v.min_points,
                                                           n_seg_tiles, ws.seg_cnt, ws.keep_idx);
                k_keep_write<<<sgrid, 256, 0, s>>>(ws.seg_start, cap, seg_geom, ws.n_vox, v.min_points, n_seg_tiles, ws.seg_cnt,
                                                  ws.keep_idx);
            }
            n_keep = ws.n_out;
        }
    }
    {
        ProfScope ps(pf, O3DR_K_OTHER, s);
        k_frame_offsets<<<1, 256, 0, s>>>(ws.geom, ws.n_vox, n_keep, F, v.passthrough, ws.n_out, ws.out_off, v.cc,
                                          v.stats, sort_geom);
    }
    const int nbx = cdiv64(cap, kPtThreads);
    if (cap > 0 && use_runs) {
        ProfScope ps(pf, O3DR_K_CENTROID_RUNS, s);
        k_centroid_runs<<<dim3(nbx, F), kPtThreads, 0, s>>>(
            v.in, v.in_fstride, ws.vals[0], ws.vals[1], cap, ws.seg_start, ws.run_start,
            v.min_points > 1 ? ws.keep_idx : nullptr, ws.geom_runs, ws.n_out, ws.out_off, v.z_offset, v.out_base);
    }
    if (cap > 0) {
        ProfScope ps(pf, O3DR_K_CENTROID, s);
        float* out_mm = (v.cloud_box && !w) ? ws.out_mm : nullptr;
        const uint32_t* keep = (v.min_points > 1 && !v.passthrough) ? ws.keep_idx : nullptr;
        if (use_runs)  // only for clouds k_run_geom left to the point sort: a small looping grid
            k_centroid<true><<<dim3(nbx < 4096 ? nbx : 4096, F), kPtThreads, 0, s>>>(
                v.in, v.in_fstride, ws.vals[0], ws.vals[1], cap, ws.seg_start, keep, ws.geom_runs, ws.n_out, ws.out_off,
                v.z_offset, v.passthrough, v.out_base, out_mm, nbx);
        else
            k_centroid<false><<<dim3(nbx, F), kPtThreads, 0, s>>>(
                v.in, v.in_fstride, ws.vals[0], ws.vals[1], cap, ws.seg_start, keep, gen_geom, ws.n_out, ws.out_off,
                v.z_offset, v.passthrough, v.out_base, out_mm, nbx);
        if (w)
            k_gather_heads<<<dim3(cdiv64(cap, kPtThreads), F), kPtThreads, 0, s>>>(v.in, cap, ws.vals[0], ws.vals[1], ws.geom_runs,
                                                                                 ws.n_out, ws.out_off, v.out_base);
    }
    if (v.cloud_box && cap > 0 && !use_runs && !w) {
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
void launch_partition(Profiler* pf, hipStream_t s, Workspace& ws, const VoxelArgs& v, int n_parts, o3dr_point* out,
                      uint64_t* counts_dev, uint32_t* overflow_dev)
{
    const int64_t cap = v.cap;
    const int n_sort_tiles = cdiv64(cap, kSortTile);
    const int64_t hist_row = (int64_t)kMaxRadix * n_sort_tiles;
    ProfScope ps(pf, O3DR_K_OTHER, s);
    k_voxel_geom<<<1, 256, 0, s>>>(ws.mm, ws.mm_stride, 1, v.n_dev, v.leaf[0], v.leaf[1], v.leaf[2], v.z_offset, ws.geom);
    k_voxel_keys<<<dim3(cdiv64(cap, kPtThreads * 4), 1), kPtThreads, 0, s>>>(v.in, 0, ws.geom, v.z_offset, cap, ws.keys[0]);
    k_part_ids<<<cdiv64(cap, 256), 256, 0, s>>>(ws.keys[0], ws.geom, n_parts);
    k_part_plan<<<1, 1, 0, s>>>(ws.geom, n_parts);
    k_radix_hist<<<dim3(n_sort_tiles, 1), kSortThreads, 0, s>>>(ws.keys[0], ws.keys[1], cap, ws.geom, 0, n_sort_tiles, ws.hist);
    launch_scan(s, ws.hist, hist_row, hist_row, 1, nullptr, nullptr, ws.scan_partial, ws.geom, 0, n_sort_tiles);
    k_radix_scatter<false><<<dim3(n_sort_tiles, 1), kSortThreads, 0, s>>>(ws.keys[0], ws.vals[0], ws.keys[1], ws.vals[1], cap,
                                                                         ws.geom, 0, n_sort_tiles, ws.hist, nullptr, nullptr, nullptr,
                                                                         0u, nullptr, nullptr);
    k_gather_points<<<cdiv64(cap, 256), 256, 0, s>>>(v.in, ws.vals[1], ws.geom, out);
    k_part_counts<<<cdiv64(n_parts, 64), 64, 0, s>>>(ws.keys[1], ws.geom, n_parts, counts_dev, overflow_dev);
}

// Statistical outlier removal of ONE cloud (`in`, count in n_dev[0], at most cap points, bounding boxes in
// ws.mm slots [0, mm_used)).  Kept points -> out (same order), their count -> n_out_dev[0], their
// bounding boxes -> ws.mm slots [0, returned value).
int launch_sor(Profiler* pf, hipStream_t s, Workspace& ws, const o3dr_point* in, const uint32_t* n_dev, int64_t cap,
               int mm_used, double stddev_mul, o3dr_point* out, uint32_t* n_out_dev)
{
    ProfScope ps(pf, O3DR_K_OTHER, s);
    const int n_sort_tiles = cdiv64(cap, kSortTile);
    const int64_t hist_row = (int64_t)kMaxRadix * n_sort_tiles;
    const int n_tiles = cdiv64(cap, 1024);
    k_sor_plan<<<1, 256, 0, s>>>(ws.mm, mm_used, n_dev, ws.sor_max_cells, ws.sor_geom, ws.geom);
    (void)hipMemsetAsync(ws.sor_cell_start, 0, (size_t)ws.sor_max_cells * 4, s);
    (void)hipMemsetAsync(ws.sor_cell_end, 0, (size_t)ws.sor_max_cells * 4, s);
    k_sor_cells<<<cdiv64(cap, 256), 256, 0, s>>>(in, ws.sor_geom, ws.keys[0]);
    for (int pass = 0; pass < kMaxPasses; ++pass) {
        k_radix_hist<<<dim3(n_sort_tiles, 1), kSortThreads, 0, s>>>(ws.keys[0], ws.keys[1], cap, ws.geom, pass, n_sort_tiles, ws.hist);
        launch_scan(s, ws.hist, hist_row, hist_row, 1, nullptr, nullptr, ws.scan_partial, ws.geom, pass, n_sort_tiles);
        k_radix_scatter<false><<<dim3(n_sort_tiles, 1), kSortThreads, 0, s>>>(ws.keys[0], ws.vals[0], ws.keys[1], ws.vals[1], cap,
                                                                             ws.geom, pass, n_sort_tiles, ws.hist, nullptr, nullptr,
                                                                             nullptr, 0u, nullptr, nullptr);
    }
    k_sor_cell_table<<<cdiv64(cap, 256), 256, 0, s>>>(in, ws.keys[0], ws.keys[1], ws.vals[0], ws.vals[1], ws.sor_geom, ws.geom,
                                                     ws.sor_xyz, ws.sor_cell_start, ws.sor_cell_end);
    k_sor_knn<<<cdiv64(cap, kSorThreads), kSorThreads, 0, s>>>(ws.sor_xyz, ws.vals[0], ws.vals[1], ws.geom, ws.sor_cell_start,
                                                              ws.sor_cell_end, ws.sor_geom, ws.sor_dist);
    constexpr int kStatBlocks = 256;
    k_sor_partial<<<kStatBlocks, 256, 0, s>>>(ws.sor_dist, ws.sor_geom, ws.sor_partial);
    k_sor_threshold<<<1, 1, 0, s>>>(ws.sor_partial, kStatBlocks, stddev_mul, ws.sor_geom);
    k_sor_count<<<n_tiles, 256, 0, s>>>(ws.sor_dist, ws.sor_geom, ws.tile_cnt);
    launch_scan(s, ws.tile_cnt, n_tiles, n_tiles, 1, n_out_dev, nullptr, ws.scan_partial);
    k_sor_emit<<<n_tiles, 256, 0, s>>>(in, ws.sor_dist, ws.sor_geom, ws.tile_cnt, out, ws.mm_stride, ws.mm);
    return n_tiles;
}

}  // namespace o3dr
