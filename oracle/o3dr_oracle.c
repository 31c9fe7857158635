/*
 * o3dr_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see o3dr_oracle.h).
 *
 * Plain-C restatement of the reference's per-frame reconstruction arithmetic.  Each function cites
 * the reference lines it follows (paths relative to the reference tree) and, where the arithmetic
 * lives in a third-party library that is not in the reference tree, the library + version whose
 * published algorithm is restated:
 *     OpenCV 3.1.0  (build/CMakeFiles/pose.dir/link.txt)   cv::Mat_<double> product, Mat /= scalar
 *     PCL 1.8.x     (build/CMakeCache.txt:412)             transformPointCloud, VoxelGrid,
 *                                                          CentroidPoint, getMinMax3D
 * Parity unpinned by the reference (it has no tests); pinned here by analytic known answers
 * (tests/test_oracle_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  The reference was compiled
 * without -O (build/CMakeFiles/pose.dir/flags.make), so no multiply-add was ever fused; x86-64 SSE2
 * evaluates float expressions in float and double expressions in double, as this file assumes.
 */
#include "o3dr_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * A1: per-pixel reprojection.  pose_functions.cpp:1110-1121 (grid pass) == :1074-1083 (keypoints):
 *     vec = (x, y, disp, 1);  vec = Q*vec;  vec /= vec(3);  pt = (float)vec(0..2);
 *     rgb = R<<16 | G<<8 | B from the BGR pixel.
 * cv::Mat_<double>(4x4)*(4x1) goes through cv::gemm's small-matrix path, which evaluates each row
 * as a0*b0 + a1*b1 + a2*b2 + a3*b3 left to right in double [OpenCV 3.1 core/src/matmul.cpp].
 * `Mat /= s` is `convertTo(m, -1, 1./s)`, i.e. every element becomes v*(1./s) + 0
 * [OpenCV 3.1 core/src/operations / convert.cpp cvtScale_].
 * ---------------------------------------------------------------------------------------------- */
static inline void reproject_pixel(const double Q[16], int x, int y, double d, const uint8_t* bgr_px,
                                   orc_point* pt)
{
    const double v0 = (double)x, v1 = (double)y, v2 = d, v3 = 1.0;
    double t[4];
    for (int r = 0; r < 4; ++r) {
        const double* q = Q + 4 * r;
        t[r] = ((q[0] * v0 + q[1] * v1) + q[2] * v2) + q[3] * v3;
    }
    const double alpha = 1. / t[3];
    const double X = t[0] * alpha + 0.0;
    const double Y = t[1] * alpha + 0.0;
    const double Z = t[2] * alpha + 0.0;
    pt->x = (float)X;
    pt->y = (float)Y;
    pt->z = (float)Z;
    /* Vec3b color = rgb_image.at<Vec3b>(Point(x,y)); color[2]<<16 | color[1]<<8 | color[0] */
    pt->rgba = ((uint32_t)bgr_px[2] << 16) | ((uint32_t)bgr_px[1] << 8) | (uint32_t)bgr_px[0];
}

/* dispImg.at<uchar>(y,x) or, with use_segment_labels, dispImg.at<double>(y,x) — pose_functions.cpp:1064-1069,1100-1105 */
static inline double disp_at(const uint8_t* disp, int64_t pitch, int f64, int y, int x)
{
    const uint8_t* row = disp + (int64_t)y * pitch;
    return f64 ? ((const double*)row)[x] : (double)row[x];
}

static int64_t create_single_impl(const uint8_t* disp, int64_t disp_pitch, int f64,
                                  const uint8_t* bgr, int64_t bgr_pitch,
                                  int32_t rows, int32_t cols, const double Q[16],
                                  int32_t bounding_box, int32_t cols_start_aft_cutout,
                                  double min_disparity, int32_t jump_pixels,
                                  const float* kp_xy, int32_t n_kp, orc_point* out)
{
    int64_t n = 0;
    /* keypoint pass — pose_functions.cpp:1057-1091 */
    if (jump_pixels != 1) {
        for (int32_t i = 0; i < n_kp; ++i) {
            const int x = (int)kp_xy[2 * i], y = (int)kp_xy[2 * i + 1]; /* :1061 float -> int */
            if (x >= cols_start_aft_cutout && x < cols - bounding_box && y >= bounding_box &&
                y < rows - bounding_box) { /* :1062 */
                const double d = disp_at(disp, disp_pitch, f64, y, x);
                if (d > min_disparity) /* :1070 */
                    reproject_pixel(Q, x, y, d, bgr + (int64_t)y * bgr_pitch + 3 * (int64_t)x, &out[n++]);
            }
        }
    }
    /* grid pass — pose_functions.cpp:1092-1130 */
    if (jump_pixels > 0) {
        for (int y = bounding_box; y < rows - bounding_box; y += jump_pixels) {
            for (int x = cols_start_aft_cutout; x < cols - bounding_box; x += jump_pixels) {
                const double d = disp_at(disp, disp_pitch, f64, y, x); /* :1104 */
                if (d > min_disparity)                                       /* :1107 */
                    reproject_pixel(Q, x, y, d, bgr + (int64_t)y * bgr_pitch + 3 * (int64_t)x, &out[n++]);
            }
        }
    }
    return n;
}

int64_t orc_create_single_img_pt_cloud(const uint8_t* disp, int64_t disp_pitch,
                                       const uint8_t* bgr, int64_t bgr_pitch,
                                       int32_t rows, int32_t cols, const double Q[16],
                                       int32_t bounding_box, int32_t cols_start_aft_cutout,
                                       double min_disparity, int32_t jump_pixels,
                                       const float* kp_xy, int32_t n_kp, orc_point* out)
{
    return create_single_impl(disp, disp_pitch, 0, bgr, bgr_pitch, rows, cols, Q, bounding_box, cols_start_aft_cutout,
                              min_disparity, jump_pixels, kp_xy, n_kp, out);
}
int64_t orc_create_single_img_pt_cloud_f64(const double* disp, int64_t disp_pitch_bytes,
                                           const uint8_t* bgr, int64_t bgr_pitch,
                                           int32_t rows, int32_t cols, const double Q[16],
                                           int32_t bounding_box, int32_t cols_start_aft_cutout,
                                           double min_disparity, int32_t jump_pixels,
                                           const float* kp_xy, int32_t n_kp, orc_point* out)
{
    return create_single_impl((const uint8_t*)disp, disp_pitch_bytes, 1, bgr, bgr_pitch, rows, cols, Q, bounding_box,
                              cols_start_aft_cutout, min_disparity, jump_pixels, kp_xy, n_kp, out);
}

/* ------------------------------------------------------------------------------------------------
 * A2: pcl::transformPointCloud(cloud_in, cloud_out, Matrix4f), dense branch
 * [PCL 1.8 common/impl/transforms.hpp]:
 *     out.x = static_cast<float>(t(0,0)*pt(0) + t(0,1)*pt(1) + t(0,2)*pt(2) + t(0,3));  (y, z alike)
 * all operands float, evaluated left to right; the other fields are copied.
 * ---------------------------------------------------------------------------------------------- */
void orc_transform_pt_cloud(const orc_point* in, int64_t n, const float T[16], orc_point* out)
{
    for (int64_t i = 0; i < n; ++i) {
        const float x = in[i].x, y = in[i].y, z = in[i].z;
        orc_point o;
        o.x = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
        o.y = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
        o.z = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
        o.rgba = in[i].rgba;
        out[i] = o;
    }
}

/* ------------------------------------------------------------------------------------------------
 * A4: pcl::VoxelGrid<PointXYZRGB>::applyFilter [PCL 1.8 filters/impl/voxel_grid.hpp], with the
 * defaults the reference leaves in place: no filter field, downsample_all_data_ = true,
 * save_leaf_layout_ = false, dense input (createSingleImgPtCloud sets is_dense, :1033).
 * ---------------------------------------------------------------------------------------------- */
typedef struct voxel_grid_geom {
    float inv[3];
    int32_t min_b[3], max_b[3], div_b[3], divb_mul[3];
    int overflow;
} voxel_grid_geom;

static void voxel_grid_setup(const orc_point* in, int64_t n, const float leaf[3], voxel_grid_geom* g)
{
    /* setLeafSize: inverse_leaf_size_ = Array4f::Ones() / leaf_size_.array()  (fp32 division) */
    for (int a = 0; a < 3; ++a) g->inv[a] = 1.0f / leaf[a];

    /* getMinMax3D(cloud, indices, min_p, max_p): component-wise fp32 min/max from +-FLT_MAX */
    float min_p[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, max_p[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int64_t i = 0; i < n; ++i) {
        const float p[3] = {in[i].x, in[i].y, in[i].z};
        for (int a = 0; a < 3; ++a) {
            if (p[a] < min_p[a]) min_p[a] = p[a];
            if (p[a] > max_p[a]) max_p[a] = p[a];
        }
    }
    /* int64_t dx = static_cast<int64_t>((max_p[0]-min_p[0]) * inverse_leaf_size_[0]) + 1; ...
       if (dx*dy*dz > INT32_MAX) { warn; output = *input_; return; } */
    int64_t d[3];
    for (int a = 0; a < 3; ++a) d[a] = (int64_t)((max_p[a] - min_p[a]) * g->inv[a]) + 1;
    g->overflow = (d[0] * d[1] * d[2]) > (int64_t)INT32_MAX;

    /* min_b_[a] = static_cast<int>(floor(min_p[a] * inverse_leaf_size_[a])); max_b_ alike */
    for (int a = 0; a < 3; ++a) {
        g->min_b[a] = (int32_t)floor((double)(min_p[a] * g->inv[a]));
        g->max_b[a] = (int32_t)floor((double)(max_p[a] * g->inv[a]));
        g->div_b[a] = g->max_b[a] - g->min_b[a] + 1;
    }
    g->divb_mul[0] = 1;
    g->divb_mul[1] = g->div_b[0];
    g->divb_mul[2] = (int32_t)((uint32_t)g->div_b[0] * (uint32_t)g->div_b[1]);
}

static inline uint32_t voxel_key(const voxel_grid_geom* g, float x, float y, float z)
{
    /* ijk0 = static_cast<int>(floor(p.x * inverse_leaf_size_[0]) - static_cast<float>(min_b_[0]));
       The product is fp32; floor() of it and the subtraction are exact for |cell| < 2^24 whether
       they are carried out in float or in double (GCC 5's unqualified floor -> double). */
    const int32_t i0 = (int32_t)(floor((double)(x * g->inv[0])) - (double)(float)g->min_b[0]);
    const int32_t i1 = (int32_t)(floor((double)(y * g->inv[1])) - (double)(float)g->min_b[1]);
    const int32_t i2 = (int32_t)(floor((double)(z * g->inv[2])) - (double)(float)g->min_b[2]);
    /* int idx = ijk0*divb_mul_[0] + ijk1*divb_mul_[1] + ijk2*divb_mul_[2]; -> unsigned int */
    return (uint32_t)i0 * (uint32_t)g->divb_mul[0] + (uint32_t)i1 * (uint32_t)g->divb_mul[1] +
           (uint32_t)i2 * (uint32_t)g->divb_mul[2];
}

uint32_t orc_voxel_keys(const orc_point* in, int64_t n, const float leaf[3], uint32_t* keys,
                        int32_t min_b[3], int32_t div_b[3])
{
    voxel_grid_geom g;
    if (n <= 0) return 0;
    voxel_grid_setup(in, n, leaf, &g);
    for (int a = 0; a < 3; ++a) {
        if (min_b) min_b[a] = g.min_b[a];
        if (div_b) div_b[a] = g.div_b[a];
    }
    if (g.overflow) return ORC_STATUS_VOXEL_OVERFLOW;
    for (int64_t i = 0; i < n; ++i) keys[i] = voxel_key(&g, in[i].x, in[i].y, in[i].z);
    return 0;
}

/* stable LSD radix sort of (key, value) pairs: equal keys keep ascending input index */
static void stable_sort_pairs(uint32_t* key, uint32_t* val, int64_t n)
{
    uint32_t* k2 = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    uint32_t* v2 = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    uint32_t *ka = key, *va = val, *kb = k2, *vb = v2;
    for (int pass = 0; pass < 4; ++pass) {
        const int sh = 8 * pass;
        int64_t cnt[257];
        memset(cnt, 0, sizeof cnt);
        for (int64_t i = 0; i < n; ++i) cnt[((ka[i] >> sh) & 255u) + 1]++;
        for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; ++i) {
            const int64_t p = cnt[(ka[i] >> sh) & 255u]++;
            kb[p] = ka[i];
            vb[p] = va[i];
        }
        uint32_t* t;
        t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    /* 4 passes: result is back in key/val */
    free(k2);
    free(v2);
}

int64_t orc_voxel_grid(const orc_point* in, int64_t n, const float leaf[3], uint32_t min_points,
                       int32_t order, orc_point* out, uint32_t* status)
{
    if (status) *status = 0;
    if (n <= 0) return 0;
    voxel_grid_geom g;
    voxel_grid_setup(in, n, leaf, &g);
    if (g.overflow) { /* "Leaf size is too small ... Integer indices would overflow": output = input */
        if (status) *status = ORC_STATUS_VOXEL_OVERFLOW;
        if (out != in) memmove(out, in, (size_t)n * sizeof(orc_point));
        return n;
    }
    /* first pass: (idx, cloud_point_index) for every point */
    uint32_t* key = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    uint32_t* pid = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    for (int64_t i = 0; i < n; ++i) {
        key[i] = voxel_key(&g, in[i].x, in[i].y, in[i].z);
        pid[i] = (uint32_t)i;
    }
    /* second pass: std::sort(index_vector.begin(), index_vector.end(), std::less<...>()) on idx.
       std::sort is not stable, so the order of the points inside one voxel is whatever libstdc++'s
       introsort leaves; ORC_ORDER_STDSORT reproduces that, ORC_ORDER_STABLE is the canonical
       ascending-input-index order. */
    if (order == ORC_ORDER_STDSORT)
        orc_stdsort_pairs(key, pid, n);
    else
        stable_sort_pairs(key, pid, n);

    /* third + fourth pass: runs of equal idx with >= min_points_per_voxel_ points become one output
       point each, via CentroidPoint<PointXYZRGB> [PCL 1.8 common/impl/accumulators.hpp]:
         AccumulatorXYZ : Vector3f xyz += p.xyz;            get: xyz / n   (fp32 sums, fp32 divide)
         AccumulatorRGBA: float r,g,b,a += channel;          get: uint32_t(c / n) per channel */
    int64_t m = 0;
    orc_point* dst = out;
    orc_point* tmp = NULL;
    if (out == in) { /* allow in-place use */
        tmp = (orc_point*)malloc((size_t)n * sizeof(orc_point));
        dst = tmp;
    }
    int64_t index = 0;
    while (index < n) {
        int64_t i = index + 1;
        while (i < n && key[i] == key[index]) ++i;
        if ((uint64_t)(i - index) >= (uint64_t)min_points) {
            float sx = 0.f, sy = 0.f, sz = 0.f, sr = 0.f, sg = 0.f, sb = 0.f, sa = 0.f;
            for (int64_t li = index; li < i; ++li) {
                const orc_point* p = &in[pid[li]];
                sx += p->x;
                sy += p->y;
                sz += p->z;
                sr += (float)((p->rgba >> 16) & 255u);
                sg += (float)((p->rgba >> 8) & 255u);
                sb += (float)(p->rgba & 255u);
                sa += (float)((p->rgba >> 24) & 255u);
            }
            /* PCL 1.8 AccumulatorXYZ::get: `xyz / n` on an Eigen::Vector3f - a TRUE division per coefficient by float(n)
             * from Eigen 3.2 on (scalar_quotient1_op: `a / m_other`; Eigen 3.0 / 3.1 multiplied by the reciprocal for
             * non-integer scalars).  The reference's build used the system Eigen at /usr/include/eigen3
             * (build/CMakeCache.txt:350, 1188); its version string was not recorded (:933 `[v()]`), but the same cache
             * pins PCL 1.8 (:412, released 2016), Boost 1.61 and CUDA 8.0 - an Ubuntu 14.04 / 16.04 system, whose
             * packaged Eigen is 3.2.0 / 3.2.92: division either way.  AccumulatorRGBA::get divides its float sums by n
             * the same way and truncates to uint32. */
            const float nf = (float)(uint64_t)(i - index);
            orc_point o;
            o.x = sx / nf;
            o.y = sy / nf;
            o.z = sz / nf;
            o.rgba = ((uint32_t)(sa / nf) << 24) | ((uint32_t)(sr / nf) << 16) |
                     ((uint32_t)(sg / nf) << 8) | (uint32_t)(sb / nf);
            dst[m++] = o;
        }
        index = i;
    }
    if (tmp) {
        memcpy(out, tmp, (size_t)m * sizeof(orc_point));
        free(tmp);
    }
    free(key);
    free(pid);
    return m;
}

/* ------------------------------------------------------------------------------------------------
 * A3a / A5: Pose::downsamplePtCloud, pose_functions.cpp:1654-1709.
 *   copy loop with `z += 500` when combined                          :1660-1669
 *   [StatisticalOutlierRemoval when !combined && jump_pixels > 0     :1673-1686 — NOT applied here]
 *   VoxelGrid leaf (vs,vs,1000) + min_points_per_voxel, or (vs/5)^3   :1689-1700
 *   `z -= 500` when combined                                          :1702-1704
 * voxel_size is a double (pose.h:118); setLeafSize takes floats, so vs and vs/5 are narrowed.
 * ---------------------------------------------------------------------------------------------- */
int64_t orc_downsample_pt_cloud(const orc_point* in, int64_t n, double voxel_size, int32_t combined,
                                uint32_t min_points_per_voxel, int32_t order, orc_point* out,
                                uint32_t* status)
{
    if (status) *status = 0;
    if (n <= 0) return 0;
    orc_point* work = (orc_point*)malloc((size_t)n * sizeof(orc_point));
    memcpy(work, in, (size_t)n * sizeof(orc_point));
    float leaf[3];
    uint32_t min_pts;
    if (combined) {
        for (int64_t i = 0; i < n; ++i) work[i].z += 500; /* fp32: float += int -> float */
        leaf[0] = (float)voxel_size;
        leaf[1] = (float)voxel_size;
        leaf[2] = 1000.0f;
        min_pts = min_points_per_voxel;
    } else {
        leaf[0] = leaf[1] = leaf[2] = (float)(voxel_size / 5);
        min_pts = 0; /* PCL default min_points_per_voxel_ */
    }
    const int64_t m = orc_voxel_grid(work, n, leaf, min_pts, order, out, status);
    if (combined)
        for (int64_t i = 0; i < m; ++i) out[i].z -= 500;
    free(work);
    return m;
}

/* ------------------------------------------------------------------------------------------------
 * A6: Pose::createAndTransformPtCloud, pose.cpp:596-636: A1 -> A2 -> (A3a unless dont_downsample).
 * ---------------------------------------------------------------------------------------------- */
int64_t orc_create_and_transform_pt_cloud(const uint8_t* disp, int64_t disp_pitch,
                                          const uint8_t* bgr, int64_t bgr_pitch,
                                          int32_t rows, int32_t cols, const double Q[16],
                                          int32_t bounding_box, int32_t cols_start_aft_cutout,
                                          double min_disparity, int32_t jump_pixels,
                                          const float* kp_xy, int32_t n_kp, const float T[16],
                                          double voxel_size, int32_t dont_downsample, int32_t order,
                                          orc_point* scratch, orc_point* out, uint32_t* status)
{
    if (status) *status = 0;
    const int64_t n = orc_create_single_img_pt_cloud(disp, disp_pitch, bgr, bgr_pitch, rows, cols, Q,
                                                     bounding_box, cols_start_aft_cutout,
                                                     min_disparity, jump_pixels, kp_xy, n_kp, scratch);
    orc_point* tr = scratch + n;
    orc_transform_pt_cloud(scratch, n, T, tr);
    if (dont_downsample) {
        memcpy(out, tr, (size_t)n * sizeof(orc_point));
        return n;
    }
    return orc_downsample_pt_cloud(tr, n, voxel_size, 0, 0, order, out, status);
}

/* ------------------------------------------------------------------------------------------------
 * A3b: pcl::StatisticalOutlierRemoval<PointXYZRGB>::applyFilterIndices
 * [PCL 1.8 filters/impl/statistical_outlier_removal.hpp], called at pose_functions.cpp:1679-1684 with
 * mean_k = 50, stddev_mul = 1.0, on an unorganised cloud (height 1) => pcl::search::KdTree ->
 * KdTreeFLANN, exact search, flann::L2_Simple<float>:
 *     d2 = ((0 + dx*dx) + dy*dy) + dz*dz           (float, dx = query - data)
 *   nearestKSearch(i, mean_k + 1): the mean_k+1 smallest d2 in ascending order (entry 0 is the query)
 *   distances[i] = float( sum_{k=1..mean_k} sqrt(double(d2[k])) / mean_k )            (double sum)
 *   sum += distances[i];  sq_sum += distances[i]*distances[i]  (float product, double sums, index order)
 *   mean = sum/n; variance = (sq_sum - sum*sum/n)/(n-1); threshold = mean + stddev_mul*sqrt(variance)
 *   keep i  iff  !(distances[i] > threshold)
 * Which of several equidistant neighbours the kd-tree returns does not matter: only the distance
 * values enter.  With n <= mean_k the reference reads past the neighbour list (undefined); here such
 * clouds pass through unfiltered.
 * The neighbour search below is a uniform XY grid with ring expansion and a conservative stop
 * (no unvisited column can hold a point closer than 0.999*r*h); it is exact for any cell size.
 * ---------------------------------------------------------------------------------------------- */
static inline float sor_d2(const orc_point* q, const orc_point* p)
{
    const float dx = q->x - p->x, dy = q->y - p->y, dz = q->z - p->z;
    return ((0.0f + dx * dx) + dy * dy) + dz * dz;
}
/* keep the kk smallest values of a stream in ascending order */
static inline void sor_insert(float* best, int kk, float d)
{
    if (!(d < best[kk - 1])) return;
    int j = kk - 1;
    while (j > 0 && best[j - 1] > d) {
        best[j] = best[j - 1];
        --j;
    }
    best[j] = d;
}

int64_t orc_statistical_outlier_removal(const orc_point* in, int64_t n, int32_t mean_k, double stddev_mul,
                                        orc_point* out, float* distances_out, int32_t brute)
{
    if (n <= 0) return 0;
    if (n <= mean_k || mean_k < 1) {
        if (out != in) memmove(out, in, (size_t)n * sizeof(orc_point));
        if (distances_out) memset(distances_out, 0, (size_t)n * sizeof(float));
        return n;
    }
    const int kk = mean_k + 1;
    float* dist = (float*)malloc((size_t)n * sizeof(float));
    float* best = (float*)malloc((size_t)kk * sizeof(float));

    /* grid over XY */
    float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
    for (int64_t i = 0; i < n; ++i) {
        if (in[i].x < mnx) mnx = in[i].x;
        if (in[i].x > mxx) mxx = in[i].x;
        if (in[i].y < mny) mny = in[i].y;
        if (in[i].y > mxy) mxy = in[i].y;
    }
    double ex = (double)mxx - mnx, ey = (double)mxy - mny;
    double h = sqrt(8.0 * (ex > 1e-9 ? ex : 1e-9) * (ey > 1e-9 ? ey : 1e-9) / (double)n);
    if (h < 1e-6) h = 1e-6;
    int64_t gx = (int64_t)(ex / h) + 1, gy = (int64_t)(ey / h) + 1;
    while (gx * gy > ((int64_t)1 << 24)) {
        h *= 1.5;
        gx = (int64_t)(ex / h) + 1;
        gy = (int64_t)(ey / h) + 1;
    }
    int32_t* cell_of = NULL;
    int64_t* start = NULL;
    int32_t* order = NULL;
    if (!brute) {
        cell_of = (int32_t*)malloc((size_t)n * sizeof(int32_t));
        start = (int64_t*)calloc((size_t)(gx * gy + 1), sizeof(int64_t));
        order = (int32_t*)malloc((size_t)n * sizeof(int32_t));
        for (int64_t i = 0; i < n; ++i) {
            int64_t cx = (int64_t)(((double)in[i].x - mnx) / h), cy = (int64_t)(((double)in[i].y - mny) / h);
            if (cx >= gx) cx = gx - 1;
            if (cy >= gy) cy = gy - 1;
            cell_of[i] = (int32_t)(cy * gx + cx);
            start[cell_of[i] + 1]++;
        }
        for (int64_t c = 0; c < gx * gy; ++c) start[c + 1] += start[c];
        int64_t* fill = (int64_t*)malloc((size_t)(gx * gy) * sizeof(int64_t));
        memcpy(fill, start, (size_t)(gx * gy) * sizeof(int64_t));
        for (int64_t i = 0; i < n; ++i) order[fill[cell_of[i]]++] = (int32_t)i;
        free(fill);
    }

    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < kk; ++k) best[k] = FLT_MAX;
        if (brute) {
            for (int64_t j = 0; j < n; ++j) sor_insert(best, kk, sor_d2(&in[i], &in[j]));
        } else {
            const int64_t cx = cell_of[i] % gx, cy = cell_of[i] / gx;
            const int64_t rmax = (gx > gy ? gx : gy);
            for (int64_t r = 0; r <= rmax; ++r) {
                for (int64_t yy = cy - r; yy <= cy + r; ++yy) {
                    if (yy < 0 || yy >= gy) continue;
                    const int64_t step = (yy == cy - r || yy == cy + r) ? 1 : 2 * r; /* ring: full rows top/bottom, ends otherwise */
                    for (int64_t xx = cx - r; xx <= cx + r; xx += (step > 0 ? step : 1)) {
                        if (xx < 0 || xx >= gx) continue;
                        const int64_t c = yy * gx + xx;
                        for (int64_t s = start[c]; s < start[c + 1]; ++s) sor_insert(best, kk, sor_d2(&in[i], &in[order[s]]));
                    }
                }
                /* every unvisited column is more than r*h away in x or y (cell rounding covered by 0.999) */
                const double bound = 0.999 * (double)r * h;
                if (best[kk - 1] < FLT_MAX && (double)best[kk - 1] <= bound * bound) break;
            }
        }
        double dist_sum = 0.0;
        for (int k = 1; k < kk; ++k) dist_sum += sqrt((double)best[k]);
        dist[i] = (float)(dist_sum / mean_k);
    }
    double sum = 0, sq_sum = 0;
    for (int64_t i = 0; i < n; ++i) {
        sum += dist[i];
        sq_sum += dist[i] * dist[i]; /* float product, as in PCL */
    }
    const double mean = sum / (double)n;
    const double variance = (sq_sum - sum * sum / (double)n) / ((double)n - 1);
    const double stddev = sqrt(variance);
    const double threshold = mean + stddev_mul * stddev;
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i)
        if (!(dist[i] > threshold)) out[m++] = in[i];
    if (distances_out) memcpy(distances_out, dist, (size_t)n * sizeof(float));
    free(dist);
    free(best);
    free(cell_of);
    free(start);
    free(order);
    return m;
}

/* ------------------------------------------------------------------------------------------------
 * A7: the fan-out of pose.cpp:365-434 (batches of frame-parallel threads, results appended in frame
 * order) followed by the combined merge of pose.cpp:530.  Used as the timed CPU baseline.
 * ---------------------------------------------------------------------------------------------- */
typedef struct frame_job {
    const uint8_t *disp, *bgr;
    int64_t disp_fstride, disp_pitch, bgr_fstride, bgr_pitch;
    int32_t rows, cols, bb, cs, jump, n_frames, sor, order;
    const double* Q;
    double min_disp, voxel_size;
    const float* poses;
    orc_point** out;
    int64_t* out_n;
    int64_t cap;
    volatile int32_t* next;
} frame_job;

static void* frame_worker(void* arg)
{
    frame_job* j = (frame_job*)arg;
    orc_point* scratch = (orc_point*)malloc((size_t)(2 * j->cap + 1) * sizeof(orc_point));
    orc_point* tmp = (orc_point*)malloc((size_t)(j->cap + 1) * sizeof(orc_point));
    for (;;) {
        const int32_t f = __sync_fetch_and_add(j->next, 1);
        if (f >= j->n_frames) break;
        const uint8_t* d = j->disp + (int64_t)f * j->disp_fstride;
        const uint8_t* c = j->bgr + (int64_t)f * j->bgr_fstride;
        const float* T = j->poses + 16 * (int64_t)f;
        int64_t n;
        if (!j->sor) {
            n = orc_create_and_transform_pt_cloud(d, j->disp_pitch, c, j->bgr_pitch, j->rows, j->cols, j->Q, j->bb, j->cs,
                                                  j->min_disp, j->jump, NULL, 0, T, j->voxel_size, 0, j->order,
                                                  scratch, tmp, NULL);
        } else {
            const int64_t n0 = orc_create_single_img_pt_cloud(d, j->disp_pitch, c, j->bgr_pitch, j->rows, j->cols, j->Q, j->bb,
                                                              j->cs, j->min_disp, j->jump, NULL, 0, scratch);
            orc_transform_pt_cloud(scratch, n0, T, scratch + n0);
            const int64_t n1 = j->jump > 0 ? orc_statistical_outlier_removal(scratch + n0, n0, 50, 1.0, scratch, NULL, 0) : n0;
            n = orc_downsample_pt_cloud(j->jump > 0 ? scratch : scratch + n0, n1, j->voxel_size, 0, 0, j->order, tmp, NULL);
        }
        j->out[f] = (orc_point*)malloc((size_t)(n + 1) * sizeof(orc_point));
        memcpy(j->out[f], tmp, (size_t)n * sizeof(orc_point));
        j->out_n[f] = n;
    }
    free(scratch);
    free(tmp);
    return NULL;
}

int64_t orc_run_frames_order(const uint8_t* disp, int64_t disp_fstride, int64_t disp_pitch, const uint8_t* bgr,
                             int64_t bgr_fstride, int64_t bgr_pitch, int32_t rows, int32_t cols, const double Q[16],
                             int32_t bounding_box, int32_t cols_start_aft_cutout, double min_disparity,
                             int32_t jump_pixels, const float* poses, int32_t n_frames, double voxel_size,
                             uint32_t min_points_per_voxel, int32_t sor, int32_t threads, int32_t order,
                             orc_point* cloud_big_out, int64_t* n_big, orc_point* merged_out)
{
    if (n_frames <= 0) return 0;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    volatile int32_t next = 0;
    frame_job j;
    memset(&j, 0, sizeof j);
    j.disp = disp; j.bgr = bgr;
    j.disp_fstride = disp_fstride; j.disp_pitch = disp_pitch; j.bgr_fstride = bgr_fstride; j.bgr_pitch = bgr_pitch;
    j.rows = rows; j.cols = cols; j.bb = bounding_box; j.cs = cols_start_aft_cutout; j.jump = jump_pixels;
    j.n_frames = n_frames; j.sor = sor; j.order = order; j.Q = Q; j.min_disp = min_disparity; j.voxel_size = voxel_size; j.poses = poses;
    j.out = (orc_point**)calloc((size_t)n_frames, sizeof(orc_point*));
    j.out_n = (int64_t*)calloc((size_t)n_frames, sizeof(int64_t));
    j.next = &next;
    if (jump_pixels > 0) {
        const int64_t h = rows - 2 * bounding_box, w = cols - bounding_box - cols_start_aft_cutout;
        j.cap = (h > 0 ? (h + jump_pixels - 1) / jump_pixels : 0) * (w > 0 ? (w + jump_pixels - 1) / jump_pixels : 0);
    }
    pthread_t th[64];
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, frame_worker, &j);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    int64_t total = 0;
    for (int f = 0; f < n_frames; ++f) total += j.out_n[f];
    orc_point* big = cloud_big_out ? cloud_big_out : (orc_point*)malloc((size_t)(total + 1) * sizeof(orc_point));
    int64_t off = 0;
    for (int f = 0; f < n_frames; ++f) {  /* frame order, pose.cpp:418-424 */
        memcpy(big + off, j.out[f], (size_t)j.out_n[f] * sizeof(orc_point));
        off += j.out_n[f];
        free(j.out[f]);
    }
    if (n_big) *n_big = total;
    orc_point* merged = merged_out ? merged_out : (orc_point*)malloc((size_t)(total + 1) * sizeof(orc_point));
    const int64_t m = orc_downsample_pt_cloud(big, total, voxel_size, 1, min_points_per_voxel, order, merged, NULL);
    if (!merged_out) free(merged);
    if (!cloud_big_out) free(big);
    free(j.out);
    free(j.out_n);
    return m;
}

int64_t orc_run_frames(const uint8_t* disp, int64_t disp_fstride, int64_t disp_pitch, const uint8_t* bgr,
                       int64_t bgr_fstride, int64_t bgr_pitch, int32_t rows, int32_t cols, const double Q[16],
                       int32_t bounding_box, int32_t cols_start_aft_cutout, double min_disparity,
                       int32_t jump_pixels, const float* poses, int32_t n_frames, double voxel_size,
                       uint32_t min_points_per_voxel, int32_t sor, int32_t threads, orc_point* cloud_big_out,
                       int64_t* n_big, orc_point* merged_out)
{
    return orc_run_frames_order(disp, disp_fstride, disp_pitch, bgr, bgr_fstride, bgr_pitch, rows, cols, Q, bounding_box,
                                cols_start_aft_cutout, min_disparity, jump_pixels, poses, n_frames, voxel_size,
                                min_points_per_voxel, sor, threads, ORC_ORDER_STABLE, cloud_big_out, n_big, merged_out);
}


/* ------------------------------------------------------------------------------------------------
 * cv::bilateralFilter on CV_8UC1 — OpenCV 3.1.0 smooth.cpp, bilateralFilter_8u + BilateralFilter_8u_Invoker
 * (called from pose_functions.cpp:1044)
 * ------------------------------------------------------------------------------------------------ */
static int reflect101(int p, int len) /* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0)
            p = -p;
        else
            p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* What bilateralFilter_8u sets up before its row loop, for cn interleaved channels: the reflect-101 padded copy, the
 * colour weights (index = sum over the channels of |v - v0|, so 256 * cn entries), the space weights and offsets of the
 * neighbours inside the disc.  Shared by the one-channel call of the hot path (pose_functions.cpp:1044) and the
 * three-channel form, whose output the reference holds (build/output/bilateralFiltered_15.png, _31.png: 1248.png through
 * exactly this call with d = 15 / 31, sigmaColor = 2 d, sigmaSpace = d / 2 in integer arithmetic;
 * tests/test_oracle_golden.py::test_bilateral_filter_equals_the_reference_runs_own_output). */
typedef struct {
    int radius, maxk, tw, th, cn;
    uint8_t* temp;
    float* color_weight;
    float* space_weight;
    int* space_ofs;
} bilateral_setup;

static void bilateral_prepare(bilateral_setup* b, const uint8_t* src, int64_t src_pitch, int rows, int cols, int cn, int d,
                              double sigma_color, double sigma_space)
{
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    const double gauss_color_coeff = -0.5 / (sigma_color * sigma_color);
    const double gauss_space_coeff = -0.5 / (sigma_space * sigma_space);
    int radius = d <= 0 ? (int)lrint(sigma_space * 1.5) /* cvRound */ : d / 2;
    if (radius < 1) radius = 1;
    d = radius * 2 + 1;
    b->radius = radius;
    b->cn = cn;
    /* copyMakeBorder(src, temp, radius, radius, radius, radius, BORDER_DEFAULT) */
    const int tw = cols + 2 * radius, th = rows + 2 * radius;
    b->tw = tw;
    b->th = th;
    b->temp = (uint8_t*)malloc((size_t)tw * (size_t)th * cn);
    for (int y = 0; y < th; ++y) {
        const uint8_t* srow = src + (int64_t)reflect101(y - radius, rows) * src_pitch;
        for (int x = 0; x < tw; ++x)
            for (int c = 0; c < cn; ++c) b->temp[((size_t)y * tw + x) * cn + c] = srow[reflect101(x - radius, cols) * cn + c];
    }
    b->color_weight = (float*)malloc(sizeof(float) * 256 * cn);
    b->space_weight = (float*)malloc(sizeof(float) * (size_t)d * d);
    b->space_ofs = (int*)malloc(sizeof(int) * (size_t)d * d);
    for (int i = 0; i < 256 * cn; ++i) b->color_weight[i] = (float)exp(i * i * gauss_color_coeff);
    int maxk = 0;
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            const double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            b->space_weight[maxk] = (float)exp(r * r * gauss_space_coeff);
            b->space_ofs[maxk++] = (i * tw + j) * cn;
        }
    b->maxk = maxk;
}

static void bilateral_release(bilateral_setup* b)
{
    free(b->space_ofs);
    free(b->space_weight);
    free(b->color_weight);
    free(b->temp);
}

void orc_bilateral_filter_u8(const uint8_t* src, int64_t src_pitch, int32_t rows, int32_t cols, int32_t d,
                             double sigma_color, double sigma_space, int32_t order, uint8_t* dst, int64_t dst_pitch)
{
    if (rows <= 0 || cols <= 0) return;
    bilateral_setup b;
    bilateral_prepare(&b, src, src_pitch, rows, cols, 1, d, sigma_color, sigma_space);
    const int radius = b.radius, tw = b.tw, maxk = b.maxk;
    const float *color_weight = b.color_weight, *space_weight = b.space_weight;
    const int* space_ofs = b.space_ofs;
    for (int i = 0; i < rows; ++i) {
        const uint8_t* sptr = b.temp + (size_t)(i + radius) * tw + radius;
        uint8_t* dptr = dst + (int64_t)i * dst_pitch;
        for (int j = 0; j < cols; ++j) {
            float sum = 0, wsum = 0;
            const int val0 = sptr[j];
            int k = 0;
            if (order == ORC_BILATERAL_SSE3) {
                for (; k <= maxk - 4; k += 4) {
                    float w[4], vw[4];
                    for (int q = 0; q < 4; ++q) {
                        const int val = sptr[j + space_ofs[k + q]];
                        w[q] = color_weight[abs(val - val0)] * space_weight[k + q]; /* mulps(_cw, _sw) */
                        vw[q] = w[q] * (float)val;                                  /* mulps(_w, _valF) */
                    }
                    /* hadd(_w,_cw) -> hadd(.,.): [ (w0+w1)+(w2+w3), (vw0+vw1)+(vw2+vw3) ] */
                    const float ws = (w[0] + w[1]) + (w[2] + w[3]);
                    const float vs = (vw[0] + vw[1]) + (vw[2] + vw[3]);
                    sum += vs;
                    wsum += ws;
                }
            }
            for (; k < maxk; ++k) {
                const int val = sptr[j + space_ofs[k]];
                const float w = space_weight[k] * color_weight[abs(val - val0)];
                sum += val * w;
                wsum += w;
            }
            dptr[j] = (uint8_t)lrintf(sum / wsum); /* cvRound: round half to even */
        }
    }
    bilateral_release(&b);
}

/* The same call on CV_8UC3 (the cn == 3 branch of BilateralFilter_8u_Invoker): one weight per neighbour from the summed
 * channel differences, three weighted sums, `wsum = 1.f / wsum; b0 = cvRound(sum_b * wsum)`.  Not on the hot path (the
 * reference only filters disparities); it exists because the reference HOLDS outputs of this branch, and it shares its
 * border, tables, neighbour order and four-at-a-time grouping with the one-channel branch above. */
void orc_bilateral_filter_u8c3(const uint8_t* src, int64_t src_pitch, int32_t rows, int32_t cols, int32_t d,
                               double sigma_color, double sigma_space, int32_t order, uint8_t* dst, int64_t dst_pitch)
{
    if (rows <= 0 || cols <= 0) return;
    bilateral_setup b;
    bilateral_prepare(&b, src, src_pitch, rows, cols, 3, d, sigma_color, sigma_space);
    const int radius = b.radius, tw = b.tw, maxk = b.maxk;
    const float *color_weight = b.color_weight, *space_weight = b.space_weight;
    const int* space_ofs = b.space_ofs;
    for (int i = 0; i < rows; ++i) {
        const uint8_t* sptr = b.temp + ((size_t)(i + radius) * tw + radius) * 3;
        uint8_t* dptr = dst + (int64_t)i * dst_pitch;
        for (int j = 0; j < cols * 3; j += 3) {
            float sum_b = 0, sum_g = 0, sum_r = 0, wsum = 0;
            const int b0 = sptr[j], g0 = sptr[j + 1], r0 = sptr[j + 2];
            int k = 0;
            if (order == ORC_BILATERAL_SSE3) {
                for (; k <= maxk - 4; k += 4) {
                    float w[4], bw[4], gw[4], rw[4];
                    for (int q = 0; q < 4; ++q) {
                        const uint8_t* sk = sptr + j + space_ofs[k + q];
                        const int bb = sk[0], gg = sk[1], rr = sk[2];
                        w[q] = color_weight[abs(bb - b0) + abs(gg - g0) + abs(rr - r0)] * space_weight[k + q];
                        bw[q] = (float)bb * w[q];
                        gw[q] = (float)gg * w[q];
                        rw[q] = (float)rr * w[q];
                    }
                    /* hadd(_w,_b), hadd(_g,_r), hadd of the two: every sum as (a0+a1)+(a2+a3) */
                    wsum += (w[0] + w[1]) + (w[2] + w[3]);
                    sum_b += (bw[0] + bw[1]) + (bw[2] + bw[3]);
                    sum_g += (gw[0] + gw[1]) + (gw[2] + gw[3]);
                    sum_r += (rw[0] + rw[1]) + (rw[2] + rw[3]);
                }
            }
            for (; k < maxk; ++k) {
                const uint8_t* sk = sptr + j + space_ofs[k];
                const int bb = sk[0], gg = sk[1], rr = sk[2];
                const float w = space_weight[k] * color_weight[abs(bb - b0) + abs(gg - g0) + abs(rr - r0)];
                sum_b += bb * w;
                sum_g += gg * w;
                sum_r += rr * w;
                wsum += w;
            }
            wsum = 1.f / wsum;
            dptr[j] = (uint8_t)lrintf(sum_b * wsum);
            dptr[j + 1] = (uint8_t)lrintf(sum_g * wsum);
            dptr[j + 2] = (uint8_t)lrintf(sum_r * wsum);
        }
    }
    bilateral_release(&b);
}

/* Pose::getMean / Pose::getVariance, pose_functions.cpp:987-1028 (planeFitted == false) */
double orc_disparity_variance(const uint8_t* disp, int64_t pitch, int32_t rows, int32_t cols, int32_t bounding_box,
                              int32_t cols_start_aft_cutout, double min_disparity)
{
    const int bb = bounding_box, cs = cols_start_aft_cutout;
    double sum = 0.0;
    for (int y = bb; y < rows - bb; ++y)
        for (int x = cs; x < cols - bb; ++x) {
            const double v = (double)disp[(int64_t)y * pitch + x];
            if (v > min_disparity) sum += v;
        }
    const double mean = sum / ((rows - 2 * bb) * (cols - bb - cs));
    double temp = 0;
    for (int y = bb; y < rows - bb; ++y)
        for (int x = cs; x < cols - bb; ++x) {
            const double v = (double)disp[(int64_t)y * pitch + x];
            if (v > min_disparity) temp += (v - mean) * (v - mean);
        }
    return temp / ((rows - 2 * bb) * (cols - bb - cs) - 1);
}
