// o3dr_oracle_stdsort.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see o3dr_oracle.h).
//
// PCL 1.8's VoxelGrid sorts its (idx, cloud_point_index) records with
//     std::sort(index_vector.begin(), index_vector.end(), std::less<cloud_point_index_idx>());
// where operator< compares idx only [PCL 1.8 filters/voxel_grid.h, struct cloud_point_index_idx;
// used from pose_functions.cpp:1700].  std::sort is not stable, so the order in which the points of
// one voxel are then summed is whatever libstdc++'s introsort leaves.  This file reproduces that
// call on the same record layout so the oracle can show how far the reference's own (arbitrary)
// summation order moves a centroid away from the canonical ascending-index order.
#include <algorithm>
#include <cstdint>
#include <functional>
#include <vector>

#include "o3dr_oracle.h"

namespace {
struct cloud_point_index_idx {
    unsigned int idx;
    unsigned int cloud_point_index;
    cloud_point_index_idx(unsigned int idx_, unsigned int cloud_point_index_)
        : idx(idx_), cloud_point_index(cloud_point_index_) {}
    bool operator<(const cloud_point_index_idx& p) const { return (idx < p.idx); }
};
}  // namespace

extern "C" void orc_stdsort_pairs(uint32_t* idx, uint32_t* point_index, int64_t n)
{
    std::vector<cloud_point_index_idx> index_vector;
    index_vector.reserve(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; ++i) index_vector.push_back(cloud_point_index_idx(idx[i], point_index[i]));
    std::sort(index_vector.begin(), index_vector.end(), std::less<cloud_point_index_idx>());
    for (int64_t i = 0; i < n; ++i) {
        idx[i] = index_vector[static_cast<size_t>(i)].idx;
        point_index[i] = index_vector[static_cast<size_t>(i)].cloud_point_index;
    }
}
