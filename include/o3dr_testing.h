/*
 * o3dr_testing.h — TEST-ONLY entry points of libo3dr.  Not part of the installed surface of include/o3dr.h: nothing a
 * caller of the reconstruction path needs is declared here, and every function below refuses to act
 * (O3DR_ERR_INVALID_ARG) unless the context was created with the environment variable O3DR_TEST_HOOKS=1.
 */
#ifndef O3DR_TESTING_H
#define O3DR_TESTING_H
#include "o3dr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook for the gather guards: the next o3dr_voxel_grid / o3dr_downsample_pt_cloud / o3dr_finalize of this
 * context finds one of its sorted payloads pointing outside the cloud, as a bookkeeping error upstream would leave it.
 * That call must return O3DR_ERR_INTERNAL with *n_out = 0 (never a GPU fault), and the context stays usable. */
int o3dr_test_corrupt_next_gather(o3dr_ctx* ctx);

/* The mean neighbour distances (pcl::StatisticalOutlierRemoval's `distances`, one per input point, input order) the
 * last o3dr_statistical_outlier_removal of this context computed for its n_in points: the inlier set alone says little
 * about an outlier's own distance, so the parity tests compare these with the oracle's bit for bit.  out: n floats in
 * host memory; n must not exceed that call's n_in. */
int o3dr_test_sor_distances(o3dr_ctx* ctx, float* out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* O3DR_TESTING_H */
