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

/* o3dr_merge_partitioned with W > 1 ranks on ONE GPU.  RCCL refuses two ranks on one device, so the multi-rank paths of
 * the exchange (slice sizes, segment order, failure agreement, statistics) would otherwise only ever run with one rank:
 * a LOCAL communicator connects n_ranks contexts of one process - one host thread each, all on the same device - through
 * device-to-device copies and a host barrier (a rank that does not show up within 30 s breaks it: O3DR_ERR_PEER).  The
 * protocol code is the one o3dr_merge_partitioned runs over RCCL; only the two transport primitives differ.
 * o3dr_test_fail_at: step `point` of this context's next exchange fails on this rank as an allocation would
 * (1 header, 2 partition, 3 buffers before the all-to-all, 4 local merge). */
int o3dr_test_local_comm_create(int32_t n_ranks, void** comm_out);
int o3dr_test_local_comm_destroy(void* comm);
int o3dr_test_merge_partitioned_local(o3dr_ctx* ctx, void* local_comm, int32_t rank, int32_t gather_result, o3dr_point* out,
                                      int64_t out_capacity, int64_t* n_out, int64_t* n_total, uint32_t* status, int32_t mem);
int o3dr_test_fail_at(o3dr_ctx* ctx, int32_t point);

#ifdef __cplusplus
}
#endif
#endif /* O3DR_TESTING_H */
