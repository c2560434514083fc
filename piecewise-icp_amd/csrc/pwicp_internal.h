/* pwicp_internal.h - measurement and test hooks of libpwicp.so that are NOT part of the public surface (include/pwicp.h).
 * They are exported from the library (bench.py, tests/ and tools/ reach them through ctypes) but no client code should bind them:
 * signatures may change from round to round.  (Moved out of include/pwicp.h in round 6) */
#ifndef PWICP_INTERNAL_H
#define PWICP_INTERNAL_H
#include "pwicp.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook of the id file's acceptance rules (no GPU, no RCCL): op 0 writes an id file with this process's job token dated age_s
 * seconds back (returns 1), op 1 reads it as a rank != 0 would: 1 accepted, 0 rejected.  A rank accepts a file only if it carries its
 * launch's token ($PWICP_JOB_ID / $TORCHELASTIC_RUN_ID / MASTER_ADDR / MASTER_PORT / WORLD_SIZE), is younger than 600 s AND was not
 * written more than 30 s before the reading process started (an earlier launch with the same token that was killed). */
PWICP_API int  pwicp_comm_debug_id_file(const char* path, int op, long age_s);

/* ---- one dense NN launch on resident data, for roofline measurement (bench.py) --------------- */
/* Runs the dense 1-NN kernel for all source patch points of `pair` against cloud1 `n_launches`
 * times on the pair's stream and returns the mean HIP-event time per launch, the number of queries
 * per launch, and Kbar (mean target points in the 27-cell stencil of a query). */
PWICP_API int pwicp_pair_bench_dense_nn(pwicp_pair* pair, int n_launches, double* ms_per_launch,
                                        long long* n_queries, double* kbar, double* cell_edge);

/* The dense 1-NN search of calPercentileDistBetween2PC (CommonFunc.cpp:266-281) by itself: the squared distance of EVERY source
 * patch point (pwicp_pair_num_patch_points: tot2 of them, in the order of the source patch arrays, current positions) to its nearest
 * point of cloud1 - the search the loop runs on the points of its stable patches, here with every patch taken as stable.
 * far_group: 0 / 1 = the far queries inside the search's own launch / on the launch that puts eight lanes on each; -1 = as the loop
 * would choose for a first search.  d2_out: tot2 floats.  (Parity tests compare it with a brute-force search.) */
PWICP_API int pwicp_pair_dense_distances(pwicp_pair* pair, int far_group, float* d2_out);

#ifdef __cplusplus
}
#endif
#endif /* PWICP_INTERNAL_H */
