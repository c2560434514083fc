/*
 * pwicp.h — C ABI of libpwicp.so: the MI355X-native (HIP, gfx950) implementation of the
 * Piecewise-ICP fine-registration loop.
 *
 * This is the drop-in boundary for the hot path of yihui4d/Piecewise-ICP.  Every entry point
 * names the reference interface it replaces (file:line in the reference tree; R = src/
 * Registration.cpp, C = src/CommonFunc.cpp, S = src/Segmentation.cpp, R.h/S.h/C.h = include/).
 * Plain pointers and sizes only — no PCL, Eigen or torch types.  The PCL-typed C++ facade
 * with the reference's own signatures (Registration.h) lives in include/pwicp/ and forwards
 * here.
 *
 * Conventions
 *  - point arrays are pcl::PointXYZ-compatible: 16 bytes per point, floats x,y,z,pad;
 *    "xyz4" below.  Normals likewise (nx,ny,nz,pad).
 *  - all pointers are HOST pointers unless the name says `_dev`; a pwicp_pair keeps its data
 *    resident in HBM between calls.
 *  - 4x4 matrices are row-major float[16]; 6x6 VCM is row-major double[36]
 *    (parameter order Rx,Ry,Rz,tx,ty,tz as R.cpp:1286).
 *  - functions return PWICP_OK (0) or a negative pwicp_status; pwicp_last_error() gives text.
 *    Nothing in this library calls exit() (the reference does: R.cpp:728-731, 864-867).
 *  - there is NO CPU fallback: every compute entry point fails with PWICP_E_NO_DEVICE when no
 *    HIP device is usable.
 */
#ifndef PWICP_H
#define PWICP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PWICP_API __attribute__((visibility("default")))

typedef enum {
    PWICP_OK = 0,
    PWICP_E_NO_DEVICE = -1,       /* no usable HIP device / HIP runtime error */
    PWICP_E_INVALID = -2,         /* bad argument (null pointer, negative size, non-finite input) */
    PWICP_E_TOO_FEW_PATCHES = -3, /* < 4 source patches            (R.cpp:728-731) */
    PWICP_E_TOO_FEW_STABLE = -4,  /* < 4 stable patches            (R.cpp:864-867) */
    PWICP_E_NOMEM = -5,
    PWICP_E_INTERNAL = -6,
    PWICP_E_NOT_CONVERGED = -7    /* PWICP_MAX_OUTER outer iterations without reaching Stage 3 (the reference would keep
                                     iterating, R.cpp:680): T16 is the last estimate, VCM was never computed */
} pwicp_status;

typedef struct pwicp_context pwicp_context;   /* one per GPU / per thread; replaces the reference's
                                                 module globals g_toStage2/3, g_isVis (R.cpp:11-14) */
typedef struct pwicp_pair pwicp_pair;         /* one target/source pair resident in HBM */

/* ---- context ------------------------------------------------------------------------------ */
PWICP_API int         pwicp_create(pwicp_context** ctx, int device_id);
PWICP_API void        pwicp_destroy(pwicp_context* ctx);
PWICP_API int         pwicp_context_device(const pwicp_context* ctx);   /* HIP device ordinal of a context (-1: null) */
PWICP_API const char* pwicp_last_error(const pwicp_context* ctx);
PWICP_API const char* pwicp_version(void);
/* number of visible HIP devices (0 when none / no runtime) */
PWICP_API int         pwicp_device_count(void);

/* ---- building blocks (each replaces one reference / PCL call site) ------------------------- */

/* Exact 1-NN of every query in the target set; all correspondences kept.
 * Replaces pcl::registration::CorrespondenceEstimation<PointXYZ,PointXYZ>::
 * determineCorrespondences(corr, DBL_MAX) as called at R.cpp:737-747, 1293-1297, 597-601 and
 * C.cpp:269-273: index_match[i] = argmin_j of the FLOAT expression ((dx*dx)+dy*dy)+dz*dz,
 * sq_distance[i] = that float value (pcl::Correspondence::distance is the squared distance).
 * Ties resolve to the lowest target index. */
PWICP_API int pwicp_nn_search(pwicp_context* ctx, const float* target_xyz4, int n_target,
                              const float* query_xyz4, int n_query,
                              int32_t* index_match, float* sq_distance);

/* p-th percentile of the 1-NN distances cloud2 -> cloud1: element int(n*percentile) of the
 * ascending sqrt(float d2) list.  Replaces calPercentileDistBetween2PC (C.cpp:266-281 with
 * C.cpp:145-179; decl C.h). */
PWICP_API int pwicp_percentile_dist(pwicp_context* ctx, const float* cloud1_xyz4, int n1,
                                    const float* cloud2_xyz4, int n2, float percentile,
                                    double* dist_out);

/* Fraction of cloud2 points whose 1-NN distance to cloud1 is < DTinit.
 * Replaces calOverlapRatioByC2Cdist (R.cpp:593-614; decl R.h:116-129). */
PWICP_API int pwicp_overlap_ratio(pwicp_context* ctx, const float* cloud1_xyz4, int n1,
                                  const float* cloud2_xyz4, int n2, float DTinit, float* ratio_out);

/* Patch plane normals: for patch i (points patch_xyz4[offsets[i]..offsets[i+1])) the smallest
 * eigenvector of the float single-pass covariance.  Replaces calPatchNormal (C.cpp:284-333 ->
 * pcl::computePointNormal) for all patches at once, i.e. the normal part of
 * generateCentroidCloudWithPatchNormals (C.cpp:357-382).  ok[i] = calPatchNormal's return value;
 * on failure the normal is (0,0,1). */
PWICP_API int pwicp_patch_normals(pwicp_context* ctx, const float* patch_xyz4, const int32_t* offsets,
                                  int n_patches, float* normals4, uint8_t* ok);

/* Patch extraction, 2-sigma refinement, selection, centroids, boundary points, sigmas from a
 * supervoxel labelling.  Replaces the body of PatchGenerationAndRefinement after the
 * segmentation call (S.cpp:97-150: PatchRefinement S.cpp:195-228, calPatchFeature S.cpp:231-257,
 * calPatchCTandBP S.cpp:260-303) plus calBPandCTSTD / calPatchSTD (S.cpp:306-321, C.cpp:336-354).
 * Two-call protocol: call with patch_xyz4 == NULL to get *n_patches and *n_patch_points, then
 * with buffers: patch_xyz4 [n_patch_points*4], offsets [n_patches+1], src_index [n_patch_points]
 * (may be NULL), centroid_xyz4 [n_patches*4], boundary_xyz4 [n_patches*24] (order Xmax,Xmin,
 * Ymax,Ymin,Zmax,Zmin), std_bp, std_ct [n_patches]. */
PWICP_API int pwicp_select_patches(pwicp_context* ctx, const float* cloud_xyz4, int n,
                                   const int32_t* labels, int n_supervoxels,
                                   int* n_patches, int* n_patch_points,
                                   float* patch_xyz4, int32_t* offsets, int32_t* src_index,
                                   float* centroid_xyz4, float* boundary_xyz4,
                                   float* std_bp, float* std_ct);

/* Centroid (compute3DCentroid, float), the six boundary points (Xmax,Xmin,Ymax,Ymin,Zmax,Zmin) and the sigmas of GIVEN
 * patches: calPatchCTandBP (S.cpp:260-303), calPatchSTD (C.cpp:336-354; decl C.h:150) = std_bp, and calBPandCTSTD
 * (S.cpp:306-321; decl S.h:451-452: std_ct = std_bp / N) for all patches at once.  Any output may be NULL. */
PWICP_API int pwicp_patch_stats(pwicp_context* ctx, const float* patch_xyz4, const int32_t* offsets, int n_patches,
                                float* centroid_xyz4, float* boundary_xyz4, float* std_bp, float* std_ct);

/* Point-to-plane ICP between centroid clouds with normals.  Replaces P2PICPwithPatchNormal
 * (R.cpp:1255-1269; decl R.h:213-214) = pcl::IterativeClosestPointWithNormals<PointNormal,
 * PointNormal>::align with TransformationEpsilon 1e-8, EuclideanFitnessEpsilon euclid_eps,
 * MaximumIterations 100 (TransformationEstimationPointToPlaneLLS + DefaultConvergenceCriteria).
 * T16: final transformation; n_iterations (optional). */
PWICP_API int pwicp_p2p_icp(pwicp_context* ctx, const float* target_xyz4, const float* target_normal4,
                            int n_target, const float* source_xyz4, const float* source_normal4,
                            int n_source, double euclid_eps, float* T16, int* n_iterations);

/* Variance-covariance matrix of the 6 transformation parameters.  Replaces calTransParaVCM
 * (R.cpp:1273-1343; decl R.h:227-229). */
PWICP_API int pwicp_trans_para_vcm(pwicp_context* ctx, const float* target_xyz4,
                                   const float* target_normal4, int n_target,
                                   const float* source_stable_xyz4, int n_source, double* VCM36);

/* ---- the loop: Piecewise_ICP (R.cpp:618-700; decl R.h:149-153) ------------------------------ */
#define PWICP_MAX_OUTER 256

typedef struct {
    float Res1, Res2;            /* average point spacing of cloud1 / cloud2            */
    float SVRes1, SVRes2;        /* patch (supervoxel) size actually used (R.cpp:635-640) */
    int   isManualDTinit;        /* 0: DTinit = 3 * p75 dense NN distance (R.cpp:627-630)  */
    float DTinit, DTmin;
} pwicp_params;

typedef struct {
    int    status;                        /* pwicp_status of the run                         */
    int    n_outer;                       /* outer (Piecewise-ICP) iterations                */
    float  T16[16];                       /* accumulated transMat (R.cpp:687)                */
    double VCM[36];                       /* R.cpp:958-961                                   */
    float  DTseries[PWICP_MAX_OUTER + 1]; /* R.cpp:675-676, 688                              */
    int    n_inner[PWICP_MAX_OUTER];      /* inner ICP iterations per outer iteration        */
    int    n_stable[PWICP_MAX_OUTER];     /* stable source patches                           */
    int    n_stable_pts[PWICP_MAX_OUTER]; /* points of the stable patches (|stablePC2|)      */
    float  LoDmin[PWICP_MAX_OUTER];
    float  maxBB[PWICP_MAX_OUTER];
    double d75[PWICP_MAX_OUTER];          /* Stage-1 percentile distance, -1 if not computed */
    float  Tk[PWICP_MAX_OUTER][16];       /* per-iteration transMatICP                       */
    long long n_corr;                     /* completed 1-NN queries inside the loop */
    long long n_corr_dense;               /* of which dense Stage-1 queries                  */
    long long n_inner_total;
    double t_loop_ms;                     /* host wall time of the loop                      */
    double t_dense_nn_ms;                 /* HIP-event time of the dense NN kernel launches  */
    int    n_dense_nn_launches;
    double t_inner_ms;                    /* HIP-event time of the inner-ICP launches        */
    double dense_kbar;                    /* mean target points examined per dense query     */
    int32_t dense_rows;                   /* 0: disc-pruned dense search (default); 9 / 3: 27-cell stencil on cells / columns */
    int32_t n_dense_bounded;              /* dense queries (of n_corr_dense) that were NOT searched to the end: their distance was proved to lie
                                             above the percentile the search feeds (C.cpp:266-281 returns Dist75 only) - far queries of a
                                             misaligned pair; the value written for them is an upper bound.  n_corr counts the reference-
                                             defined queries; completed ones = n_corr - n_dense_bounded */
} pwicp_result;

/* Uploads both clouds and their supervoxel labellings, runs patch selection/statistics and builds
 * the static target-side search structures in HBM.  Everything PatchGenerationAndRefinement
 * (S.cpp:11-192) produces after the segmentation call + calBPandCTSTD (R.cpp:660-664). */
PWICP_API int pwicp_pair_create(pwicp_context* ctx,
                                const float* cloud1_xyz4, int n1, const int32_t* labels1, int nsv1,
                                const float* cloud2_xyz4, int n2, const int32_t* labels2, int nsv2,
                                const pwicp_params* params, pwicp_pair** pair);
/* Same, from already selected patches (the arrays pwicp_select_patches returns). */
PWICP_API int pwicp_pair_create_from_patches(pwicp_context* ctx,
                                const float* cloud1_xyz4, int n1,
                                const float* patch1_xyz4, const int32_t* off1, int m1,
                                const float* cloud2_xyz4, int n2,
                                const float* patch2_xyz4, const int32_t* off2, int m2,
                                const pwicp_params* params, pwicp_pair** pair);
/* Same, with the per-patch arrays of the CALLER instead of the ones recomputed from the patch points — what the arguments of
 * PwICP_singleIteration (R.h:181-188) carry: CTcloud (m x 16 B), BPcloud (6 m), the sigmas of calBPandCTSTD.  Any of the
 * eight arrays may be NULL (= computed).  Needed because the loop transforms the source centroids and boundary points
 * themselves (R.cpp:946-949): after the first iteration they are no longer the centroids of the (transformed) patch points
 * to the last bit. */
PWICP_API int pwicp_pair_create_from_arrays(pwicp_context* ctx,
                                const float* cloud1_xyz4, int n1, const float* patch1_xyz4, const int32_t* off1, int m1,
                                const float* centroid1_xyz4, const float* boundary1_xyz4, const float* std_bp1, const float* std_ct1,
                                const float* cloud2_xyz4, int n2, const float* patch2_xyz4, const int32_t* off2, int m2,
                                const float* centroid2_xyz4, const float* boundary2_xyz4, const float* std_bp2, const float* std_ct2,
                                const pwicp_params* params, pwicp_pair** pair);
/* The static target side of a pair as a handle of its own: cloud, patches, normals, search grids.  Every pair of a
 * Direct2Ref series registers against the same target scan (R.cpp:94-103; the reference rebuilds all of it per pair,
 * R.cpp:653): build it once, create the pairs with it.  params->Res1 / SVRes1 of those pairs must be the target's.
 * The target must outlive the pairs created with it. */
typedef struct pwicp_target pwicp_target;
PWICP_API int pwicp_target_create(pwicp_context* ctx, const float* cloud1_xyz4, int n1, const int32_t* labels1,
                                  int n_supervoxels1, float Res1, float SVRes1, pwicp_target** out);
PWICP_API void pwicp_target_destroy(pwicp_target* target);
PWICP_API int pwicp_pair_create_with_target(pwicp_target* target, const float* cloud2_xyz4, int n2,
                                            const int32_t* labels2, int n_supervoxels2, const pwicp_params* params,
                                            pwicp_pair** out);
/* The same on a context (= stream) of the caller's choice on the target's device: the target is read-only once built, so the
 * pairs of a streamed series can alternate between two contexts and the upload / patch selection of epoch k + 1 (one host
 * thread) runs beside the loop of epoch k (another).  ctx == the target's own context: the call above. */
PWICP_API int pwicp_pair_create_with_target_on(pwicp_context* ctx, pwicp_target* target, const float* cloud2_xyz4, int n2,
                                               const int32_t* labels2, int n_supervoxels2, const pwicp_params* params,
                                               pwicp_pair** out);
PWICP_API void pwicp_pair_destroy(pwicp_pair* pair);
PWICP_API int  pwicp_pair_num_patches(const pwicp_pair* pair, int* m1, int* m2);
/* Puts the source side back into its uploaded state (the loop transforms it in place, R.cpp:943-954) so that the same pair
 * can be registered again.  Nothing is copied here: the next pwicp_pair_run reads the pristine device copies until its first
 * transform has rewritten the working arrays; any other consumer of the working arrays restores them on demand. */
PWICP_API int pwicp_pair_reset(pwicp_pair* pair);
/* The while-loop of Piecewise_ICP (R.cpp:680-694) = repeated PwICP_singleIteration
 * (R.cpp:704-972; decl R.h:181-188), entirely on the device. */
PWICP_API int pwicp_pair_run(pwicp_pair* pair, pwicp_result* result);
/* pwicp_pair_run for n INDEPENDENT pairs side by side - the iterations of the reference's pair loop (R.cpp:89-187) share nothing:
 * one host thread per CONTEXT among the pairs inside the call, the pairs of one context one after the other on its thread, in the
 * order given (so K contexts - pwicp_pair_create_with_target_on - and any number of pairs dealt to them are K registrations in flight
 * with no barrier between them).  reset_first != 0: pwicp_pair_reset before each run.  results[k] is bit for bit what
 * pwicp_pair_run(pairs[k]) gives alone; returns the first status that is not PWICP_OK; the same pair twice is PWICP_E_INVALID.
 * One registration is a chain of dependent short launches: four in flight cost about half the time each (GPU_MAX_HW_QUEUES=8,
 * INTEGRATION.md). */
PWICP_API int pwicp_pairs_run_concurrent(pwicp_pair* const* pairs, int n, pwicp_result* results, int reset_first);
/* ONE outer iteration: PwICP_singleIteration (R.cpp:704-972; decl R.h:181-188) on the pair's resident data.  The
 * caller owns what the reference keeps between calls: currDT, BBchange_1/2 (reference parameters of R.h:187) and the two
 * stage flags (the reference's module globals g_toStage2 / g_toStage3, R.cpp:11-14), and runs the loop of Piecewise_ICP
 * (R.cpp:680-694) itself:   st = {DTinit, 0, 0, 0, 0};  while (!st.toStage3) { pwicp_pair_step(pair, &st);  T = st.T16 * T; }
 * Source arrays are transformed in place (R.cpp:943-954).  VCM is written by the call that sets toStage3 (R.cpp:958-961).
 * Stepping a freshly created / reset pair to Stage 3 gives bit for bit the result of pwicp_pair_run. */
typedef struct {
    float currDT;                    /* in/out */
    float BBchange_1, BBchange_2;    /* in/out */
    int   toStage2, toStage3;        /* in/out */
    int   status;                    /* out: pwicp_status of this call */
    float T16[16];                   /* out: transMatICP of this iteration */
    double VCM[36];                  /* out: valid once toStage3 is set */
    int   n_stable, n_stable_pts, n_inner;
    float LoDmin, maxBB;
    double d75;                      /* Stage-1 percentile distance, -1 if not computed */
} pwicp_step;
PWICP_API int pwicp_pair_step(pwicp_pair* pair, pwicp_step* step);
/* DTinit when it is not given in the configuration: 3 x the 75th percentile 1-NN distance cloud2 -> cloud1 (R.cpp:626-631) */
PWICP_API int pwicp_pair_auto_dtinit(pwicp_pair* pair, float* DTinit);
/* Copies the current (transformed) source cloud back: cloud2 after the loop (R.cpp:943-945). */
PWICP_API int pwicp_pair_download_source(pwicp_pair* pair, float* cloud2_xyz4);

/* Everything PwICP_singleIteration transforms in place (R.cpp:943-954) — cloud2 (n2), CTcloud2 (m2), BPcloud2 (6 m2), the
 * source patch points (pwicp_pair_num_patch_points, concatenated in patch order); any pointer may be NULL. */
PWICP_API int pwicp_pair_download_state(pwicp_pair* pair, float* cloud2_xyz4, float* centroid2_xyz4, float* boundary2_xyz4,
                                        float* patch2_xyz4);
PWICP_API int pwicp_pair_num_patch_points(const pwicp_pair* pair, int* n_patch_points1, int* n_patch_points2);

/* ---- host-side setup stages and the reference's file-in / file-out entry points -------------------------------- */
/* The stages the reference runs before the loop, exported so that the two entry points below
 * are a complete drop-in.  The *_dev functions (and pwicp_knn) run on the GPU and are what the entry points use; the
 * functions without a context are host implementations of the SAME stage with identical output, kept as separate,
 * explicitly named entry points (the tests cross-check the two; nothing ever selects them as a fallback - with one
 * documented exception: a supervoxel fusion whose search queue outgrows the wavefront's LDS queue hands that one pass to
 * the serial host code, same labels). */

/* Supervoxel label of every point.  Replaces the first half of PatchGenerationAndRefinement (S.cpp:18-68):
 * k-NN (knn = 45 in the reference, C.h:41, query point included), PCA normals, boundary-preserving supervoxel
 * segmentation at `sv_resolution`. */
PWICP_API int pwicp_frontend_segment(const float* cloud_xyz4, int n, float sv_resolution, int knn,
                                     int32_t* labels, int* n_supervoxels);
/* Exact k nearest neighbours of every point within its own cloud, on the GPU: neighbors[i*k .. i*k+k) in ascending
 * order of the double squared distance, ties by index, the point itself first.  Replaces the
 * cl::KDTree::FindKNearestNeighbors loop of S.cpp:30-41.  cell_edge <= 0: estimated from the cloud. */
PWICP_API int pwicp_knn(pwicp_context* ctx, const float* cloud_xyz4, int n, int k, float cell_edge, int32_t* neighbors);
/* pwicp_frontend_segment on the GPU (csrc/frontend.hip): k-NN graph, neighbourhood scatter, supervoxel fusion and boundary
 * refinement as speculative fixed points over the reference's serial visiting order - identical labels.  Host: the
 * closed-form eigen step of the normals.  Work buffers stay with the context (grow-only) until pwicp_destroy.
 * point_spacing <= 0: estimated.  $PWICP_FRONTEND=host: the serial host passes behind the GPU k-NN graph. */
PWICP_API int pwicp_frontend_segment_dev(pwicp_context* ctx, const float* cloud_xyz4, int n, float sv_resolution,
                                         int knn, float point_spacing, int32_t* labels, int* n_supervoxels);
/* PCpreprocessing(cloud, out, true, voxel_size, sor_k, sor_mult) (C.cpp:423-452). out_xyz4 holds n points.
 * As pcl::VoxelGrid does, a leaf size whose voxel indices would overflow int32 makes the voxel stage a pass-through (warning
 * on stderr) and the SOR stage runs on the full cloud. */
PWICP_API int pwicp_preprocess(const float* cloud_xyz4, int n, float voxel_size, int sor_k, double sor_mult,
                               float* out_xyz4, int* n_out);
/* pwicp_preprocess on the GPU (identical output): voxel keys + stable radix sort + per-voxel centroids, then the
 * SOR mean-distance statistic from an exact float-metric k-NN on the uniform grid. */
PWICP_API int pwicp_preprocess_dev(pwicp_context* ctx, const float* cloud_xyz4, int n, float voxel_size, int sor_k,
                                   double sor_mult, float* out_xyz4, int* n_out);
/* SORfilter (C.cpp:441-452; decl C.h:209) = PCpreprocessing(..., isDownSamp = false, ...): pcl::StatisticalOutlierRemoval
 * (mean_k = sor_k, stddev multiplier sor_mult) alone.  _dev: the k-NN statistic on the GPU (identical output); spacing_hint > 0
 * sizes the search grid only (<= 0: estimated), it never changes the result. */
PWICP_API int pwicp_sor_filter(const float* cloud_xyz4, int n, int sor_k, double sor_mult, float* out_xyz4, int* n_out);
PWICP_API int pwicp_sor_filter_dev(pwicp_context* ctx, const float* cloud_xyz4, int n, int sor_k, double sor_mult,
                                   float spacing_hint, float* out_xyz4, int* n_out);
/* calPCresolution (C.cpp:239-263) */
PWICP_API float pwicp_pc_resolution(const float* cloud_xyz4, int n);
/* the same value with the nearest-neighbour distances computed on the GPU */
PWICP_API int pwicp_pc_resolution_dev(pwicp_context* ctx, const float* cloud_xyz4, int n, float* resolution);

/* The reference's exported functions, same signatures (include/Registration.h:36, 49; python/main.py:15-18).
 * Device: $PWICP_DEVICE or $LOCAL_RANK (default 0).  Never exit(): false on any failure. */
PWICP_API bool PiecewiseICP_pair_call(const char* confile, const char* outfile);
PWICP_API bool PiecewiseICP_4D_call(const char* confile, int startEpoch, int epochNum, int pairMode, float overlapThd);

/* ---- the remaining functions of the reference's Registration.h / CommonFunc.h as C entry points ---------------------------
 * (include/pwicp/Registration.h wraps them in the reference's exact PCL / Eigen signatures) */
/* Piecewise_ICP_4D (R.cpp:402-548; decl R.h:74-78): PCpreprocessing with SOR multiplier 5.0, reduction by the target
 * centroid, Piecewise_ICP, T_final = S^-1 T S, parameters (Rx,Ry,Rz [gon], tx,ty,tz [m]); outfileIdx != NULL: writes
 * "<outfileIdx>TransMatrix.txt" (R.cpp:492-540). */
PWICP_API int pwicp_piecewise_icp_4d(pwicp_context* ctx, const float* cloud1_xyz4, int n1, const float* cloud2_xyz4, int n2,
                                     int isSetResSVsize, float Res1, float Res2, float SVsize1, float SVsize2, int isManualDTinit,
                                     float DTinit, float DTmin, const char* outfileIdx, float* transMat16, float* transPara6,
                                     double* VCM36);
/* calAdaptivePairSequence (R.cpp:552-589; decl R.h:93-94): targets[k] = target of source k+1, relative to startEpoch,
 * n_files - startEpoch - 1 entries; adaptivePairFile (may be NULL) receives the "source target" lines. */
PWICP_API int pwicp_adaptive_pair_sequence(pwicp_context* ctx, const char* const* fileNameList, int n_files, int startEpoch,
                                           float DTinit, float ratioThd, int32_t* targets, const char* adaptivePairFile);
/* calTransToReferenceEpoch (R.cpp:977-1153; decl R.h:127-129).  Optional outputs hold epochNum entries (16 / 36 values each). */
PWICP_API int pwicp_trans_to_reference_epoch(const char* transMatFile, int pairMode, const char* adaptivePairFile, int epochNum,
                                             const char* transMat2RefFile, const char* transPara2RefFile, int32_t* timeStamp,
                                             float* allTransMat2Ref16, double* allVCM2Ref36);
/* calAbsErrorOfTransPara (R.cpp:1157-1251; decl R.h:198-199) */
PWICP_API int pwicp_abs_error_of_trans_para(const char* transMatFile, const char* GTtransMatFile, int allEpochNum, int startEpoch,
                                            const char* transParaErrorFile);
/* matrix2angle (C.cpp:385-407; decl C.h:172) and calBoundingBoxCornerChange (C.cpp:410-419; decl C.h:183): host arithmetic */
PWICP_API void  pwicp_matrix2angle(const float* transMat16, float* rotAngle3);
PWICP_API float pwicp_bbox_corner_change(const double* boundingBox6, const float* transMat16);
/* The per-pair result file <prefix>TransMatrix.txt as PiecewiseICP_pair_call / Piecewise_ICP_4D write it (R.cpp:341-388,
 * 492-539): 4x4 (fixed, 12 digits), angles in gon and translation (10 digits), 6x6 VCM (12 digits), standard deviations in
 * mgon / mm (10 digits).  Layout known-answer test: tests/test_host_stages.py feeds the reference's own numbers through it. */
PWICP_API int pwicp_write_trans_matrix_file(const char* path, const float* transMat16, const double* VCM36);

/* ---- the 4D series as a handle: independent pairs on any GPU (R.cpp:89-187) -------------------------------
 * PiecewiseICP_4D_call is open + run_pair over all pairs + write_results + close on one GPU.  On a multi-GPU node
 * every rank opens the same configuration, runs the pairs p with p mod world == rank, the fixed-size records are
 * all-gathered (RCCL) and rank 0 writes the files (pwicp_amd/series.py is the worked example). */
typedef struct pwicp_series pwicp_series;
typedef struct pwicp_pair_record {   /* 384 bytes, the unit of the all-gather */
    int32_t pair;                    /* index in the series' pair order, -1 = empty slot */
    int32_t status;                  /* PWICP_OK or the error code of the failed step */
    int32_t n_outer, n_inner;
    float T[16];                     /* T_final, row-major (R.cpp:461) */
    double VCM[36];
    int64_t n_corr;
    float t_loop_ms, t_pair_ms;      /* registration loop / whole pair incl. file input and setup */
} pwicp_pair_record;
/* adaptive_targets (pairMode < 0 only): NULL = compute the pair map here (calAdaptivePairSequence, R.cpp:552-589,
 * overlap ratios on the GPU) and write RegPairFile.txt; otherwise the map computed elsewhere, entry k = target of
 * source k+1, both relative to startEpoch, n_adaptive = #scan files - startEpoch - 1. */
PWICP_API int pwicp_series_open(const char* confile, int startEpoch, int epochNum, int pairMode, float overlapThd,
                                int device, const int32_t* adaptive_targets, int n_adaptive, pwicp_series** out);
PWICP_API void pwicp_series_close(pwicp_series* s);
PWICP_API int pwicp_series_num_pairs(const pwicp_series* s);
PWICP_API int pwicp_series_num_scans(const pwicp_series* s);     /* Epoch_### files found in the input folder */
PWICP_API int pwicp_series_pair_epochs(const pwicp_series* s, int pair, int* target_index, int* source_index,
                                       long* source_stamp);
PWICP_API int pwicp_series_adaptive_targets(const pwicp_series* s, int32_t* targets, int n);
/* The adaptive pair map in pieces (a multi-process run shards the independent overlap ratios, R.cpp:593-614, and replays the
 * sequential target scan, R.cpp:552-589, from the gathered table): open the series with adaptive_targets = NULL and
 * n_adaptive = -1 (deferred), compute any (target, source) file-index pairs with pwicp_series_overlap_ratios, then hand the
 * table (#files x #files floats, [i * #files + j], NaN = unknown: computed on the spot) to pwicp_series_adaptive_from_ratios. */
PWICP_API int pwicp_series_overlap_ratios(pwicp_series* s, const int32_t* target_source_pairs, int n_pairs, float* ratios);
PWICP_API int pwicp_series_adaptive_from_ratios(pwicp_series* s, const float* ratio_table, float overlapThd, int write_pair_file);
PWICP_API int pwicp_series_run_pair(pwicp_series* s, int pair, pwicp_pair_record* rec);
/* Any subset of the pairs, pipelined: scans read and supervoxels computed on host threads for several pairs at once,
 * GPU stages one after the other.  Same records as n calls of pwicp_series_run_pair. */
PWICP_API int pwicp_series_run_pairs(pwicp_series* s, const int32_t* pairs, int n_pairs, pwicp_pair_record* recs);
/* wall time per stage of the pairs run so far, ms5 = {read scans, GPU preparation, front ends (rest), registrations} in ms and
 * the raw scan bytes handed to the GPU */
PWICP_API int pwicp_series_stage_times(pwicp_series* s, double* ms5);
PWICP_API int pwicp_series_write_results(pwicp_series* s, const pwicp_pair_record* recs, int n_recs);
/* Several GPUs inside ONE process: n >= 1 HIP device ids (duplicates allowed: two workers sharing a GPU, for functional
 * tests on a 1-GPU box).  pwicp_series_run_pairs then deals the pairs of a call to the devices (pair k -> device k mod n),
 * one host thread and one context per device, the shared target of a Direct2Ref series prepared once per device.  Same
 * records as on one device.  PiecewiseICP_4D_call does this with every visible device unless $PWICP_DEVICES ("all" or a
 * comma-separated list) or $PWICP_DEVICE / $LOCAL_RANK (one rank of a multi-process run) say otherwise. */
PWICP_API int pwicp_series_set_devices(pwicp_series* s, const int32_t* devices, int n);
PWICP_API int pwicp_series_num_devices(const pwicp_series* s);

/* ---- a target that several PROCESSES share (Direct2Ref: every pair has the reference epoch as its target; the reference rebuilds
 * it for every pair, Registration.cpp:653).  One rank segments it; the others preprocess it themselves (they need its centroid for
 * their source, Registration.cpp:419-436) and take the supervoxel labels - 4 bytes per point - from that rank instead of running the
 * front end (Segmentation.cpp:18-68) once more per rank.  The labelling is a pure function of the preprocessed cloud: the records
 * are the ones every rank would have computed alone.  On the ranks that take: pwicp_series_expect_target_labels(scan) before
 * pwicp_series_run_pairs, then - from another thread, while the run preprocesses and segments this rank's sources -
 * pwicp_series_supply_target_labels with what the exchange delivered (m < 0, or a count that does not fit: the run segments the
 * target itself).  On the rank that gives: pwicp_series_wait_target_labels from another thread while its own run is under way
 * (labels == NULL: sizes only; PWICP_E_INTERNAL: the run failed on that target or ended without it), and
 * pwicp_series_close_target_labels once its run has returned.  pwicp_series_run_distributed and pwicp_amd.series do this with a
 * broadcast of their communicator.  `scan`: index into the folder's sorted scan files (startEpoch for a Direct2Ref series). */
PWICP_API int  pwicp_series_expect_target_labels(pwicp_series* s, int scan);
PWICP_API int  pwicp_series_supply_target_labels(pwicp_series* s, int scan, int m, int n_supervoxels, const int32_t* labels);
PWICP_API int  pwicp_series_wait_target_labels(pwicp_series* s, int scan, int timeout_ms, int* m, int* n_supervoxels, int32_t* labels, int cap);
PWICP_API void pwicp_series_close_target_labels(pwicp_series* s);
/* diagnostic: targets of this series whose labels came from another rank / were made by this one, so far */
PWICP_API int  pwicp_series_target_label_counts(pwicp_series* s, int* received, int* segmented);

/* ---- several PROCESSES, one GPU each: the series' one exchange over RCCL directly (no Python, no torch) -----------------------
 * librccl.so is opened with dlopen on first use.  Rendezvous of the ncclUniqueId through `id_file` (single node): rank 0
 * writes it, the others poll for it.  Buffers are host memory (staged through device memory: RCCL moves it over xGMI). */
typedef struct pwicp_comm pwicp_comm;
PWICP_API int  pwicp_comm_init(int rank, int world, int device, const char* id_file, pwicp_comm** comm);
PWICP_API void pwicp_comm_destroy(pwicp_comm* comm);
PWICP_API int  pwicp_comm_rank(const pwicp_comm* comm);
PWICP_API int  pwicp_comm_world(const pwicp_comm* comm);
PWICP_API int  pwicp_comm_allgather(pwicp_comm* comm, const void* send, size_t bytes, void* recv /* world * bytes */);
PWICP_API int  pwicp_comm_broadcast(pwicp_comm* comm, void* buf, size_t bytes, int root);
/* One rank of PiecewiseICP_4D_call sharded over `world` processes: pair p -> rank p mod world, adaptive pair map from rank 0
 * (broadcast), one all-gather of the 384-byte records, rank 0 writes the files.  Returns the same value on every rank.
 * PiecewiseICP_4D_call takes this path by itself when $PWICP_RCCL=1 and $WORLD_SIZE > 1 (rank / device from $RANK /
 * $LOCAL_RANK, id file $PWICP_RCCL_ID_FILE or /tmp/pwicp_rccl_<MASTER_PORT>.id). */
PWICP_API bool pwicp_series_run_distributed(const char* confile, int startEpoch, int epochNum, int pairMode, float overlapThd,
                                            int rank, int world, int device, const char* id_file);

/* How often, in this process, the serial host passes of the front end (host/frontend.cpp: the order-defining restatement of
 * supervoxel_segmentation.h:98-236) took over from the device passes - same labels either way, 10 - 20 x the time:
 * [0] clouds through the device front end, [1] fusions finished on the host because the device pass gave up (queue / arena
 * overflow, sweep cap), [2] boundary refinements finished on the host, [3] list arenas doubled and the fusion restarted on the
 * device, [4] fusions / [5] whole front ends run on the host because the environment asked for it ($PWICP_FUSION=host,
 * $PWICP_FRONTEND=host).  bench.py prints them beside the series' stage walls. */
PWICP_API int pwicp_frontend_fallback_counts(long long* counts6);

/* Host threads the library's one pool (host/parallel.h) works with in this process: the CPUs it may really use - affinity mask, cut by
 * the cgroup CPU quota, divided by $LOCAL_WORLD_SIZE (the ranks of a node share its CPUs) - at most 32, at least 1;
 * $PWICP_HOST_THREADS overrides.  Fixed at the first use. */
PWICP_API int pwicp_host_threads(void);

/* A series that is closed leaves its device contexts and front-end work spaces PARKED (one set per device) for the next series of
 * the process on that device: setting them up costs 0.15 - 0.25 s, a third of an 8 x 1 M-point series ($PWICP_SERIES_KEEP=0: off).
 * PiecewiseICP_pair_call parks its context the same way.  This frees both (also done at process exit). */
PWICP_API void pwicp_series_release_parked(void);

/* ---- measurement hooks ------------------------------------------------------------------------------------
 * HIP events on the pair's stream feed pwicp_result.t_dense_nn_ms / t_inner_ms.  An event record costs a ~5 us
 * bubble on the stream: none is recorded unless asked for (bench.py asks for PWICP_PROF_DENSE in every timed step). */
enum {
    PWICP_PROF_DENSE = 1,    /* events around every dense 1-NN launch   -> t_dense_nn_ms */
    PWICP_PROF_INNER = 2,    /* events around every inner-ICP batch     -> t_inner_ms */
    PWICP_PROF_REPLAY = 4    /* keep the stable flags of the first dense launch (the measurement replay of csrc/pwicp_internal.h) */
};
PWICP_API int pwicp_pair_set_profiling(pwicp_pair* pair, int flags);

#ifdef __cplusplus
}
#endif
#endif /* PWICP_H */
