/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU oracle for the Piecewise-ICP fine-registration loop: a plain-C restatement (single-threaded unless
 * orc_set_num_threads asks for the OpenMP variant of its batch nearest-neighbour queries; same results)
 * of the reference's algorithm (yihui4d/Piecewise-ICP @ 2025-09-05) for the
 * path src/Registration.cpp:618-972, 1255-1343, src/CommonFunc.cpp:145-179, 266-452,
 * src/Segmentation.cpp:97-150, 195-321 and of the PCL 1.8.1 / FLANN routines those lines
 * call (PCL is a third-party dependency that is NOT vendored in the reference tree and
 * NOT installed in this image; its published algorithms are restated from PCL 1.8.1:
 * registration/impl/{correspondence_estimation,icp,transformation_estimation_point_to_plane_lls,
 * default_convergence_criteria}.hpp, common/impl/{centroid,eigen,pca,transforms}.hpp,
 * features/normal_3d.h, filters/impl/{voxel_grid,statistical_outlier_removal}.hpp,
 * octree/impl/octree_pointcloud.hpp).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (libpwicp.so) never links, loads or calls it.
 *
 * Parity pinning: end-to-end against the reference's own checked-in results
 * (results/4DPCReg/<e>_Direct2Ref_TransMatrix.txt, see tests/golden/ and
 * tests/test_oracle_golden.py); the front end used for that pinning is the reference's
 * own codelibrary compiled into oracle/_ref/ (oracle/ref_frontend_driver.cpp).
 * Intermediate quantities (NN indices, 6x6 systems, normals) have no golden vectors in
 * the reference ("inner-loop parity unpinned by the reference"); they are
 * cross-checked against scipy/numpy in tests/test_oracle_pieces.py.
 *
 * All point arrays are pcl::PointXYZ-compatible: 4 floats (x, y, z, pad) per point.
 * PointNormal-like arrays are passed as separate xyz4 / normal4 arrays.
 * Matrices are row-major.
 */
#ifndef PWICP_ORACLE_H
#define PWICP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- exact 1-NN / k-NN (FLANN KDTreeSingleIndex semantics: float L2_Simple) -------- */
typedef struct orc_kdtree orc_kdtree;
orc_kdtree* orc_kdtree_build(const float* pts4, int n);
void        orc_kdtree_free(orc_kdtree* t);
/* nearest neighbour of each query; ties on the float d2 resolved to the lowest index */
void orc_kdtree_nn1(const orc_kdtree* t, const float* qry4, int nq, int* idx, float* d2);
/* k nearest, sorted ascending by (d2, idx) */
void orc_kdtree_knn(const orc_kdtree* t, const float* q3, int k, int* idx, float* d2);
/* pcl::registration::CorrespondenceEstimation::determineCorrespondences(.., DBL_MAX):
   builds the tree on the target (as the reference does at every call site) and searches. */
/* host threads of the batch nearest-neighbour searches (1 = faithful single-threaded cost, the default) */
void orc_set_num_threads(int n);
int orc_get_max_threads(void);
void orc_determine_correspondences(const float* tgt4, int nt, const float* src4, int ns,
                                   int* idx, float* d2);

/* ---- per-patch statistics ----------------------------------------------------------- */
/* pcl::computePointNormal + reference post-checks (C.cpp:284-333). returns 1 ok / 0 fail */
int   orc_cal_patch_normal(const float* pts4, int n, float* nx, float* ny, float* nz);
/* C.cpp:336-354 */
float orc_cal_patch_std(const float* pts4, int n);
/* S.cpp:195-228; keep[i]=1 if kept; returns number kept */
int   orc_patch_refinement(const float* pts4, int n, double sigma_mul, unsigned char* keep);
/* S.cpp:231-257 */
void  orc_cal_patch_feature(const float* pts4, int n, float* variation, float* planarity,
                            float* linearity);
/* S.cpp:260-303; ct4[4], bp4[6*4] */
void  orc_cal_patch_ct_bp(const float* pts4, int n, float* ct4, float* bp4);

/* S.cpp:97-150 + 306-321: group points by supervoxel label (point order), refine, select,
 * compute CT / BP / sigma.  Outputs are malloc'ed (free with orc_free):
 *   *pat4  refined patch points (CSR, concatenated), *off  [npatch+1],
 *   *src_index  original point index of each patch point,
 *   *ct4 [npatch*4], *bp4 [npatch*24], *bpstd, *ctstd [npatch].  Returns npatch. */
int orc_select_patches(const float* cloud4, int n, const int* labels, int nsv,
                       float** pat4, int** off, int** src_index,
                       float** ct4, float** bp4, float** bpstd, float** ctstd);
void orc_free(void* p);

/* ---- inner point-to-plane ICP (R.cpp:1255-1269 -> PCL IterativeClosestPointWithNormals) */
/* returns number of inner iterations; T16 row-major float 4x4 (final transformation);
 * n_corr_total (optional) accumulates the number of correspondences searched */
int orc_p2p_icp(const float* tgt4, const float* tgt_n4, int nt,
                const float* src4, const float* src_n4, int ns,
                double euclid_eps, float* T16, long long* n_corr_total);
/* one TransformationEstimationPointToPlaneLLS step on given correspondences:
 * ATA[36] row-major (mirrored), ATb[6], x[6], T16 */
void orc_p2p_lls(const float* src4, const float* tgt4, const float* tgt_n4,
                 const int* match, int ns, double* ATA, double* ATb, double* x, float* T16);

/* R.cpp:1273-1343 */
void orc_cal_trans_para_vcm(const float* tgt4, const float* tgt_n4, int nt,
                            const float* src_stable4, int ns, double* VCM36);

/* ---- helpers ------------------------------------------------------------------------- */
/* C.cpp:266-281 (+145-179) */
double orc_percentile_dist(const float* cloud1_4, int n1, const float* cloud2_4, int n2,
                           float percentile);
/* PCL OctreePointCloud::defineBoundingBox/getBoundingBox as used at R.cpp:881-886 */
void   orc_octree_bbox(const float* cloud4, int n, double resolution, double* bb6);
/* C.cpp:410-419 */
float  orc_bb_corner_change(const double* bb6, const float* T16);
/* C.cpp:385-407 */
void   orc_matrix2angle(const float* T16, float* ang3);
/* pcl::transformPointCloud, in place */
void   orc_transform_points(float* pts4, int n, const float* T16);
/* Eigen Matrix4f product C = A*B (row-major storage here) */
void   orc_mat4_mul(const float* A, const float* B, float* C);

/* ---- preprocessing (C.cpp:423-452, 239-263) ------------------------------------------ */
/* pcl::VoxelGrid; out4 must hold n points; returns number of output points */
int   orc_voxel_grid(const float* in4, int n, float leaf, float* out4);
/* pcl::StatisticalOutlierRemoval; returns number kept; out4 holds n points */
int   orc_sor_filter(const float* in4, int n, int mean_k, double std_mul, float* out4);
float orc_pc_resolution(const float* cloud4, int n);
/* R.cpp:593-614 */
float orc_overlap_ratio(const float* cloud1_4, int n1, const float* cloud2_4, int n2,
                        float DTinit);

/* ---- the loop (R.cpp:618-700 with patches already generated, and 704-972) ------------- */
#define ORC_MAX_OUTER 256
typedef struct {
    /* inputs */
    float Res1, Res2, SVRes1, SVRes2;
    int   isManualDTinit;
    float DTinit, DTmin;
    int   faithful_cost;   /* 1: rebuild KD-trees / recompute normals at the reference's
                              call sites (CPU-baseline timing); 0: hoist (same results) */
    /* outputs */
    int    status;             /* 0 ok; 1: <4 source patches; 2: <4 stable patches */
    int    n_outer;
    float  T16[16];            /* accumulated transMat (row-major) */
    double VCM[36];
    float  DTseries[ORC_MAX_OUTER + 1];
    int    n_inner[ORC_MAX_OUTER];
    int    n_stable[ORC_MAX_OUTER];
    int    n_stable_pts[ORC_MAX_OUTER];
    float  LoDmin[ORC_MAX_OUTER];
    float  maxBB[ORC_MAX_OUTER];
    double d75[ORC_MAX_OUTER];   /* -1 when the dense NN did not run */
    float  Tk[ORC_MAX_OUTER][16];
    long long n_corr;          /* correspondences searched inside the loop */
    double t_loop_s;           /* wall time of the while-loop */
    double t_inner_s;          /* wall time spent in the inner ICP calls */
    long long n_inner_total;
} orc_loop_io;

/* Runs R.cpp:626-631 (DTinit) and 660-694 on already generated patches.
 * cloud2 / ct2 / bp2 / pat2 are transformed in place, exactly as the reference does. */
int orc_piecewise_icp_loop(const float* cloud1_4, int n1, float* cloud2_4, int n2,
                           const float* pat1_4, const int* off1, int m1,
                           const float* ct1_4, const float* bp1_4,
                           float* pat2_4, const int* off2, int m2,
                           float* ct2_4, float* bp2_4,
                           orc_loop_io* io);

/* ---- diagnosis hooks (tools/rootcause_golden.py) --------------------------------------- */
typedef struct { int outer, patch, which; float dist, thr, rel; int stable; } orc_dbg_rec;
/* rel_margin >= 0 records every decision of R.cpp:828-853 with |dist-thr|/thr <= rel_margin (which: 0 CT plane, 1..6 BP
 * plane, 7 CT point distance); flips = (outer, patch) pairs whose verdict is inverted; variant bits: 1 eigen33 trig through
 * sinf/cosf/atan2f, 2 NN ties to the highest index, 4 LoD through a double sqrt, 8 voxel-grid points summed in input order
 * instead of the order MSVC's std::sort leaves them in, 16 / 64 the plane-fit / patch-feature scatter matrix accumulated in double
 * and rounded once instead of summed in float, 32 Eigen's float eigen-solver for the plane fit.  orc_debug_config(-1, NULL, 0, 0) = off. */
void orc_debug_config(double rel_margin, const int* flips, int nflip, unsigned variant);
void orc_debug_force_inner(int outer, int count);
/* patch-selection side: records (outer = kind, patch = supervoxel, which = point) for kind 0 refinement |d| < 2 sigma
 * (S.cpp:220-225), 1 variation gate, 2 planarity gate (S.cpp:127); flips = (kind, supervoxel, point) triples, kind 3 drops
 * the supervoxel.  Applies to the next orc_select_patches calls until switched off with (-1, NULL, 0). */
void orc_debug_select_config(double rel_margin, const int* flips, int nflip);
int  orc_debug_records(orc_dbg_rec* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
