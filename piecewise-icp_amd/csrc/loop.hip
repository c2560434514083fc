// pwicp_pair_*: one target/source pair resident in HBM and the Piecewise-ICP outer loop.
//
// Reference: Piecewise_ICP src/Registration.cpp:618-700 (loop 680-694) and PwICP_singleIteration
// Registration.cpp:704-972.  Device residency: clouds, patches (CSR), centroids, boundary points, normals,
// sigmas and both search grids are uploaded/built once (pwicp_pair_create); per outer iteration only a few
// scalars (stable counts, LoD_min, the 4x4 of the inner ICP, the p75 distance) come back to the host, which
// runs the reference's distance-threshold schedule (Registration.cpp:891-935) verbatim.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>
#include <thread>

#include "common.h"
#include "devmath.h"
#include "icp.h"
#include "nn_device.h"
#include "patch.h"
#include "pwicp_internal.h"
#include "select_dev.h"
#include "stage_dev.h"
#include "xform_dev.h"

using namespace pwdev;


namespace {

constexpr int kBlock = 256;

inline float ord2f_host(unsigned u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// scal layout (unsigned words): [0] LoDmin bits, [1] LoDmax bits, [2] n stable, [3] n stable points,
//                               [4..6] bbox min (ordered), [7..9] bbox max (ordered)

// target centroids with normals for the ICP / VCM (C.cpp:357-382)
__global__ void k_with_norm(int m, const int* __restrict__ off, const float4* __restrict__ nrm,
                            float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    float4 n = nrm[i];
    if (!((off[i + 1] - off[i]) > 6 && n.w != 0.0f)) n = make_float4(0.f, 0.f, 1.f, 0.f);
    n.w = 0.f;
    out[i] = n;
}

// (8) R.cpp:943-954 in ONE launch: blocks [0, nb_cloud) transform cloud2 and reduce its new bounding box (for the
// next iteration's octree box, R.cpp:881-886); the remaining blocks transform centroids+boundary points and
// the patch points.  min/max are exact whatever the reduction order.
// The transformation is read from the ICP state on the device, so that the launch can be enqueued BEFORE the host
// has seen the ICP result; it does nothing unless that ICP call has converged on >= 4 stable patches (the host then
// takes the slow path and enqueues it again).
// `*_in` == the output arrays except in the first transform of a run on a reset pair, which reads the PRISTINE copies (the
// source state is restored by that transform instead of by a 48 MB device-to-device copy per pwicp_pair_reset).
__global__ void __launch_bounds__(kBlock) k_transform_all(const float4* cloud_in, const float4* ctbp_in, const float4* pat_in,
                                                          float4* cloud, int n, int nb_cloud,
                                                          float4* ctbp, int n_ctbp,
                                                          float4* pat, int n_pat,
                                                          const IcpState* __restrict__ st, const unsigned* __restrict__ ns_dev,
                                                          unsigned* __restrict__ bbox_part, unsigned* __restrict__ slot,
                                                          int nb_work, FusedSelect fs) {
    __shared__ float sh[kBlock / 64][6];
    const int nsel = fs.scratch ? fs.nblk : 0;
    if ((int)blockIdx.x < nsel) {
        // leading blocks: pass 1 of the percentile selection of this iteration's dense search (select_dev.h)
        __shared__ unsigned s_hist[kFsBins];
        fs_pass_embedded<1>(s_hist, fs, (int)blockIdx.x);
        return;
    }
    const int bid = (int)blockIdx.x - nsel;
    const int done_in = st->done;               // (flag, count and T requested together)
    const unsigned ns_in = *ns_dev;
    Mat4 T;
#pragma unroll
    for (int i = 0; i < 16; ++i) T.m[i] = st->Tfinal[i];
    if (!done_in || ns_in < 4u) return;
    if (bid >= nb_cloud) {
        xf_rest_block(T, ctbp_in, ctbp, n_ctbp, pat_in, pat, n_pat, bid - nb_cloud, nb_work - nb_cloud);
        return;
    }
    xf_cloud_block(T, cloud_in, cloud, n, bid, nb_cloud, bbox_part, slot, sh);
}

// arms the partial boxes and their counters (once, at pair creation; afterwards the folding block re-arms them)
__global__ void __launch_bounds__(64) k_bbox_arm(unsigned* __restrict__ bbox_part) {
    const int t = threadIdx.x;
#pragma unroll
    for (int d = 0; d < 3; ++d) { bbox_part[t * 32 + d] = 0xffffffffu; bbox_part[t * 32 + 3 + d] = 0u; }
    bbox_part[t * 32 + 8] = 0u;
    if (t == 0) bbox_part[kBoxParts * 32] = 0u;
}

// Device -> host mailbox: copies up to three word ranges into pinned, coherent host memory and then publishes a
// sequence number (common.h: mail_store / mail_drain / mail_publish).  The host spins on the sequence word instead of paying a
// hipMemcpy + hipStreamSynchronize round trip per outer iteration (stream order guarantees the producers ran).
__global__ void __launch_bounds__(64) k_mail(const unsigned* __restrict__ a, int na, const unsigned* __restrict__ b, int nb,
                                             const unsigned* __restrict__ c, int nc, unsigned* __restrict__ dst,
                                             unsigned* seq_ptr, unsigned seq) {
    const int t = threadIdx.x;
    for (int i = t; i < na; i += 64) mail_store(&dst[i], a[i]);
    for (int i = t; i < nb; i += 64) mail_store(&dst[na + i], b[i]);
    for (int i = t; i < nc; i += 64) mail_store(&dst[na + nb + i], c[i]);
    mail_drain();
    __syncthreads();
    if (t == 0) mail_publish(seq_ptr, seq);
}

// sums the 256 spread diagnostic counters into ctr[256*16] (one wave)
__global__ void __launch_bounds__(64) k_fold_examined(unsigned long long* __restrict__ ctr) {
    unsigned long long c = 0;
    for (int i = threadIdx.x; i < 256; i += 64) c += ctr[i * 16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (threadIdx.x == 0) ctr[256 * 16] = c;
}

// one 16-word scalar slot per outer iteration: [0] LoDmin [1] LoDmax [2] n stable [3] n stable points
// [4..6] bbox min, [7..9] bbox max of cloud2 AFTER this iteration's transform (ordered-uint encoded)
constexpr int kSlot = 16;
__global__ void k_scal_init(unsigned* __restrict__ scal, int n_slots, unsigned long long* __restrict__ zero, int n_zero) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_zero) zero[i] = 0ull;             // the run's diagnostic counters, re-armed in the same launch
    if (i >= n_slots * kSlot) return;
    const int w = i % kSlot;
    scal[i] = (w == 0 || (w >= 4 && w <= 6)) ? 0xffffffffu : 0u;
}

// pwicp_pair_reset: three device-to-device copies in one launch
__global__ void __launch_bounds__(kBlock) k_restore3(float4* __restrict__ d1, const float4* __restrict__ s1, long long n1,
                                                     float4* __restrict__ d2, const float4* __restrict__ s2, long long n2,
                                                     float4* __restrict__ d3, const float4* __restrict__ s3, long long n3) {
    const long long tot = n1 + n2 + n3, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        if (i < n1) d1[i] = s1[i];
        else if (i < n1 + n2) d2[i - n1] = s2[i - n1];
        else d3[i - n1 - n2] = s3[i - n1 - n2];
    }
}

// transform + bounding box of the result (for the next iteration's octree box, R.cpp:881-886).
// min/max are exact whatever the reduction order: wave shuffles -> LDS -> one atomic set per block.
__global__ void __launch_bounds__(kBlock) k_transform_bbox(float4* __restrict__ p, int n, Mat4 T, int apply,
                                                           unsigned* __restrict__ scal) {
    __shared__ float sh[kBlock / 64][6];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 v = p[i];
        if (apply) { v = xform_point(T.m, v); p[i] = v; }
        mn[0] = fminf(mn[0], v.x); mx[0] = fmaxf(mx[0], v.x);
        mn[1] = fminf(mn[1], v.y); mx[1] = fmaxf(mx[1], v.y);
        mn[2] = fminf(mn[2], v.z); mx[2] = fmaxf(mx[2], v.z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], o));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int d = 0; d < 3; ++d) { sh[wave][d] = mn[d]; sh[wave][3 + d] = mx[d]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float a = sh[0][threadIdx.x], b = sh[0][3 + threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) { a = fminf(a, sh[w][threadIdx.x]); b = fmaxf(b, sh[w][3 + threadIdx.x]); }
        atomicMin(&scal[4 + threadIdx.x], f2ord_dev(a));
        atomicMax(&scal[7 + threadIdx.x], f2ord_dev(b));
    }
}

// ---- host-side scalar pieces of the reference's control logic --------------------------------------------
// pcl::octree::OctreePointCloud::defineBoundingBox + getKeyBitSize, from the tight
// float min/max of the cloud.  resolution = double(Res2 * 2)  (R.cpp:882)
// (one source for the host loop and the ICP tail's stage guard: stage_dev.h)
void octree_bbox(const float* mn, const float* mx, double resolution, double* bb) { pw_octree_bbox(mn, mx, resolution, bb); }

// calBoundingBoxCornerChange, C.cpp:410-419
float bb_corner_change(const double* bb, const float* T) { return pw_bb_corner_change(bb, T); }

}  // namespace

// ==========================================================================================================
// The static target side of a pair: cloud, patches, normals and the two search grids.  Shareable between pairs with the
// same target scan (every pair of a Direct2Ref series, R.cpp:94-103): built once, read-only afterwards.
struct pwicp_target {
    pwicp_context* ctx = nullptr;
    float Res1 = 0.f, SVRes1 = 0.f;
    int n1 = 0;
    DevBuf<float4> cloud1;
    Grid g_c1, g_ct1;
    PatchSet P1;
    DevBuf<float4> nrm1;    // calPatchNormal per target patch, w = ok
    DevBuf<float4> ct1n;    // normals of CTcloud1_withNorm
};

struct pwicp_pair {
    pwicp_context* ctx = nullptr;
    pwicp_params prm{};
    pwicp_target* tgt = nullptr;     // target side
    bool owns_tgt = false;           // created by pwicp_pair_create (destroyed with the pair) or borrowed
    // source (transformed in place by the loop) + pristine copies for reset
    int n2 = 0;
    DevBuf<float4> cloud2, cloud2_0;
    PatchSet P2;
    DevBuf<float4> pat2_0;
    DevBuf<float4> ctbp2, ctbp2_0;   // live / pristine source centroids [0,m2) followed by boundary points [m2,7m2)
    float bmin0[3] = {0, 0, 0}, bmax0[3] = {0, 0, 0};   // tight bbox of the uploaded source cloud
    float step_bmin[3] = {0, 0, 0}, step_bmax[3] = {0, 0, 0};   // ... of the current source cloud (pwicp_pair_step)
    // Lazy reset: after pwicp_pair_reset the working arrays (cloud2, P2.pat, ctbp2) are stale and the pristine copies ARE the
    // source state; pwicp_pair_run reads them until its first transform has written the working arrays.  Everything else that
    // touches the working arrays calls materialize() first.
    bool dirty = false;      // the working arrays differ from the pristine copies
    bool lazy = false;       // ... and a reset is pending
    const float4* src_cloud() const { return lazy ? cloud2_0.p : cloud2.p; }
    const float4* src_pat() const { return lazy ? pat2_0.p : P2.pat.p; }
    const float4* src_ctbp() const { return lazy ? ctbp2_0.p : ctbp2.p; }
    DevBuf<float4> nrm2;
    DevBuf<int> pt_patch2;   // patch id of every source patch point
    DevBuf<int> qorder;      // source patch points in Morton order of their initial target-grid cell
    DevBuf<int> qpatch;      // pt_patch2[qorder[i]]
    DevBuf<float4> patq0;    // pat2_0[qorder[i]]: the queries of a dense search on the source as uploaded, in launch order
    const GridLevel* dense_lv = nullptr;   // small-cell level of the target this pair's dense search uses (pw_dense_level_for)
    double dense_far0 = 0.0;               // share of this pair's queries that start far from the target (same probe)
    DenseFarBuffers dense_far;             // hand-over of the far queries of a dense launch to the launch that puts 8 lanes on each
    DevBuf<int> all_stable;  // all-ones flags (bench replay over every patch)
    // per-iteration work
    DevBuf<int> mCTBP, stable;   // matches of the 7*m2 centroid+boundary queries
    DevBuf<float> dCTBP, d2dense;
    DevBuf<float4> stCT, stN;
    IcpWork icp;
    DevBuf<unsigned> scal, sel_scratch, bbox_part;
    DevBuf<unsigned> fs_scratch;     // fused percentile selection (select_dev.h), zeroed once
    bool no_fused_select = false;    // PWICP_FUSED_SELECT=0: three selection launches of their own (A/B measurements)
    DevBuf<float> sel_out;
    DevBuf<unsigned long long> examined;
    // stable flags of the first Stage-1 dense NN launch of the last run (replayed by bench_dense_nn)
    DevBuf<int> stable0;
    int ns0 = 0, nsp0 = 0;
    int profiling = 0;                    // pwicp_pair_set_profiling (no events unless asked for: a record is a ~5 us bubble on the stream)
    std::vector<hipEvent_t> ev;
    // mailbox in pinned coherent host memory: [0] sequence word, [16..] payload
    unsigned* mail_h = nullptr;
    unsigned* mail_d = nullptr;
    unsigned mail_seq = 0, sel_mail_seq = 0, vcm_mail_seq = 0;
    ~pwicp_pair() {
        for (auto e : ev) (void)hipEventDestroy(e);
        if (mail_h) (void)hipHostFree(mail_h);
        if (owns_tgt) delete tgt;
    }
    hipEvent_t event(size_t i) {
        while (ev.size() <= i) {
            hipEvent_t e;
            // (timing only: no system-scope fence when the event completes)
            if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return nullptr;
            ev.push_back(e);
        }
        return ev[i];
    }
};

namespace {

// The source patch normals of an outer iteration (generateCentroidCloudWithPatchNormals for CTcloud2_withNorm, R.cpp:823-824).
// In the reference they are DEAD values: the only reader of stableCT2's normal fields is pcl::IterativeClosestPointWithNormals,
// whose estimator (TransformationEstimationPointToPlaneLLS, PCL 1.8.1 transformation_estimation_point_to_plane_lls.hpp) forms
// its rows from the source POINT and the TARGET normal; the source normals are rotated along with the points
// (transformPointCloudWithNormals) and dropped with the aligned cloud.  Nothing of them reaches T, the VCM, a count, a
// threshold or a record.  They are computed all the same, in every iteration, as the reference does (row a5 of the path).
// PWICP_SOURCE_NORMALS=0 leaves them out (the front launches then only move the patch points; the working normals hold the
// (0,0,1) the reference uses for a failed fit): every result is bit-identical (tests/test_gpu_parity.py: scheduling switches)
// and a registration of the 1 M-point pair is 2 % shorter (k_front 16.8 -> 13.2 us, k_xf_front 22.0 -> 20.2 us) - the
// launches are bound by their slowest centroid queries and by the cloud's transform, not by the normals.
bool source_normals() {
    static const bool on = !(getenv("PWICP_SOURCE_NORMALS") && atoi(getenv("PWICP_SOURCE_NORMALS")) == 0);
    return on;
}


// static target side once its cloud and patches are on the device: patch normals, centroid normals, grids
int finish_target(pwicp_target* t) {
    pwicp_context* ctx = t->ctx;
    const int m1 = t->P1.m;
    HIPCHK(ctx, t->nrm1.reserve((size_t)std::max(m1, 1)));
    HIPCHK(ctx, t->ct1n.reserve((size_t)std::max(m1, 1)));
    PWCHK(pw_patch_normals_launch(ctx, t->P1.pat.p, t->P1.off.p, m1, t->nrm1.p));
    if (m1 > 0)
        hipLaunchKernelGGL(k_with_norm, dim3(div_up(m1, kBlock)), dim3(kBlock), 0, ctx->stream, m1, t->P1.off.p,
                           t->nrm1.p, t->ct1n.p);
    // cell edges: dense cloud grid = 3 x point spacing (27-cell stencil ~ 80 points; measured optimum: most first-
    // iteration queries, 1-2 spacings from the target, still resolve in the stencil); centroid grid = 1 x patch
    // size (its 2x coarse level resolves the far queries of displaced, unstable patches).
    // Tuning knobs for experiments only (results do not depend on them; the search is exact for any edge).
    float f_dense = 3.0f, f_ct = 1.0f;
    if (const char* e = getenv("PWICP_DENSE_CELL_FACTOR")) { float v = (float)atof(e); if (v > 0.f) f_dense = v; }
    if (const char* e = getenv("PWICP_CT_CELL_FACTOR")) { float v = (float)atof(e); if (v > 0.f) f_ct = v; }
    PWCHK(pw_grid_build(ctx, t->cloud1.p, t->n1, f_dense * t->Res1, &t->g_c1));
    // small cells for the disc-pruned dense search (PWICP_DISC_CELL_FACTOR x point spacing; 0 = off: 27-cell stencil kernel)
    float f_disc = 2.0f;
    if (const char* e = getenv("PWICP_DISC_CELL_FACTOR")) f_disc = (float)atof(e);
    if (f_disc > 0.f) PWCHK(pw_grid_add_dense(ctx, t->cloud1.p, t->n1, f_disc * t->Res1, &t->g_c1));
    PWCHK(pw_grid_build(ctx, t->P1.ct.p, m1, f_ct * t->SVRes1, &t->g_ct1));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PWICP_OK;
}

int finish_create(pwicp_pair* pr) {
    pwicp_context* ctx = pr->ctx;
    const int m2 = pr->P2.m;
    PWCHK(pw_check_finite(ctx, pr->cloud2.p, pr->n2));
    // dense-query order (one-off): Morton order of the source patch points in the target grid
    HIPCHK(ctx, pr->pt_patch2.reserve((size_t)std::max(pr->P2.tot, 1)));
    PWCHK(pw_point_patch_ids_launch(ctx, pr->P2.off.p, m2, pr->pt_patch2.p));
    PWCHK(pw_dense_level_for(ctx, pr->tgt->g_c1, pr->P2.pat.p, pr->P2.tot, &pr->dense_lv, &pr->dense_far0));
    PWCHK(pw_morton_order(ctx, pr->tgt->g_c1.d, pr->P2.pat.p, pr->P2.tot, &pr->qorder, pr->dense_lv));
    HIPCHK(ctx, pr->qpatch.reserve((size_t)std::max(pr->P2.tot, 1)));
    PWCHK(pw_gather_int_launch(ctx, pr->pt_patch2.p, pr->qorder.p, pr->P2.tot, pr->qpatch.p));
    // pristine source copies; centroids and boundary points live in ONE buffer so that a single NN launch and a
    // single transform launch serve both (R.cpp:737-747, 946-949)
    HIPCHK(ctx, pr->cloud2_0.reserve((size_t)std::max(pr->n2, 1)));
    HIPCHK(ctx, pr->pat2_0.reserve((size_t)std::max(pr->P2.tot, 1)));
    HIPCHK(ctx, pr->ctbp2.reserve((size_t)std::max(m2, 1) * 7));
    HIPCHK(ctx, pr->ctbp2_0.reserve((size_t)std::max(m2, 1) * 7));
    HIPCHK(ctx, hipMemcpyAsync(pr->cloud2_0.p, pr->cloud2.p, (size_t)pr->n2 * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(pr->pat2_0.p, pr->P2.pat.p, (size_t)pr->P2.tot * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, pr->patq0.reserve((size_t)std::max(pr->P2.tot, 1)));
    PWCHK(pw_gather_f4_launch(ctx, pr->P2.pat.p, pr->qorder.p, pr->P2.tot, pr->patq0.p));
    HIPCHK(ctx, hipMemcpyAsync(pr->ctbp2_0.p, pr->P2.ct.p, (size_t)m2 * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(pr->ctbp2_0.p + m2, pr->P2.bp.p, (size_t)m2 * 6 * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(pr->ctbp2.p, pr->ctbp2_0.p, (size_t)m2 * 7 * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    // work buffers
    const size_t M2 = (size_t)std::max(m2, 1);
    HIPCHK(ctx, pr->nrm2.reserve(M2));
    HIPCHK(ctx, pr->mCTBP.reserve(M2 * 7));
    HIPCHK(ctx, pr->dCTBP.reserve(M2 * 7));
    HIPCHK(ctx, pr->stable.reserve(M2));
    HIPCHK(ctx, pr->stable0.reserve(M2 + 1));
    HIPCHK(ctx, pr->stCT.reserve(M2));
    HIPCHK(ctx, pr->stN.reserve(M2));
    HIPCHK(ctx, pr->d2dense.reserve((size_t)std::max(std::max(pr->P2.tot, pr->n2), 1)));
    HIPCHK(ctx, pr->scal.reserve((size_t)kSlot * (PWICP_MAX_OUTER + 1)));
    HIPCHK(ctx, pr->bbox_part.reserve((size_t)kBoxParts * 32 + 32));
    hipLaunchKernelGGL(k_bbox_arm, dim3(1), dim3(64), 0, ctx->stream, pr->bbox_part.p);
    {   // tight bbox of the uploaded source cloud (R.cpp:881-886 needs it in the first iteration)
        hipLaunchKernelGGL(k_scal_init, dim3(1), dim3(64), 0, ctx->stream, pr->scal.p, 1, (unsigned long long*)nullptr, 0);
        Mat4 I{};
        hipLaunchKernelGGL(k_transform_bbox, dim3(std::min(div_up(pr->n2, kBlock), ctx->n_cu * 4)), dim3(kBlock), 0,
                           ctx->stream, pr->cloud2.p, pr->n2, I, 0, pr->scal.p);
        unsigned hb[kSlot];
        HIPCHK(ctx, hipMemcpyAsync(hb, pr->scal.p, sizeof(hb), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int d = 0; d < 3; ++d) { pr->bmin0[d] = ord2f_host(hb[4 + d]); pr->bmax0[d] = ord2f_host(hb[7 + d]); }
        for (int d = 0; d < 3; ++d) { pr->step_bmin[d] = pr->bmin0[d]; pr->step_bmax[d] = pr->bmax0[d]; }
    }
    HIPCHK(ctx, pr->sel_scratch.reserve(8 + 3 * 2048));
    HIPCHK(ctx, hipMemsetAsync(pr->sel_scratch.p, 0, (8 + 3 * 2048) * sizeof(unsigned), ctx->stream));   // armed: see pw_select_kth_launch
    HIPCHK(ctx, pr->sel_out.reserve(1));
    HIPCHK(ctx, pr->fs_scratch.reserve(kFsWords));
    HIPCHK(ctx, hipMemsetAsync(pr->fs_scratch.p, 0, kFsWords * sizeof(unsigned), ctx->stream));
    if (const char* e = getenv("PWICP_FUSED_SELECT")) pr->no_fused_select = atoi(e) == 0;
    HIPCHK(ctx, pr->examined.reserve(256 * 16 + 2));
    PWCHK(pr->icp.reserve(ctx, m2));
    HIPCHK(ctx, hipHostMalloc((void**)&pr->mail_h, 256 * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
    memset(pr->mail_h, 0, 256 * sizeof(unsigned));
    HIPCHK(ctx, hipHostGetDevicePointer((void**)&pr->mail_d, pr->mail_h, 0));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int upload4(pwicp_context* ctx, const float* h, int n, DevBuf<float4>* d) {
    HIPCHK(ctx, d->reserve((size_t)std::max(n, 1)));
    if (n > 0) HIPCHK(ctx, hipMemcpyAsync(d->p, h, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    return PWICP_OK;
}

int upload_patches(pwicp_context* ctx, const float* pat, const int32_t* off, int m, PatchSet* P) {
    P->m = m;
    P->tot = m > 0 ? off[m] : 0;
    PWCHK(upload4(ctx, pat, P->tot, &P->pat));
    HIPCHK(ctx, P->off.reserve((size_t)m + 1));
    HIPCHK(ctx, hipMemcpyAsync(P->off.p, off, ((size_t)m + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, P->ct.reserve((size_t)std::max(m, 1)));
    HIPCHK(ctx, P->bp.reserve((size_t)std::max(m, 1) * 6));
    HIPCHK(ctx, P->bpstd.reserve((size_t)std::max(m, 1)));
    HIPCHK(ctx, P->ctstd.reserve((size_t)std::max(m, 1)));
    PWCHK(pw_patch_stats_launch(ctx, P->pat.p, P->off.p, m, P->ct.p, P->bp.p, P->bpstd.p, P->ctstd.p));
    return PWICP_OK;
}

bool params_ok(const pwicp_params* p) {
    return p && p->Res1 > 0 && p->Res2 > 0 && p->SVRes1 > 0 && p->SVRes2 > 0 && p->DTmin > 0 &&
           (!p->isManualDTinit || p->DTinit > 0);
}

}  // namespace

extern "C" {

int pwicp_target_create(pwicp_context* ctx, const float* cloud1, int n1, const int32_t* labels1, int nsv1, float Res1,
                        float SVRes1, pwicp_target** out) {
    if (!ctx) return PWICP_E_INVALID;
    if (!out || !cloud1 || !labels1 || n1 <= 0 || nsv1 < 0 || !(Res1 > 0.f) || !(SVRes1 > 0.f)) {
        ctx->set_err("pwicp_target_create: invalid argument");
        return PWICP_E_INVALID;
    }
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pwicp_target* t = new (std::nothrow) pwicp_target();
    if (!t) return PWICP_E_NOMEM;
    t->ctx = ctx; t->n1 = n1; t->Res1 = Res1; t->SVRes1 = SVRes1;
    int rc = PWICP_OK;
    DevBuf<int> l1;
    do {
        if ((rc = upload4(ctx, cloud1, n1, &t->cloud1)) != PWICP_OK) break;
        if ((rc = pw_check_finite(ctx, t->cloud1.p, n1)) != PWICP_OK) break;       // before anything walks the points
        if (l1.reserve((size_t)n1) != hipSuccess) { rc = PWICP_E_NOMEM; break; }
        if (hipMemcpyAsync(l1.p, labels1, (size_t)n1 * sizeof(int), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = PWICP_E_NO_DEVICE; break; }
        if ((rc = pw_select_patches_dev(ctx, t->cloud1.p, n1, l1.p, nsv1, &t->P1)) != PWICP_OK) break;
        rc = finish_target(t);
    } while (0);
    if (rc != PWICP_OK) { delete t; return rc; }
    *out = t;
    return PWICP_OK;
}

void pwicp_target_destroy(pwicp_target* t) {
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    delete t;
}

int pwicp_pair_create_with_target(pwicp_target* t, const float* cloud2, int n2, const int32_t* labels2, int nsv2,
                                  const pwicp_params* params, pwicp_pair** out) {
    if (!t) return PWICP_E_INVALID;
    return pwicp_pair_create_with_target_on(t->ctx, t, cloud2, n2, labels2, nsv2, params, out);
}

int pwicp_pair_create_with_target_on(pwicp_context* ctx, pwicp_target* t, const float* cloud2, int n2, const int32_t* labels2, int nsv2,
                                     const pwicp_params* params, pwicp_pair** out) {
    if (!t || !ctx) return PWICP_E_INVALID;
    if (ctx->device != t->ctx->device) {
        ctx->set_err("pwicp_pair_create_with_target_on: the context is on another device than the target");
        return PWICP_E_INVALID;
    }
    if (!out || !cloud2 || !labels2 || n2 <= 0 || nsv2 < 0 || !params_ok(params) || params->Res1 != t->Res1 ||
        params->SVRes1 != t->SVRes1) {
        ctx->set_err("pwicp_pair_create_with_target: invalid argument (Res1 / SVRes1 must be the target's)");
        return PWICP_E_INVALID;
    }
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pwicp_pair* pr = new (std::nothrow) pwicp_pair();
    if (!pr) return PWICP_E_NOMEM;
    pr->ctx = ctx; pr->prm = *params; pr->tgt = t; pr->owns_tgt = false; pr->n2 = n2;
    int rc = PWICP_OK;
    DevBuf<int> l2;
    do {
        if ((rc = upload4(ctx, cloud2, n2, &pr->cloud2)) != PWICP_OK) break;
        if ((rc = pw_check_finite(ctx, pr->cloud2.p, n2)) != PWICP_OK) break;
        if (l2.reserve((size_t)n2) != hipSuccess) { rc = PWICP_E_NOMEM; break; }
        if (hipMemcpyAsync(l2.p, labels2, (size_t)n2 * sizeof(int), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = PWICP_E_NO_DEVICE; break; }
        if ((rc = pw_select_patches_dev(ctx, pr->cloud2.p, n2, l2.p, nsv2, &pr->P2)) != PWICP_OK) break;
        rc = finish_create(pr);
    } while (0);
    if (rc != PWICP_OK) { delete pr; return rc; }
    *out = pr;
    return PWICP_OK;
}

int pwicp_pair_create(pwicp_context* ctx, const float* cloud1, int n1, const int32_t* labels1, int nsv1,
                      const float* cloud2, int n2, const int32_t* labels2, int nsv2, const pwicp_params* params,
                      pwicp_pair** out) {
    if (!ctx) return PWICP_E_INVALID;
    if (!out || !cloud1 || !cloud2 || !labels1 || !labels2 || n1 <= 0 || n2 <= 0 || nsv1 < 0 || nsv2 < 0 ||
        !params_ok(params)) {
        ctx->set_err("pwicp_pair_create: invalid argument");
        return PWICP_E_INVALID;
    }
    *out = nullptr;
    pwicp_target* t = nullptr;
    PWCHK(pwicp_target_create(ctx, cloud1, n1, labels1, nsv1, params->Res1, params->SVRes1, &t));
    const int rc = pwicp_pair_create_with_target(t, cloud2, n2, labels2, nsv2, params, out);
    if (rc != PWICP_OK) { pwicp_target_destroy(t); return rc; }
    (*out)->owns_tgt = true;
    return PWICP_OK;
}

// optional per-patch arrays that replace the ones computed from the patch points (see pwicp_pair_create_from_arrays)
static int override_patch_arrays(pwicp_context* ctx, PatchSet* P, const float* ct, const float* bp, const float* std_bp, const float* std_ct) {
    const size_t m = (size_t)P->m;
    if (m == 0) return PWICP_OK;
    if (ct) HIPCHK(ctx, hipMemcpyAsync(P->ct.p, ct, m * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    if (bp) HIPCHK(ctx, hipMemcpyAsync(P->bp.p, bp, m * 6 * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    if (std_bp) HIPCHK(ctx, hipMemcpyAsync(P->bpstd.p, std_bp, m * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (std_ct) HIPCHK(ctx, hipMemcpyAsync(P->ctstd.p, std_ct, m * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return PWICP_OK;
}

int pwicp_pair_create_from_arrays(pwicp_context* ctx, const float* cloud1, int n1, const float* patch1, const int32_t* off1, int m1,
                                  const float* ct1, const float* bp1, const float* bpstd1, const float* ctstd1, const float* cloud2,
                                  int n2, const float* patch2, const int32_t* off2, int m2, const float* ct2, const float* bp2,
                                  const float* bpstd2, const float* ctstd2, const pwicp_params* params, pwicp_pair** out) {
    if (!ctx) return PWICP_E_INVALID;
    if (!out || !cloud1 || !cloud2 || !patch1 || !patch2 || !off1 || !off2 || n1 <= 0 || n2 <= 0 || m1 < 0 || m2 < 0 ||
        !params_ok(params)) {
        ctx->set_err("pwicp_pair_create_from_patches / _from_arrays: invalid argument");
        return PWICP_E_INVALID;
    }
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pwicp_pair* pr = new (std::nothrow) pwicp_pair();
    pwicp_target* t = new (std::nothrow) pwicp_target();
    if (!pr || !t) { delete pr; delete t; return PWICP_E_NOMEM; }
    t->ctx = ctx; t->n1 = n1; t->Res1 = params->Res1; t->SVRes1 = params->SVRes1;
    pr->ctx = ctx; pr->prm = *params; pr->tgt = t; pr->owns_tgt = true; pr->n2 = n2;
    int rc = PWICP_OK;
    do {
        if ((rc = upload4(ctx, cloud1, n1, &t->cloud1)) != PWICP_OK) break;
        if ((rc = upload4(ctx, cloud2, n2, &pr->cloud2)) != PWICP_OK) break;
        if ((rc = pw_check_finite(ctx, t->cloud1.p, n1)) != PWICP_OK) break;
        if ((rc = upload_patches(ctx, patch1, off1, m1, &t->P1)) != PWICP_OK) break;
        if ((rc = upload_patches(ctx, patch2, off2, m2, &pr->P2)) != PWICP_OK) break;
        if ((rc = override_patch_arrays(ctx, &t->P1, ct1, bp1, bpstd1, ctstd1)) != PWICP_OK) break;
        if ((rc = override_patch_arrays(ctx, &pr->P2, ct2, bp2, bpstd2, ctstd2)) != PWICP_OK) break;
        if ((rc = finish_target(t)) != PWICP_OK) break;
        rc = finish_create(pr);
    } while (0);
    if (rc != PWICP_OK) { delete pr; return rc; }
    *out = pr;
    return PWICP_OK;
}

int pwicp_pair_create_from_patches(pwicp_context* ctx, const float* cloud1, int n1, const float* patch1,
                                   const int32_t* off1, int m1, const float* cloud2, int n2, const float* patch2,
                                   const int32_t* off2, int m2, const pwicp_params* params, pwicp_pair** out) {
    return pwicp_pair_create_from_arrays(ctx, cloud1, n1, patch1, off1, m1, nullptr, nullptr, nullptr, nullptr, cloud2, n2, patch2, off2,
                                         m2, nullptr, nullptr, nullptr, nullptr, params, out);
}

void pwicp_pair_destroy(pwicp_pair* pr) {
    if (!pr) return;
    (void)hipSetDevice(pr->ctx->device);
    (void)hipStreamSynchronize(pr->ctx->stream);
    delete pr;
}

int pwicp_pair_num_patches(const pwicp_pair* pr, int* m1, int* m2) {
    if (!pr) return PWICP_E_INVALID;
    if (m1) *m1 = pr->tgt->P1.m;
    if (m2) *m2 = pr->P2.m;
    return PWICP_OK;
}

// the pristine copies back into the working arrays, in ONE launch (three copy commands cost two more launch gaps than they
// move bytes); only when something needs the working arrays of a reset pair before a run has rewritten them
static int materialize(pwicp_pair* pr) {
    if (!pr->lazy) return PWICP_OK;
    pwicp_context* ctx = pr->ctx;
    const long long n1 = pr->n2, n2 = pr->P2.tot, n3 = (long long)pr->P2.m * 7, tot = n1 + n2 + n3;
    if (tot > 0) {
        const int nb = (int)std::min<long long>((tot + kBlock - 1) / kBlock, (long long)ctx->n_cu * 16);
        hipLaunchKernelGGL(k_restore3, dim3(nb), dim3(kBlock), 0, ctx->stream, pr->cloud2.p, (const float4*)pr->cloud2_0.p, n1,
                           pr->P2.pat.p, (const float4*)pr->pat2_0.p, n2, pr->ctbp2.p, (const float4*)pr->ctbp2_0.p, n3);
    }
    pr->lazy = false;
    pr->dirty = false;
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pwicp_pair_reset(pwicp_pair* pr) {
    if (!pr) return PWICP_E_INVALID;
    pwicp_context* ctx = pr->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // nothing moves here: the next pwicp_pair_run reads the pristine copies until its first transform has rewritten the
    // working arrays (a reset used to be a 48 MB device-to-device restore per registration at 1 M points)
    if (pr->dirty) pr->lazy = true;
    for (int d = 0; d < 3; ++d) { pr->step_bmin[d] = pr->bmin0[d]; pr->step_bmax[d] = pr->bmax0[d]; }
    return PWICP_OK;
}

// Registrations of INDEPENDENT pairs side by side (the pair loop of R.cpp:89-187: no state flows between its iterations): one host
// thread per CONTEXT (= stream, pool, mailbox) among the pairs; the pairs of one context run one after the other on its thread, in
// the order given.  A registration is a chain of ~14 dependent short launches that leaves most of the chip idle; four chains in flight
// cost 0.13 ms each instead of 0.24 (bench.py pairs_side_by_side), the reference's own 19 pairs on four contexts 4.3 ms instead of
// 11.8 (tools/real_pairs_concurrent.py).  Each result is bit for bit what pwicp_pair_run gives for that pair alone
// (tests/test_gpu_parity.py).
int pwicp_pairs_run_concurrent(pwicp_pair* const* pairs, int n, pwicp_result* results, int reset_first) {
    if (!pairs || !results || n <= 0) return PWICP_E_INVALID;
    std::vector<pwicp_context*> ctxs;                 // distinct contexts, in order of first appearance
    std::vector<std::vector<int>> mine;               // pairs of each
    for (int a = 0; a < n; ++a) {
        if (!pairs[a]) return PWICP_E_INVALID;
        for (int b = 0; b < a; ++b)
            if (pairs[a] == pairs[b]) { pairs[a]->ctx->set_err("pwicp_pairs_run_concurrent: the same pair twice"); return PWICP_E_INVALID; }
        size_t c = 0;
        while (c < ctxs.size() && ctxs[c] != pairs[a]->ctx) ++c;
        if (c == ctxs.size()) { ctxs.push_back(pairs[a]->ctx); mine.emplace_back(); }
        mine[c].push_back(a);
    }
    std::vector<int> rc((size_t)n, PWICP_OK);
    auto one = [&](size_t c) {
        for (int k : mine[c]) {
            int r = reset_first ? pwicp_pair_reset(pairs[k]) : PWICP_OK;
            if (r == PWICP_OK) r = pwicp_pair_run(pairs[k], &results[k]);
            rc[(size_t)k] = r;
        }
    };
    std::vector<std::thread> th;
    th.reserve(ctxs.size());
    size_t started = 1;
    try {
        for (; started < ctxs.size(); ++started) th.emplace_back(one, started);
    } catch (...) {                                   // (no thread to be had: the remaining contexts on this one, afterwards)
    }
    one(0);
    for (size_t c = started; c < ctxs.size(); ++c) one(c);
    for (auto& t : th) t.join();
    for (int k = 0; k < n; ++k) if (rc[(size_t)k] != PWICP_OK) return rc[(size_t)k];
    return PWICP_OK;
}

int pwicp_pair_download_source(pwicp_pair* pr, float* cloud2_xyz4) {
    if (!pr || !cloud2_xyz4) return PWICP_E_INVALID;
    pwicp_context* ctx = pr->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PWCHK(materialize(pr));
    HIPCHK(ctx, hipMemcpyAsync(cloud2_xyz4, pr->cloud2.p, (size_t)pr->n2 * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PWICP_OK;
}

// everything PwICP_singleIteration mutates in place (R.cpp:943-954): cloud2, CTcloud2, BPcloud2, the source patch clouds.
// Any pointer may be NULL.  Sizes: n2 | m2 | 6 m2 | number of source patch points (pwicp_pair_num_patch_points).
int pwicp_pair_download_state(pwicp_pair* pr, float* cloud2_xyz4, float* centroid2_xyz4, float* boundary2_xyz4, float* patch2_xyz4) {
    if (!pr) return PWICP_E_INVALID;
    pwicp_context* ctx = pr->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int m2 = pr->P2.m;
    PWCHK(materialize(pr));
    if (cloud2_xyz4) HIPCHK(ctx, hipMemcpyAsync(cloud2_xyz4, pr->cloud2.p, (size_t)pr->n2 * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (centroid2_xyz4 && m2) HIPCHK(ctx, hipMemcpyAsync(centroid2_xyz4, pr->ctbp2.p, (size_t)m2 * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (boundary2_xyz4 && m2) HIPCHK(ctx, hipMemcpyAsync(boundary2_xyz4, pr->ctbp2.p + m2, (size_t)m2 * 6 * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (patch2_xyz4 && pr->P2.tot) HIPCHK(ctx, hipMemcpyAsync(patch2_xyz4, pr->P2.pat.p, (size_t)pr->P2.tot * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PWICP_OK;
}

int pwicp_pair_num_patch_points(const pwicp_pair* pr, int* tot1, int* tot2) {
    if (!pr) return PWICP_E_INVALID;
    if (tot1) *tot1 = pr->tgt->P1.tot;
    if (tot2) *tot2 = pr->P2.tot;
    return PWICP_OK;
}

constexpr int kSelMailSeq = 4, kSelMailPayload = 8;      // mailbox words of the percentile selection
constexpr int kVcmMailSeq = 6, kVcmMailPayload = 128;    // ... of the run's closing message (VCM | diagnostic counter)

// waits until the mailbox sequence word reaches `seq` (spin, then fall back to a stream synchronisation)
// PWICP_HOST_TRACE=1: host-side time stamps of pwicp_pair_run's enqueues and mailbox waits (stderr, us since the loop began)
struct HostTrace {
    bool on = getenv("PWICP_HOST_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::vector<std::pair<const char*, double>> ev;
    void operator()(const char* what) {
        if (on) ev.push_back({what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()});
    }
    ~HostTrace() {
        if (!on) return;
        double prev = 0;
        for (auto& e : ev) { fprintf(stderr, "[host] %8.1f us (+%6.1f)  %s\n", e.second, e.second - prev, e.first); prev = e.second; }
    }
};

static int mail_wait(pwicp_pair* pr, unsigned seq, int word = 0) {
    pwicp_context* ctx = pr->ctx;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(&pr->mail_h[word], __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0x3ff) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (__atomic_load_n(&pr->mail_h[word], __ATOMIC_ACQUIRE) != seq) {
                ctx->set_err("pwicp: device mailbox never signalled");
                return PWICP_E_INTERNAL;
            }
            break;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    return PWICP_OK;
}

// n_slots entries in d2dense of which n_valid are real distances (the rest carry the sentinel)
// 75th percentile of the dense distances (C.cpp:266-281): enqueue the selection and its mailbox message ...
static int select_p75_enqueue(pwicp_pair* pr, int n_slots, int n_valid, unsigned* seq_out) {
    pwicp_context* ctx = pr->ctx;
    int k = (int)((float)n_valid * 0.75f);      // C.cpp:177
    if (k >= n_valid) k = n_valid - 1;
    // the selection has mailbox words of its own (sequence word 4, payload word 8): its message may be overtaken by later
    // ICP messages when the search was enqueued speculatively
    const unsigned seq = ++pr->sel_mail_seq;
    SelectMail mail;
    mail.dst = pr->mail_d + kSelMailPayload; mail.seq_ptr = pr->mail_d + kSelMailSeq; mail.seq = seq;
    PWCHK(pw_select_kth_launch(ctx, pr->d2dense.p, n_slots, k, pr->sel_scratch.p, pr->sel_out.p, /*armed*/ true, &mail));
    *seq_out = seq;
    return PWICP_OK;
}

// ... and pick the value up (work enqueued in between hides the round trip)
static int select_p75_finish(pwicp_pair* pr, unsigned seq, double* out) {
    PWCHK(mail_wait(pr, seq, kSelMailSeq));
    float v;
    memcpy(&v, pr->mail_h + kSelMailPayload, 4);
    *out = (double)sqrtf(v);                    // C.cpp:277
    return PWICP_OK;
}

static int select_p75(pwicp_pair* pr, int n_slots, int n_valid, double* out) {
    unsigned seq = 0;
    PWCHK(select_p75_enqueue(pr, n_slots, n_valid, &seq));
    return select_p75_finish(pr, seq, out);
}

int pwicp_pair_run(pwicp_pair* pr, pwicp_result* res) {
    if (!pr || !res) return PWICP_E_INVALID;
    pwicp_context* ctx = pr->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    memset(res, 0, sizeof(*res));
    for (int i = 0; i < 16; ++i) res->T16[i] = (i % 5 == 0) ? 1.f : 0.f;
    const pwicp_params& prm = pr->prm;
    const int m2 = pr->P2.m, nbp2 = 6 * m2;
    size_t n_ev = 0;
    std::vector<std::pair<size_t, int>> ev_kind;   // (start event index, kind 0 dense / 1 inner)

    // R.cpp:626-631
    float DTinit = prm.DTinit;
    if (!prm.isManualDTinit) {
        PWCHK(pw_nn_launch(ctx, pr->tgt->g_c1.d, pr->src_cloud(), pr->n2, nullptr, pr->d2dense.p, nullptr));
        double d75 = 0;
        PWCHK(select_p75(pr, pr->n2, pr->n2, &d75));
        DTinit = (float)(d75 * 3.0);
    }
    float currDT = DTinit;
    const float DTmin = prm.DTmin;
    bool stage2 = false, stage3 = false;     // g_toStage2 / g_toStage3 (R.cpp:623-624), per call here
    float BB1 = 0.f, BB2 = 0.f;              // R.cpp:672-673
    res->DTseries[0] = currDT;

    // tight bbox of the current source cloud: uploaded state now, then refreshed by every transform launch
    float bmin[3], bmax[3];
    for (int d = 0; d < 3; ++d) { bmin[d] = pr->bmin0[d]; bmax[d] = pr->bmax0[d]; }
    // (the scalar slots and the run's diagnostic counters are armed by the front launches: patch.hip front_init)
    const int n_zero = 256 * 16 + 2;
    int status = PWICP_OK;
    int prev_inner = 2;
    bool vcm_pending = false;
    unsigned vcm_seq = 0;
    // Work that does not depend on the host's decisions is enqueued BEFORE the host waits for the mailbox, so that
    // the device never idles during a round trip: the transform update (8) reads T from the ICP state, and the
    // "front" of the next iteration (NN of centroids/boundary points + source patch normals) needs no threshold.
    bool front_ready = false;                 // front of iteration k already enqueued by iteration k-1
    double last_d75 = -1.0;                   // percentile of the run's previous dense search
    float prev_lod = NAN;
    // src_now: the front of the CURRENT iteration's state (pristine arrays on a lazily reset pair until the first transform);
    // false: the front of the NEXT iteration, enqueued behind a transform, reads the working arrays that transform writes
    auto enqueue_front = [&](unsigned* slot_k, const FusedSelect* fs = nullptr, bool src_now = false, bool run_start = false) -> int {
        // (1) R.cpp:737-747 — CT2 and BP2 queries against the static target-centroid grid — and the source patch
        // normals for CTcloud2_withNorm (R.cpp:824), recomputed from the transformed patch points: one launch
        // (+ pass 2 of the percentile selection on a few extra blocks when a dense search has just run)
        FrontInit in;
        in.slot = slot_k;
        if (run_start) { in.zero = pr->examined.p; in.n_zero = n_zero; }
        return pw_front_launch(ctx, src_now ? pr->src_pat() : pr->P2.pat.p, pr->P2.off.p, m2, source_normals() ? pr->nrm2.p : nullptr, pr->tgt->g_ct1.d,
                               src_now ? pr->src_ctbp() : pr->ctbp2.p, 7 * m2, pr->mCTBP.p, pr->dCTBP.p, fs, &in);
    };
    auto enqueue_transform = [&](unsigned* slot, const FusedSelect* fs = nullptr) {
        // (8) R.cpp:943-954: cloud2 (+ its new bbox into this slot), centroids + boundary points, patch points
        // (+ pass 1 of the percentile selection on a few extra blocks when a dense search has just run)
        const int nb_cloud = std::min(div_up(pr->n2, kBlock), ctx->n_cu * 8);
        const int nb_rest = std::min(div_up(7 * m2 + pr->P2.tot, kBlock), ctx->n_cu * 8);
        FusedSelect none{};
        hipLaunchKernelGGL(k_transform_all, dim3(nb_cloud + nb_rest + (fs ? fs->nblk : 0)), dim3(kBlock), 0, ctx->stream, pr->src_cloud(),
                           pr->src_ctbp(), pr->src_pat(), pr->cloud2.p, pr->n2, nb_cloud, pr->ctbp2.p, 7 * m2, pr->P2.pat.p, pr->P2.tot,
                           (const IcpState*)pr->icp.state.p, (const unsigned*)(slot + 2), pr->bbox_part.p, slot, nb_cloud + nb_rest,
                           fs ? *fs : none);
    };
    // (8) and the front of the NEXT iteration in one launch (patch.hip: k_xf_front): every role only needs T
    auto enqueue_xf_front = [&](unsigned* slot, const FusedSelect* fs = nullptr, const unsigned* guard = nullptr) -> int {
        FrontInit in;
        in.slot = slot + kSlot;                 // the front is the next iteration's
        return pw_xf_front_launch(ctx, pr->src_pat(), pr->P2.pat.p, pr->P2.off.p, m2, source_normals() ? pr->nrm2.p : nullptr, pr->tgt->g_ct1.d, pr->src_ctbp(),
                                  pr->ctbp2.p, 7 * m2, pr->mCTBP.p, pr->dCTBP.p, pr->src_cloud(), pr->cloud2.p, pr->n2,
                                  (const IcpState*)pr->icp.state.p, (const unsigned*)(slot + 2), pr->bbox_part.p, slot, fs, &in, guard, pr->P2.tot);
    };
    // the run's LAST update and the VCM (9) in one launch (icp.hip: k_xf_vcm).  guess: enqueued before the host has seen the
    // iteration's result; the VCM part then only runs if the iteration reaches Stage 3 (currDT == LoDet_min, R.cpp:896)
    auto enqueue_xf_vcm = [&](unsigned* slot, float currDT_now, bool guess, unsigned seq) -> int {
        VcmMail vm;
        vm.examined = pr->examined.p;
        vm.dst = pr->mail_d + kVcmMailPayload; vm.seq_ptr = pr->mail_d + kVcmMailSeq; vm.seq = seq;
        unsigned bits = 0;
        if (guess) memcpy(&bits, &currDT_now, 4);
        return pw_xf_vcm_launch(ctx, pr->tgt->g_ct1.d, pr->tgt->P1.ct.p, pr->tgt->ct1n.p, &pr->icp, pr->stCT.p, m2, &vm, bits,
                                pr->src_cloud(), pr->src_ctbp(), pr->src_pat(), pr->cloud2.p, pr->n2, pr->ctbp2.p, 7 * m2, pr->P2.pat.p,
                                pr->P2.tot, pr->bbox_part.p, slot);
    };
    // (7) of a Stage-1 iteration: dense NN of the stable patches' points against the full target cloud (C.cpp:266-281) with the
    // percentile selection riding on the launches that follow (select_dev.h): pass 0 in the dense kernel, pass 1 beside the
    // transform, pass 2 beside the next front (or on its own when no front follows).  `rank_dev`: the rank of the percentile is
    // derived on the device from the slot's stable-point count (speculative enqueue, before the host knows the count).
    constexpr long long kDenseSmallBlock = 256;       // (threads per block of the dense search, grid.hip: kDenseBlock)
    unsigned sel_seq = 0;
    // ahead_of_icp (out): only the search is enqueued and *ahead_of_icp describes the selection whose passes 1 / 2 the caller puts
    // on the ICP launches that follow (first iteration: the search does not depend on the ICP, and the percentile then arrives
    // with the ICP's result); the update + front go out behind the ICP as usual, without selection blocks
    auto enqueue_dense_tail = [&](unsigned* slot, int nsp, bool rank_dev, bool with_front, FusedSelect* ahead_of_icp = nullptr) -> int {
        // (a dispatch with events attached - hipExtLaunchKernelGGL - was measured too: the same two ~5 us bubbles as the records)
        const bool ev = (pr->profiling & PWICP_PROF_DENSE) != 0;
        if (ev) {
            ev_kind.push_back({n_ev, 0});
            HIPCHK(ctx, hipEventRecord(pr->event(n_ev), ctx->stream));
        }
        const bool fused = pr->dense_lv && !pr->no_fused_select;
        // the run's first search of a pair most of whose queries start far from the target (the probe at pair creation: 56 % on
        // the reference's scans, where the first transformation is centimetres; 16 % on the synthetic pair, whose far queries the
        // search handles faster inside its own blocks): their far queries go to a launch that puts eight lanes on each (grid.hip).
        // A later search of the run: when the percentile of the one before was still more than 1.5 cells of the small-cell level.
        static const int far_group_env = getenv("PWICP_DENSE_FAR_GROUP") ? atoi(getenv("PWICP_DENSE_FAR_GROUP")) : -1;
        // A SMALL search (round 5: at most four blocks per CU - the reference's own scans are < 2) always hands its far queries on: the
        // chip is nearly empty, and a block's far queries on one lane each of its first wave are half of its life (block trace:
        // Epoch_002's second search, far queries done +27.9 us of a 28.9 us block).  Epoch_016 / 019: loop 0.98 / 0.99 -> 0.78 / 0.83 ms,
        // the other scans +-1 %; the 1 M-point pair (3 700 blocks) keeps the rule above (47 instead of 32 us with the far launch).
        static const bool far_small_env = !(getenv("PWICP_DENSE_FAR_SMALL") && atoi(getenv("PWICP_DENSE_FAR_SMALL")) == 0);      // (0: round 4's rule)
        const bool small_search = far_small_env && pr->dense_lv && (long long)pr->P2.tot <= 4ll * kDenseSmallBlock * ctx->n_cu;
        const bool far_group_now = far_group_env >= 0 ? far_group_env != 0 : (small_search || (res->n_dense_nn_launches == 0 ? pr->dense_far0 > 0.3 : (pr->dense_lv && last_d75 > 1.5 * (double)pr->dense_lv->h)));
        FusedSelect fs{};
        if (fused) {
            int kk = (int)((float)nsp * 0.75f);         // C.cpp:177
            if (kk >= nsp) kk = nsp - 1;
            fs.scratch = pr->fs_scratch.p; fs.vals = pr->d2dense.p; fs.n = pr->P2.tot; fs.k = kk; fs.nblk = kFsBlocks;
            if (rank_dev) fs.n_valid_dev = slot + 3;
            fs.out = pr->sel_out.p;
            sel_seq = ++pr->sel_mail_seq;
            fs.mail.dst = pr->mail_d + kSelMailPayload; fs.mail.seq_ptr = pr->mail_d + kSelMailSeq; fs.mail.seq = sel_seq;
        }
        // (the source as uploaded - a reset pending, or nothing has moved it yet: the queries are at hand in launch order)
        static const bool patq_on = !(getenv("PWICP_DENSE_QUERY_COPY") && atoi(getenv("PWICP_DENSE_QUERY_COPY")) == 0);
        const bool unmoved = pr->lazy || !pr->dirty;
        PWCHK(pw_nn_dense_launch(ctx, pr->tgt->g_c1.d, pr->src_pat(), pr->qorder.p, pr->pt_patch2.p, pr->stable.p, pr->P2.tot,
                                 pr->d2dense.p, pr->examined.p, pr->dense_lv, pr->qpatch.p, fused ? &fs : nullptr,
                                 far_group_now ? &pr->dense_far : nullptr, (patq_on && unmoved) ? pr->patq0.p : nullptr));
        if (ev) {
            HIPCHK(ctx, hipEventRecord(pr->event(n_ev + 1), ctx->stream));
            n_ev += 2;
        }
        if (res->n_dense_nn_launches == 0 && (pr->profiling & PWICP_PROF_REPLAY))      // remember the first launch for stand-alone replays
            HIPCHK(ctx, hipMemcpyAsync(pr->stable0.p, pr->stable.p, (size_t)m2 * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        if (ahead_of_icp) { *ahead_of_icp = fs; return PWICP_OK; }       // (only taken with `fused`)
        if (!fused) PWCHK(select_p75_enqueue(pr, pr->P2.tot, nsp, &sel_seq));
        // the percentile only steers the threshold: transform and next front go out while it travels
        fs.nblk = 64;
        if (with_front) PWCHK(enqueue_xf_front(slot, fused ? &fs : nullptr));
        else enqueue_transform(slot, fused ? &fs : nullptr);
        fs.nblk = kFsBlocks;
        if (fused && !with_front) PWCHK(pw_fs_pass_launch(ctx, 2, fs));      // (with a front: pass 2 on the merged launch's last blocks)
        return PWICP_OK;
    };
    static const bool stage_guard = !(getenv("PWICP_STAGE_GUARD") && atoi(getenv("PWICP_STAGE_GUARD")) == 0);   // stage_dev.h
    static int speculate = -1;             // PWICP_SPECULATE_DENSE=0: never enqueue the first dense search ahead of the ICP result
    if (speculate < 0) { const char* e = getenv("PWICP_SPECULATE_DENSE"); speculate = e ? atoi(e) : 1; }
    const auto t0 = std::chrono::steady_clock::now();
    HostTrace ht;
    while (!stage3) {                                                   // R.cpp:680
        const int k = res->n_outer;
        if (k >= PWICP_MAX_OUTER) { status = PWICP_E_NOT_CONVERGED; break; }     // Stage 3 never reached: no VCM, not a success
        if (currDT <= DTmin) currDT = DTmin;                            // R.cpp:724-725
        if (4 > m2 || 1 > pr->tgt->P1.m) { status = PWICP_E_TOO_FEW_PATCHES; break; }   // R.cpp:728-731; an empty target has no match
        unsigned* const slot = pr->scal.p + (size_t)kSlot * k;

        ht("iteration begins");
        if (!front_ready) { PWCHK(enqueue_front(slot, nullptr, true, k == 0)); ht("front enqueued"); }
        front_ready = false;
        const float4* const ct2 = pr->src_ctbp();
        const float4* const bp2 = pr->src_ctbp() + m2;
        res->n_corr += (long long)m2 + nbp2;
        // (2)-(4)
        const float DTctct = currDT + 1 * (prm.SVRes1 + prm.SVRes2);   // R.cpp:817
        ClassifyArgs cls;
        cls.m2 = m2; cls.mCT = pr->mCTBP.p; cls.dCT = pr->dCTBP.p; cls.mBP = pr->mCTBP.p + m2; cls.dBP = pr->dCTBP.p + m2;
        cls.ctstd1 = pr->tgt->P1.ctstd.p; cls.bpstd2 = pr->P2.bpstd.p; cls.nrm1 = pr->tgt->nrm1.p; cls.ct1 = pr->tgt->P1.ct.p;
        cls.ct1n = pr->tgt->ct1n.p; cls.ct2 = ct2; cls.bp2 = bp2; cls.nrm2 = source_normals() ? pr->nrm2.p : nullptr; cls.off2 = pr->P2.off.p;
        cls.currDT = currDT; cls.DTmin = DTmin; cls.DTctct = DTctct;
        // (5) R.cpp:875-877: inner ICP enqueued right behind, its point count read from the slot on the device;
        // ONE host round trip returns the counts, LoD_min and the ICP state together.  Once Stage 2 is reached the
        // dense search (7) cannot run any more, so the transform and — unless this looks like the last iteration
        // (Stage 3 needs currDT == LoD_min, R.cpp:897) — the next front go out before the host waits.
        const bool early_xf = stage2;
        const bool early_front = stage2 && !(currDT == prev_lod);
        bool xf_enqueued = false;
        unsigned vcm_guess_seq = 0;        // != 0: the update went out merged with a guarded VCM (k_xf_vcm) under this mailbox tag
        // First iteration, still Stage 1: the dense search will almost certainly be needed (the clouds have just moved by the
        // whole initial misalignment), and it reads the PRE-transform positions, which do not change while the ICP iterates.
        // So it goes out right behind the first ICP batch, with the transform and the next front behind it, instead of after
        // the host has seen the ICP result (an exposed round trip of ~16 us).  Should the schedule switch to Stage 2 in
        // this very iteration, its result is simply not used; if the ICP needs more than the first batch, the transform /
        // front that were enqueued were no-ops / premature and are enqueued again.
        const bool spec_dense = speculate && k == 0 && !stage2 && pr->dense_lv && !pr->no_fused_select && !(pr->profiling & PWICP_PROF_REPLAY);
        bool spec_done = false, spec_xf_valid = false;
        // Still Stage 1 after the first iteration: whether THIS iteration ends it (R.cpp:891-894) depends on its transformation.
        // The ICP tail that converges takes that decision itself (stage_dev.h) and the update + next front go out behind the
        // batch, guarded by its flag - a no-op of ~4 us if Stage 1 goes on, instead of a host round trip (~12 us) if it ends.
        const bool guard_xf = stage_guard && k > 0 && !stage2 && !pr->lazy && !(pr->profiling & PWICP_PROF_REPLAY);
        StageGuard sg{};
        if (guard_xf) { sg.bbox6 = slot - kSlot + 4; sg.out = slot + 10; sg.resolution = (double)(prm.Res2 * 2); sg.DTmin = DTmin; }
        bool guard_ran = false;
        unsigned hs[kSlot], hb[6];
        IcpState hst;
        {
            // launches enqueued ahead: a launch after convergence is a ~4 us no-op, a missing one an exposed round trip
            // (~13 us).  On the reference's 57 registrations (tests/golden/oracle_vs_reference.json) the first outer iteration
            // needs 4 (27 x), 5 (16 x), 3-8 inner iterations, a later one after n: n = 1 -> 1 (11 of 11), 2 -> 2 (44) or 1 (27),
            // 3 -> 2 (34) or 3 (20), 4 -> 2 (17) or 3 (15), 5 -> 3 (10).  Hence 4, then 2 after 2 and half of the count (rounded
            // up) otherwise, then 2 at a time.
            int batch = (k == 0) ? 4 : (prev_inner == 2 ? 2 : std::max(1, (prev_inner + 1) / 2));
            bool first_batch = true;
            for (;;) {
                // (an event record costs a ~6 us bubble on the stream: the inner-loop timing is opt-in)
                const bool ev = (pr->profiling & PWICP_PROF_INNER) != 0;
                if (ev) {
                    ev_kind.push_back({n_ev, 1});
                    HIPCHK(ctx, hipEventRecord(pr->event(n_ev), ctx->stream));
                }
                // one mailbox message, sent by the batch's last launch: this slot | bbox words of the PREVIOUS slot
                // (cloud2 after the previous iteration's transform) | the ICP state
                const unsigned seq = ++pr->mail_seq;
                IcpMail mail;
                mail.a = slot; mail.na = kSlot;
                mail.b = k > 0 ? slot - kSlot + 4 : slot + 4; mail.nb = 6;
                mail.dst = pr->mail_d + 16; mail.seq_ptr = pr->mail_d; mail.seq = seq;
                // the first batch opens with the fused launch: classification, compaction and inner iteration 0 (whose
                // correspondences are the front's centroid matches); it carries the message when nothing follows it
                int n_iter = batch;
                FusedSelect fs_icp{};
                const bool spec_now = spec_dense && !spec_done;
                if (first_batch) {
                    n_iter = batch - 1;
                    PWCHK(pw_classify_icp0_launch(ctx, cls, pr->stable.p, pr->stCT.p, pr->stN.p, &pr->icp, slot, 1e-6, n_iter == 0 ? &mail : nullptr,
                                                  guard_xf ? &sg : nullptr));
                    first_batch = false;
                    // first iteration: the dense search right behind the classification (it needs the stable flags, not the ICP),
                    // passes 1 / 2 of its percentile on the ICP launches (n_iter = 3 here)
                    if (spec_now && n_iter >= 2) {
                        // (round 5, measured and removed in round 6: the search on a SECOND stream beside the ICP launches - bit-identical,
                        // 0.263 against 0.252 ms per step: the search keeps every CU's wave slots filled and a 16-wave ICP block finds
                        // room only once the search's grid has been dealt out)
                        PWCHK(enqueue_dense_tail(slot, 0, /*rank_dev*/ true, /*with_front*/ true, &fs_icp));
                    }
                }
                if (n_iter > 0)
                    PWCHK(pw_icp_enqueue(ctx, pr->tgt->g_ct1.d, pr->tgt->P1.ct.p, pr->tgt->ct1n.p, &pr->icp, m2, slot + 2, 1e-6, n_iter, &mail,
                                         fs_icp.scratch ? &fs_icp : nullptr, guard_xf ? &sg : nullptr));
                ht("classify+icp batch enqueued");
                if (ev) {
                    HIPCHK(ctx, hipEventRecord(pr->event(n_ev + 1), ctx->stream));
                    n_ev += 2;
                }
                if (guard_xf) PWCHK(enqueue_xf_front(slot, nullptr, slot + 11));
                if (early_xf) {                              // no-op on the device while the ICP has not converged
                    if (early_front) PWCHK(enqueue_xf_front(slot));
                    else {                                   // looks like the last iteration: the update together with the VCM
                        if (!vcm_guess_seq) vcm_guess_seq = ++pr->vcm_mail_seq;
                        PWCHK(enqueue_xf_vcm(slot, currDT, true, vcm_guess_seq));
                    }
                }
                if (spec_now) {
                    if (fs_icp.scratch) PWCHK(enqueue_xf_front(slot));          // the search and its selection are already on a stream
                    else PWCHK(enqueue_dense_tail(slot, 0, /*rank_dev*/ true, /*with_front*/ true));
                    spec_done = true;
                }
                ht("early transform / front / dense enqueued");
                PWCHK(mail_wait(pr, seq));
                ht("icp mail arrived");
                memcpy(hs, pr->mail_h + 16, sizeof(hs));
                memcpy(hb, pr->mail_h + 16 + kSlot, sizeof(hb));
                memcpy(&hst, pr->mail_h + 16 + kSlot + 6, sizeof(IcpState));
                // (fewer than 4 stable patches: R.cpp:864-867 stops below; the ICP state is then meaningless)
                if (hst.done || hst.iters >= 100 || (int)hs[2] < 4) {
                    xf_enqueued = early_xf; front_ready = early_xf && early_front;
                    // the guarded update behind THIS batch ran iff the batch held the converged state and the tail set the flag
                    guard_ran = guard_xf && hst.done && (int)hs[2] >= 4 && hs[11] == 1u;
                    spec_xf_valid = spec_now && hst.done && (int)hs[2] >= 4;     // the transform behind THIS batch saw the converged state
                    break;
                }
                batch = 2;
            }
        }
        if (k > 0)
            for (int d = 0; d < 3; ++d) { bmin[d] = ord2f_host(hb[d]); bmax[d] = ord2f_host(hb[3 + d]); }
        float LoDet_min;
        memcpy(&LoDet_min, &hs[0], 4);
        prev_lod = LoDet_min;
        const int ns = (int)hs[2], nsp = (int)hs[3];
        res->n_stable[k] = ns; res->n_stable_pts[k] = nsp; res->LoDmin[k] = LoDet_min;
        if (4 > ns) { status = PWICP_E_TOO_FEW_STABLE; break; }        // R.cpp:864-867
        float Tk[16];
        memcpy(Tk, hst.Tfinal, sizeof(Tk));
        const int n_in = hst.iters;
        prev_inner = n_in;
        res->n_inner[k] = n_in; res->n_inner_total += n_in;
        res->n_corr += (long long)ns * std::max(n_in, 1);
        memcpy(res->Tk[k], Tk, sizeof(Tk));

        // (6) R.cpp:881-888
        double bb[6];
        octree_bbox(bmin, bmax, (double)(prm.Res2 * 2), bb);
        float maxBB = bb_corner_change(bb, Tk);
        if (guard_xf && hst.done) {
            // the device has evaluated the same expression (stage_dev.h, one source) and acted on it: its value is the record
            float dev_maxBB;
            memcpy(&dev_maxBB, &hs[10], 4);
            if (memcmp(&dev_maxBB, &maxBB, 4) != 0 && getenv("PWICP_TRACE"))
                fprintf(stderr, "[pwicp] stage guard: device maxBBchange %.9g, host %.9g\n", (double)dev_maxBB, (double)maxBB);
            maxBB = dev_maxBB;
        }
        res->maxBB[k] = maxBB;

        // (7) R.cpp:891-935, verbatim control flow
        const float minLoD = DTmin;
        res->d75[k] = -1.0;
        if (!stage2 && maxBB < minLoD) stage2 = true;
        else if (currDT == LoDet_min) stage3 = true;
        if (spec_done && stage2) {
            // the schedule switched to Stage 2 in this iteration: the speculative search is not used (its mailbox message is
            // consumed so that the sequence numbers stay in step)
            double unused = 0;
            PWCHK(select_p75_finish(pr, sel_seq, &unused));
        }
        if (!stage2) {
            if (!spec_done) {
                PWCHK(enqueue_dense_tail(slot, nsp, false, !stage3));
                xf_enqueued = true;
                front_ready = !stage3;
            }
            if (pr->profiling & PWICP_PROF_REPLAY) { pr->ns0 = ns; pr->nsp0 = nsp; }
            double Dist75 = 0;
            PWCHK(select_p75_finish(pr, sel_seq, &Dist75));
            ht("percentile mail arrived");
            res->n_corr += nsp; res->n_corr_dense += nsp; res->n_dense_nn_launches++;
            res->d75[k] = Dist75;
            last_d75 = Dist75;
            if ((double)currDT > Dist75) currDT = (float)Dist75; else stage2 = true;
            if (currDT <= LoDet_min) currDT = LoDet_min;
            BB2 = BB1; BB1 = maxBB;
        }
        if (stage2 && !stage3) {
            const float upperBound = 0.8f, lowerBound = 0.5f;
            const float alpha = fabsf(BB1 / BB2);
            if (std::isnan(alpha) || std::isinf(alpha)) currDT = currDT * upperBound;
            else if (alpha < lowerBound) currDT = currDT * lowerBound;
            else if (alpha > upperBound) currDT = currDT * upperBound;
            else currDT = currDT * alpha;
            if (currDT <= LoDet_min) currDT = LoDet_min;
            BB2 = BB1; BB1 = maxBB;
        }

        if (spec_done) {
            // what went out behind the first ICP batch: valid if that batch already held the converged state
            xf_enqueued = spec_xf_valid;
            front_ready = spec_xf_valid && !stage3;
        }
        if (guard_ran) { xf_enqueued = true; front_ready = true; }      // (the flag is `maxBB < DTmin` itself: Stage 2 began above)
        // (8) the update, (9) R.cpp:958-961 the VCM of the last iteration on the stable centroids as copied BEFORE the update
        // (R.cpp:868); its launch also sends the run's closing message (VCM | diagnostic counter)
        bool vcm_done = false;
        if (xf_enqueued && vcm_guess_seq && stage3) {      // went out merged and guarded: the device has taken the same decision
            vcm_seq = vcm_guess_seq;
            vcm_done = true;
        }
        if (!xf_enqueued) {
            if (!stage3) { PWCHK(enqueue_xf_front(slot)); front_ready = true; }
            else {
                vcm_seq = ++pr->vcm_mail_seq;
                PWCHK(enqueue_xf_vcm(slot, currDT, false, vcm_seq));
                vcm_done = true;
            }
            ht("transform enqueued (late)");
        }
        pr->lazy = false;                   // a valid transform is on the stream: from here on the working arrays are the source state
        pr->dirty = true;
        if (stage3) {
            if (!vcm_done) {                                // the update went out with the next front (wrong guess): VCM on its own
                VcmMail vm;
                vm.examined = pr->examined.p;
                vm.dst = pr->mail_d + kVcmMailPayload; vm.seq_ptr = pr->mail_d + kVcmMailSeq; vm.seq = vcm_seq = ++pr->vcm_mail_seq;
                PWCHK(pw_vcm_enqueue(ctx, pr->tgt->g_ct1.d, pr->tgt->P1.ct.p, pr->tgt->ct1n.p, &pr->icp, pr->stCT.p, ns, &vm, /*have_match*/ true));
            }
            vcm_pending = true;
            res->n_corr += ns;
        }
        // R.cpp:687-689
        mat4_mul(Tk, res->T16, res->T16);
        res->n_outer = k + 1;
        res->DTseries[k + 1] = currDT;
    }
    // closing message: the VCM (R.cpp:958-961) and the diagnostic counter; its arrival also means the stream is idle
    unsigned long long ex = 0;
    if (vcm_pending) {
        PWCHK(mail_wait(pr, vcm_seq, kVcmMailSeq));
        ht("vcm mail arrived");
        memcpy(res->VCM, pr->mail_h + kVcmMailPayload, 36 * sizeof(double));
        memcpy(&ex, pr->mail_h + kVcmMailPayload + 72, sizeof(ex));
    } else {                                              // the loop ended without Stage 3 (error / iteration cap)
        hipLaunchKernelGGL(k_fold_examined, dim3(1), dim3(64), 0, ctx->stream, pr->examined.p);
        const unsigned seq = ++pr->mail_seq;
        hipLaunchKernelGGL(k_mail, dim3(1), dim3(64), 0, ctx->stream, (const unsigned*)(pr->examined.p + 256 * 16), 2,
                           (const unsigned*)nullptr, 0, (const unsigned*)nullptr, 0, pr->mail_d + 16, pr->mail_d, seq);
        PWCHK(mail_wait(pr, seq));
        memcpy(&ex, pr->mail_h + 16, sizeof(ex));
    }
    res->t_loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // The closing message has arrived: every result is on the host, and whatever the last launch still does (the cloud blocks
    // of k_xf_vcm run a few microseconds longer than its message) is ordered before anything a later call enqueues on this
    // stream.  So no hipStreamSynchronize here (~6 us per call) unless an event of this run is not complete yet - which the
    // recorded ones (the dense search's, early in the run) are.  PWICP_RUN_SYNC=1: always synchronise.
    static const bool always_sync = getenv("PWICP_RUN_SYNC") && atoi(getenv("PWICP_RUN_SYNC")) != 0;
    bool synced = false;
    if (always_sync || !vcm_pending || status != PWICP_OK) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); synced = true; }
    for (auto& e : ev_kind) {
        float ms = 0.f;
        hipError_t ee = hipEventElapsedTime(&ms, pr->ev[e.first], pr->ev[e.first + 1]);
        if (ee == hipErrorNotReady && !synced) {
            (void)hipGetLastError();
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            synced = true;
            ee = hipEventElapsedTime(&ms, pr->ev[e.first], pr->ev[e.first + 1]);
        }
        if (ee == hipSuccess) {
            if (e.second == 0) res->t_dense_nn_ms += ms; else res->t_inner_ms += ms;
        }
    }
    // bits 40..: dense queries that were cut short at the percentile's edge (k_nn_dense_far), bits 0..39: candidates examined
    res->n_dense_bounded = (int32_t)std::min<unsigned long long>(ex >> 40, 0x7fffffffull);
    ex &= (1ull << 40) - 1ull;
    res->dense_kbar = res->n_corr_dense > 0 ? (double)ex / (double)res->n_corr_dense : 0.0;
    // 0: disc-pruned search (rows vary with the candidate's distance); 3 / 9: the stencil kernel on columns / cells
    res->dense_rows = pr->dense_lv ? 0 : ((pr->tgt->g_c1.d.fine.ny == 1 || pr->tgt->g_c1.d.fine.nz == 1) ? 3 : 9);
    res->status = status;
    HIPCHK(ctx, hipGetLastError());
    return status;
}

// One outer iteration as a call of its own: PwICP_singleIteration (R.cpp:704-972; decl R.h:181-188).  The caller keeps
// what the reference keeps between calls — currDT, BBchange_1/2 and the two stage flags (module globals g_toStage2 /
// g_toStage3 there, R.cpp:11-14) — in a pwicp_step and drives the loop of Piecewise_ICP (R.cpp:680-694) itself.
// Same kernels as pwicp_pair_run, none of its cross-iteration pipelining: every hand-over is a plain synchronous copy.
// Stepping a freshly reset pair until toStage3 gives bit for bit the result of pwicp_pair_run.
int pwicp_pair_step(pwicp_pair* pr, pwicp_step* sp) {
    if (!pr || !sp) return PWICP_E_INVALID;
    pwicp_context* ctx = pr->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const pwicp_params& prm = pr->prm;
    const int m2 = pr->P2.m;
    float4* const ct2 = pr->ctbp2.p;
    float4* const bp2 = pr->ctbp2.p + m2;
    const float DTmin = prm.DTmin;
    sp->status = PWICP_OK;
    sp->d75 = -1.0;
    sp->n_inner = sp->n_stable = sp->n_stable_pts = 0;
    for (int i = 0; i < 16; ++i) sp->T16[i] = (i % 5 == 0) ? 1.f : 0.f;
    if (sp->currDT <= DTmin) sp->currDT = DTmin;                                    // R.cpp:724-725
    if (4 > m2 || 1 > pr->tgt->P1.m) return sp->status = PWICP_E_TOO_FEW_PATCHES;  // R.cpp:728-731
    PWCHK(materialize(pr));                  // (a lazily reset pair: the working arrays are restored now)
    const float currDT_in = sp->currDT;
    // one scalar slot, re-armed for this call
    unsigned* const slot = pr->scal.p;
    hipLaunchKernelGGL(k_scal_init, dim3(1), dim3(64), 0, ctx->stream, pr->scal.p, 1, (unsigned long long*)nullptr, 0);
    // (1) R.cpp:737-747 + source patch normals (R.cpp:824)
    PWCHK(pw_front_launch(ctx, pr->P2.pat.p, pr->P2.off.p, m2, source_normals() ? pr->nrm2.p : nullptr, pr->tgt->g_ct1.d, pr->ctbp2.p, 7 * m2, pr->mCTBP.p,
                          pr->dCTBP.p));
    // (2)-(4) R.cpp:750-871
    const float DTctct = currDT_in + 1 * (prm.SVRes1 + prm.SVRes2);
    ClassifyArgs cls;
    cls.m2 = m2; cls.mCT = pr->mCTBP.p; cls.dCT = pr->dCTBP.p; cls.mBP = pr->mCTBP.p + m2; cls.dBP = pr->dCTBP.p + m2;
    cls.ctstd1 = pr->tgt->P1.ctstd.p; cls.bpstd2 = pr->P2.bpstd.p; cls.nrm1 = pr->tgt->nrm1.p; cls.ct1 = pr->tgt->P1.ct.p;
    cls.ct1n = pr->tgt->ct1n.p; cls.ct2 = ct2; cls.bp2 = bp2; cls.nrm2 = source_normals() ? pr->nrm2.p : nullptr; cls.off2 = pr->P2.off.p;
    cls.currDT = currDT_in; cls.DTmin = DTmin; cls.DTctct = DTctct;
    // the same fused launch as pwicp_pair_run (classification, compaction, inner iteration 0): identical sums, identical T
    PWCHK(pw_classify_icp0_launch(ctx, cls, pr->stable.p, pr->stCT.p, pr->stN.p, &pr->icp, slot, 1e-6, nullptr));
    unsigned hs[kSlot];
    HIPCHK(ctx, hipMemcpyAsync(hs, slot, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    float LoDet_min;
    memcpy(&LoDet_min, &hs[0], 4);
    const int ns = (int)hs[2], nsp = (int)hs[3];
    sp->n_stable = ns; sp->n_stable_pts = nsp; sp->LoDmin = LoDet_min;
    if (4 > ns) return sp->status = PWICP_E_TOO_FEW_STABLE;                         // R.cpp:864-867
    // (5) R.cpp:875-877
    IcpState hst;
    for (;;) {
        PWCHK(pw_icp_enqueue(ctx, pr->tgt->g_ct1.d, pr->tgt->P1.ct.p, pr->tgt->ct1n.p, &pr->icp, ns, nullptr, 1e-6, 3, nullptr));
        HIPCHK(ctx, hipMemcpyAsync(&hst, pr->icp.state.p, sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (hst.done || hst.iters >= 100) break;
    }
    memcpy(sp->T16, hst.Tfinal, sizeof(sp->T16));
    sp->n_inner = hst.iters;
    // (6) R.cpp:881-888: octree box of the current source cloud
    double bb[6];
    octree_bbox(pr->step_bmin, pr->step_bmax, (double)(prm.Res2 * 2), bb);
    const float maxBB = bb_corner_change(bb, sp->T16);
    sp->maxBB = maxBB;
    // (7) R.cpp:891-935, verbatim control flow
    bool stage2 = sp->toStage2 != 0, stage3 = sp->toStage3 != 0;
    float currDT = currDT_in, BB1 = sp->BBchange_1, BB2 = sp->BBchange_2;
    if (!stage2 && maxBB < DTmin) stage2 = true;
    else if (currDT == LoDet_min) stage3 = true;
    if (!stage2) {
        PWCHK(pw_nn_dense_launch(ctx, pr->tgt->g_c1.d, pr->P2.pat.p, pr->qorder.p, pr->pt_patch2.p, pr->stable.p, pr->P2.tot,
                                 pr->d2dense.p, nullptr, pr->dense_lv, pr->qpatch.p));
        double Dist75 = 0;
        PWCHK(select_p75(pr, pr->P2.tot, nsp, &Dist75));
        sp->d75 = Dist75;
        if ((double)currDT > Dist75) currDT = (float)Dist75; else stage2 = true;
        if (currDT <= LoDet_min) currDT = LoDet_min;
        BB2 = BB1; BB1 = maxBB;
    }
    if (stage2 && !stage3) {
        const float upperBound = 0.8f, lowerBound = 0.5f;
        const float alpha = fabsf(BB1 / BB2);
        if (std::isnan(alpha) || std::isinf(alpha)) currDT = currDT * upperBound;
        else if (alpha < lowerBound) currDT = currDT * lowerBound;
        else if (alpha > upperBound) currDT = currDT * upperBound;
        else currDT = currDT * alpha;
        if (currDT <= LoDet_min) currDT = LoDet_min;
        BB2 = BB1; BB1 = maxBB;
    }
    // (8) R.cpp:943-954 and the tight box of the moved cloud for the next call
    {
        const int nb_cloud = std::min(div_up(pr->n2, kBlock), ctx->n_cu * 8);
        const int nb_rest = std::min(div_up(7 * m2 + pr->P2.tot, kBlock), ctx->n_cu * 8);
        FusedSelect none{};
        hipLaunchKernelGGL(k_transform_all, dim3(nb_cloud + nb_rest), dim3(kBlock), 0, ctx->stream, (const float4*)pr->cloud2.p,
                           (const float4*)pr->ctbp2.p, (const float4*)pr->P2.pat.p, pr->cloud2.p, pr->n2, nb_cloud, pr->ctbp2.p, 7 * m2,
                           pr->P2.pat.p, pr->P2.tot, (const IcpState*)pr->icp.state.p, (const unsigned*)(slot + 2), pr->bbox_part.p, slot,
                           nb_cloud + nb_rest, none);
        pr->dirty = true;
        HIPCHK(ctx, hipMemcpyAsync(hs, slot, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int d = 0; d < 3; ++d) { pr->step_bmin[d] = ord2f_host(hs[4 + d]); pr->step_bmax[d] = ord2f_host(hs[7 + d]); }
    }
    // (9) R.cpp:958-961
    if (stage3) PWCHK(pw_vcm_run(ctx, pr->tgt->g_ct1.d, pr->tgt->P1.ct.p, pr->tgt->ct1n.p, &pr->icp, pr->stCT.p, ns, sp->VCM));
    sp->currDT = currDT; sp->BBchange_1 = BB1; sp->BBchange_2 = BB2;
    sp->toStage2 = stage2 ? 1 : 0; sp->toStage3 = stage3 ? 1 : 0;
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

// DTinit of Piecewise_ICP when it is not given (R.cpp:626-631): 3 x the 75th percentile of the dense 1-NN distances
int pwicp_pair_auto_dtinit(pwicp_pair* pr, float* DTinit) {
    if (!pr || !DTinit) return PWICP_E_INVALID;
    pwicp_context* ctx = pr->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PWCHK(pw_nn_launch(ctx, pr->tgt->g_c1.d, pr->src_cloud(), pr->n2, nullptr, pr->d2dense.p, nullptr));
    double d75 = 0;
    PWCHK(select_p75(pr, pr->n2, pr->n2, &d75));
    *DTinit = (float)(d75 * 3.0);
    return PWICP_OK;
}

// calBoundingBoxCornerChange (C.cpp:410-419; decl C.h:183): largest displacement of the two extreme corners of the box
// (min x,y,z | max x,y,z) under transMat (row-major 4x4); pure host arithmetic, the function the loop itself uses
float pwicp_bbox_corner_change(const double* boundingBox6, const float* transMat16) {
    if (!boundingBox6 || !transMat16) return 0.f;
    return bb_corner_change(boundingBox6, transMat16);
}

int pwicp_pair_set_profiling(pwicp_pair* pr, int flags) {
    if (!pr) return PWICP_E_INVALID;
    pr->profiling = flags;
    return PWICP_OK;
}

int pwicp_pair_bench_dense_nn(pwicp_pair* pr, int n_launches, double* ms_per_launch, long long* n_queries,
                              double* kbar, double* cell_edge) {
    if (!pr || n_launches <= 0) return PWICP_E_INVALID;
    pwicp_context* ctx = pr->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // Replays the first Stage-1 dense launch of the last pwicp_pair_run (same stable flags) on the pristine
    // source patch points; over ALL source patches if no run happened yet.
    const int m2 = pr->P2.m, tot = pr->P2.tot;
    if (m2 <= 0 || tot <= 0) { ctx->set_err("bench_dense_nn: no source patches"); return PWICP_E_INVALID; }
    const int* flags = pr->stable0.p;
    int npts = pr->nsp0;
    if (pr->ns0 <= 0) {
        HIPCHK(ctx, pr->all_stable.reserve((size_t)m2));
        std::vector<int> ones((size_t)m2, 1);
        HIPCHK(ctx, hipMemcpy(pr->all_stable.p, ones.data(), (size_t)m2 * sizeof(int), hipMemcpyHostToDevice));
        flags = pr->all_stable.p;
        npts = tot;
    }
    HIPCHK(ctx, hipMemsetAsync(pr->examined.p, 0, 256 * 16 * sizeof(unsigned long long), ctx->stream));
    // warm-up launch (also measures Kbar)
    const GridLevel* dense = pr->dense_lv;
    PWCHK(pw_nn_dense_launch(ctx, pr->tgt->g_c1.d, pr->pat2_0.p, pr->qorder.p, pr->pt_patch2.p, flags, tot,
                                 pr->d2dense.p, pr->examined.p, dense, pr->qpatch.p));
    unsigned long long ex = 0;
    {
        std::vector<unsigned long long> hx(256 * 16);
        HIPCHK(ctx, hipMemcpyAsync(hx.data(), pr->examined.p, hx.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < 256; ++i) ex += hx[(size_t)i * 16] & ((1ull << 40) - 1ull);     // (bits 40..: bounded far queries)
    }
    hipEvent_t e0 = pr->event(0), e1 = pr->event(1);
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < n_launches; ++i)
        PWCHK(pw_nn_dense_launch(ctx, pr->tgt->g_c1.d, pr->pat2_0.p, pr->qorder.p, pr->pt_patch2.p, flags, tot,
                                     pr->d2dense.p, nullptr, dense, pr->qpatch.p));
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    if (ms_per_launch) *ms_per_launch = (double)ms / n_launches;
    if (n_queries) *n_queries = npts;
    if (kbar) *kbar = (double)ex / (double)npts;
    if (cell_edge) *cell_edge = dense ? dense->h : pr->tgt->g_c1.d.fine.h;
    return PWICP_OK;
}

int pwicp_pair_dense_distances(pwicp_pair* pr, int far_group, float* d2_out) {
    if (!pr || !d2_out) return PWICP_E_INVALID;
    pwicp_context* ctx = pr->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int m2 = pr->P2.m, tot = pr->P2.tot;
    if (m2 <= 0 || tot <= 0) { ctx->set_err("pwicp_pair_dense_distances: no source patches"); return PWICP_E_INVALID; }
    PWCHK(materialize(pr));
    HIPCHK(ctx, pr->all_stable.reserve((size_t)m2));
    {
        std::vector<int> ones((size_t)m2, 1);
        HIPCHK(ctx, hipMemcpyAsync(pr->all_stable.p, ones.data(), (size_t)m2 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    const bool far = far_group < 0 ? pr->dense_far0 > 0.3 : far_group != 0;
    PWCHK(pw_nn_dense_launch(ctx, pr->tgt->g_c1.d, pr->P2.pat.p, pr->qorder.p, pr->pt_patch2.p, pr->all_stable.p, tot, pr->d2dense.p, nullptr,
                             pr->dense_lv, pr->qpatch.p, nullptr, far ? &pr->dense_far : nullptr, nullptr));
    std::vector<float> d2((size_t)tot);
    std::vector<int> order((size_t)tot);
    HIPCHK(ctx, hipMemcpyAsync(d2.data(), pr->d2dense.p, (size_t)tot * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(order.data(), pr->qorder.p, (size_t)tot * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < tot; ++i) d2_out[order[(size_t)i]] = d2[(size_t)i];      // launch order -> order of the patch arrays
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

}  // extern "C"
