// Per-patch statistics kernels: plane normals (float single-pass covariance + closed-form smallest
// eigenvector), PCA plane, 2-sigma refinement, eigen-feature gate, centroid, 6 boundary points, sigmas.
//
// Reference: calPatchNormal src/CommonFunc.cpp:284-333, generateCentroidCloudWithPatchNormals
// CommonFunc.cpp:357-382, calPatchSTD CommonFunc.cpp:336-354, PatchRefinement src/Segmentation.cpp:195-228,
// calPatchFeature Segmentation.cpp:231-257, calPatchCTandBP Segmentation.cpp:260-303, calBPandCTSTD
// Segmentation.cpp:306-321, patch extraction/selection Segmentation.cpp:97-150.
//
// Summation order inside a patch is the storage order (the reference's float sums are order dependent
// and feed threshold decisions), so one lane owns one patch and walks it sequentially.
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "devmath.h"
#include "nn_device.h"
#include "icp.h"
#include "select_dev.h"
#include "patch.h"
#include "xform_dev.h"

using namespace pwdev;

namespace {

constexpr int kBlock = 256;
constexpr int kFrontBlock = 256;     // a multiple of kGroup

// -DPWICP_KTRACE: start / end stamp of every block of the LAST k_xf_front launch (s_memrealtime, 10 ns; plain stores, no
// atomics: 20 k same-address atomics would take longer than the kernel), reduced per role by tools/ktrace_front.py
#ifdef PWICP_KTRACE
constexpr int kFtBlocks = 8192;
__device__ unsigned long long pw_fblk[3 * kFtBlocks];          // begin | end (thread 0's wave) | role
#define FT_ROLE_BEGIN(r_) do { if (threadIdx.x == 0 && blockIdx.x < kFtBlocks) { pw_fblk[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); pw_fblk[3 * blockIdx.x + 2] = (r_); } } while (0)
#define FT_ROLE_END(r_) do { if (threadIdx.x == 0 && blockIdx.x < kFtBlocks) pw_fblk[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int pwicp_debug_ftrace(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pw_fblk), sizeof(unsigned long long) * 3 * kFtBlocks) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[3 * kFtBlocks];
        if (hipMemcpyToSymbol(HIP_SYMBOL(pw_fblk), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#ifdef PWICP_QSTAT
extern "C" __attribute__((visibility("default"))) int pwicp_debug_qstat(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pw_qstat), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[32];
        if (hipMemcpyToSymbol(HIP_SYMBOL(pw_qstat), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#else
#define FT_ROLE_BEGIN(r_) do { } while (0)
#define FT_ROLE_END(r_) do { } while (0)
#endif

// pcl::computeMeanAndCovarianceMatrix (float, single pass) + solvePlaneParameters; false if n < 3
__device__ inline bool point_normal(const float4* __restrict__ p, int n, float* nrm) {
    if (n < 3) return false;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
    for (int i = 0; i < n; ++i) {
        float4 v = p[i];
        a0 += v.x * v.x; a1 += v.x * v.y; a2 += v.x * v.z;
        a3 += v.y * v.y; a4 += v.y * v.z; a5 += v.z * v.z;
        a6 += v.x; a7 += v.y; a8 += v.z;
    }
    float fn = (float)n;
    a0 /= fn; a1 /= fn; a2 /= fn; a3 /= fn; a4 /= fn; a5 /= fn; a6 /= fn; a7 /= fn; a8 /= fn;
    float cov[9];
    cov[0] = a0 - a6 * a6; cov[1] = a1 - a6 * a7; cov[2] = a2 - a6 * a8;
    cov[4] = a3 - a7 * a7; cov[5] = a4 - a7 * a8; cov[8] = a5 - a8 * a8;
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    eigen33_smallest(cov, nrm);
    return true;
}

// scatter of the kept points' demeaned float coordinates; returns the float mean.  The reference forms it as a float GEMM
// (pcl::PCA: alpha = D D^T, common/impl/pca.hpp; calPatchFeature / calPatchNormal: cloud_mat^T cloud_mat, S.cpp:246, C.cpp:311)
// whose inner dimension is the points: every element is a FLOAT sum over the points in order.  Accumulating in double and
// rounding once flips refinement decisions |d| < 2 sigma (S.cpp:220-225) at ~1e-6 relative margins: 3 of the reference's 57
// result files then sit at 3e-7 .. 9e-7 rad instead of <= 2e-8 (tools/rootcause_golden.py, oracle variant 16).
template <bool FILTER>
__device__ inline void mean_and_scatter(const float4* __restrict__ p, const unsigned char* __restrict__ keep,
                                        int n, int cnt, float* mean, double* S) {
    float c0 = 0, c1 = 0, c2 = 0;
    for (int i = 0; i < n; ++i) {
        if (FILTER && !keep[i]) continue;
        float4 v = p[i];
        c0 += v.x; c1 += v.y; c2 += v.z;
    }
    c0 /= (float)cnt; c1 /= (float)cnt; c2 /= (float)cnt;
    float s0 = 0, s1 = 0, s2 = 0, s4 = 0, s5 = 0, s8 = 0;
    for (int i = 0; i < n; ++i) {
        if (FILTER && !keep[i]) continue;
        float4 v = p[i];
        float dx = v.x - c0, dy = v.y - c1, dz = v.z - c2;
        s0 += dx * dx; s1 += dx * dy; s2 += dx * dz;
        s4 += dy * dy; s5 += dy * dz; s8 += dz * dz;
    }
    mean[0] = c0; mean[1] = c1; mean[2] = c2;
    S[0] = s0; S[1] = s1; S[2] = s2; S[3] = s1; S[4] = s4; S[5] = s5; S[6] = s2; S[7] = s5; S[8] = s8;
}

// pcl::PCA plane (a,b,c,d): normal = eigenvector of the smallest eigenvalue of the float scatter matrix
template <bool FILTER>
__device__ inline void pca_plane(const float4* __restrict__ p, const unsigned char* __restrict__ keep, int n,
                                 int cnt, float* abcd) {
    float c[3];
    double S[9], A[9], w[3], V[9];
    mean_and_scatter<FILTER>(p, keep, n, cnt, c, S);
    for (int i = 0; i < 9; ++i) A[i] = (double)(float)S[i];
    jacobi3(A, w, V);
    float a = (float)V[0], b = (float)V[3], cc = (float)V[6];
    abcd[0] = a; abcd[1] = b; abcd[2] = cc;
    abcd[3] = -((a * c[0] + b * c[1]) + cc * c[2]);
}

__device__ __forceinline__ double pt2plane(float4 v, const float* abcd) {
    float s = abcd[0] * v.x + abcd[1] * v.y + abcd[2] * v.z + abcd[3];
    return (double)fabsf(s);
}

// calPatchSTD over the kept points
template <bool FILTER>
__device__ inline float patch_std(const float4* __restrict__ p, const unsigned char* __restrict__ keep, int n,
                                  int cnt) {
    float abcd[4];
    pca_plane<FILTER>(p, keep, n, cnt, abcd);
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        if (FILTER && !keep[i]) continue;
        double d = pt2plane(p[i], abcd);
        s += d * d;
    }
    return (float)sqrt(s / (double)(cnt - 1));
}

// calPatchNormal incl. its fallback branch; returns its bool
__device__ inline bool cal_patch_normal(const float4* __restrict__ p, int n, float* out) {
    float nrm[3];
    if (n > 4 && point_normal(p, n, nrm)) {
        float nLen = sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        if (fabs((double)nLen - 1.0) < 1e-5) {
            out[0] = nrm[0]; out[1] = nrm[1]; out[2] = nrm[2];
            return true;
        }
        float mean[3];
        double S[9], C[9], w[3], V[9];
        mean_and_scatter<false>(p, nullptr, n, n, mean, S);
        for (int i = 0; i < 9; ++i) C[i] = (double)((float)S[i] / (float)n);
        jacobi3(C, w, V);
        out[0] = (float)V[0]; out[1] = (float)V[3]; out[2] = (float)V[6];
        float nLen2 = sqrtf(out[0] * out[0] + out[1] * out[1] + out[2] * out[2]);
        return fabs((double)nLen2 - 1.0) < 1e-5;
    }
    out[0] = 0; out[1] = 0; out[2] = 1;
    return false;
}

template <bool FILTER>
__device__ inline void ct_bp(const float4* __restrict__ p, const unsigned char* __restrict__ keep, int n, int cnt,
                             float4* ct, float4* bp) {
    float c0 = 0, c1 = 0, c2 = 0;
    const float inf = INFINITY;
    float4 b0 = make_float4(-inf, 0, 0, 1), b1 = make_float4(inf, 0, 0, 1), b2 = make_float4(0, -inf, 0, 1),
           b3 = make_float4(0, inf, 0, 1), b4 = make_float4(0, 0, -inf, 1), b5 = make_float4(0, 0, inf, 1);
    for (int i = 0; i < n; ++i) {
        if (FILTER && !keep[i]) continue;
        float4 v = p[i];
        v.w = 1.0f;
        c0 += v.x; c1 += v.y; c2 += v.z;
        if (v.x > b0.x) b0 = v;
        if (v.x < b1.x) b1 = v;
        if (v.y > b2.y) b2 = v;
        if (v.y < b3.y) b3 = v;
        if (v.z > b4.z) b4 = v;
        if (v.z < b5.z) b5 = v;
    }
    *ct = make_float4(c0 / (float)cnt, c1 / (float)cnt, c2 / (float)cnt, 1.0f);
    bp[0] = b0; bp[1] = b1; bp[2] = b2; bp[3] = b3; bp[4] = b4; bp[5] = b5;
}

// ------------------------------------------------------------------------------------------------------
// Plane normal of patch i, computed by a group of kGroup (8) consecutive lanes.  The group loads 64 consecutive points
// per pass (8 coalesced requests in flight per lane) into its LDS tile; then every lane owns ONE of the nine running
// sums of pcl::computeMeanAndCovarianceMatrix (lane 7 owns two) and adds the points' terms in storage order — the
// same float operations in the same order as the serial loop, but ~7 instead of ~21 instructions per point and lane,
// and 1-2 memory round trips per patch instead of ~10.
//   lane:      0    1    2    3    4    5    6    7
//   sum:       xx   xy   xz   yy   yz   zz   x    y (and z)
constexpr int kAhead = 4;
constexpr int kSumBatch = 8;                          // LDS reads in flight per lane while summing (8 costs the kernel two registers too many for 8 waves per SIMD)
constexpr int kTilePts = kGroup * kAhead;            // 32 points per pass (16 KiB of LDS per 256-thread block)
constexpr int kTileStride = kTilePts + 1;            // float4 units; +1: the 8 tiles of a wave start on different banks
// XF: the points are read from pat_in, moved by T on the way (pcl::transformPointCloud: xform_point, the same float
// operations as the stand-alone transform launch) and written to `pat`; the sums run over the moved points.
template <bool XF = false>
__device__ __forceinline__ void patch_normal_group(const float4* pat, const int* __restrict__ off, int i, int sub,
                                                   float4* __restrict__ nrm_out, float4* __restrict__ tile,
                                                   const float4* pat_in = nullptr, const float* T = nullptr, float4* pat_w = nullptr) {
    const int lo = off[i], hi = off[i + 1];
    const int gbase = (int)(__lane_id() & ~(unsigned)(kGroup - 1));       // first lane of this group within the wave
    // operand components of this lane's sum (w = 1 turns a plain sum into the same mul-then-add)
    const int c1 = sub < 3 ? 0 : (sub < 5 ? 1 : (sub == 5 ? 2 : (sub == 6 ? 0 : 1)));
    const int c2 = sub == 0 ? 0 : (sub == 1 || sub == 3) ? 1 : (sub == 2 || sub == 4 || sub == 5) ? 2 : 3;
    const float* tw = (const float*)tile;
    float acc = 0.f, acc8 = 0.f;
    float4 nxt[kAhead];                                // the next pass's points travel while this pass is summed
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
        const int j = lo + u * kGroup + sub;
        if (XF) {
            nxt[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < hi) { nxt[u] = xform_point(T, pat_in[j]); pat_w[j] = nxt[u]; }
        } else
        nxt[u] = (j < hi) ? pat[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int base = lo; base < hi; base += kTilePts) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            float4 v = nxt[u];
            v.w = 1.0f;
            tile[u * kGroup + sub] = v;
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int j = base + kTilePts + u * kGroup + sub;
            if (XF) {
                nxt[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < hi) { nxt[u] = xform_point(T, pat_in[j]); pat_w[j] = nxt[u]; }
            } else
            nxt[u] = (j < hi) ? pat[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");          // the tile is private to this group (one wave)
        __builtin_amdgcn_wave_barrier();
        const int cnt = min(kTilePts, hi - base);
        int t = 0;
        for (; t + kSumBatch <= cnt; t += kSumBatch) {
            float p[kSumBatch], q[kSumBatch], z[kSumBatch];
#pragma unroll
            for (int u = 0; u < kSumBatch; ++u) { p[u] = tw[(t + u) * 4 + c1]; q[u] = tw[(t + u) * 4 + c2]; z[u] = tw[(t + u) * 4 + 2]; }
#pragma unroll
            for (int u = 0; u < kSumBatch; ++u) { acc += p[u] * q[u]; acc8 += z[u]; }
        }
        for (; t < cnt; ++t) { acc += tw[t * 4 + c1] * tw[t * 4 + c2]; acc8 += tw[t * 4 + 2]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    float a0 = __shfl(acc, gbase + 0), a1 = __shfl(acc, gbase + 1), a2 = __shfl(acc, gbase + 2), a3 = __shfl(acc, gbase + 3),
          a4 = __shfl(acc, gbase + 4), a5 = __shfl(acc, gbase + 5), a6 = __shfl(acc, gbase + 6), a7 = __shfl(acc, gbase + 7),
          a8 = __shfl(acc8, gbase + 7);
    const int n = hi - lo;
    float nv[3] = {0.f, 0.f, 1.f};
    bool ok = false;
    if (n > 4 && n >= 3) {
        const float fn = (float)n;
        a0 /= fn; a1 /= fn; a2 /= fn; a3 /= fn; a4 /= fn; a5 /= fn; a6 /= fn; a7 /= fn; a8 /= fn;
        float cov[9];
        cov[0] = a0 - a6 * a6; cov[1] = a1 - a6 * a7; cov[2] = a2 - a6 * a8;
        cov[4] = a3 - a7 * a7; cov[5] = a4 - a7 * a8; cov[8] = a5 - a8 * a8;
        cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
        float e[3];
        eigen33_smallest(cov, e);
        const float nLen = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        if (fabs((double)nLen - 1.0) < 1e-5) {
            nv[0] = e[0]; nv[1] = e[1]; nv[2] = e[2];
            ok = true;
        } else {
            if (XF) {                                    // the group's own writes of the moved points, read back
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                __builtin_amdgcn_wave_barrier();
            }
            ok = cal_patch_normal(pat + lo, n, nv);      // takes the SVD-fallback branch (C.cpp:303-326)
        }
    }
    // w carries calPatchNormal's return value (1 / 0)
    if (sub == 0) nrm_out[i] = make_float4(nv[0], nv[1], nv[2], ok ? 1.0f : 0.0f);
}

__global__ void __launch_bounds__(kFrontBlock) k_patch_normals(const float4* __restrict__ pat, const int* __restrict__ off,
                                                               int m, float4* __restrict__ nrm_out) {
    __shared__ float4 tiles[(kFrontBlock / kGroup) * kTileStride];
    const int t = blockIdx.x * kFrontBlock + threadIdx.x;
    const int i = t / kGroup;
    if (i < m) patch_normal_group(pat, off, i, t % kGroup, nrm_out, tiles + (threadIdx.x / kGroup) * kTileStride);   // a whole group is in or out of range together
}

// What the launches of an outer iteration expect to find armed, done by the front launch that precedes them instead of by a
// launch of its own: words [0] / [1] of the iteration's scalar slot (atomicMin / atomicMax targets of the classification) and,
// at the start of a run, the diagnostic counters.  Idempotent: a front launch that is enqueued twice arms twice.
__device__ __forceinline__ void front_init(const FrontInit& in) {
    if (in.slot && blockIdx.x == 0 && threadIdx.x < 2) in.slot[threadIdx.x] = threadIdx.x == 0 ? 0xffffffffu : 0u;
    if (in.slot && blockIdx.x == 0 && threadIdx.x >= 2 && threadIdx.x < 4) in.slot[8 + threadIdx.x] = 0u;      // the stage guard's two words (stage_dev.h)
    if (in.zero && blockIdx.x < 16)
        for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < in.n_zero; i += 16 * (int)blockDim.x) in.zero[i] = 0ull;
}


// The "front" of an outer iteration in ONE launch: blocks [0, nb_nrm) compute the source patch normals (R.cpp:824),
// the remaining blocks the 1-NN of the source centroids and boundary points among the target centroids
// (R.cpp:737-747), 8 lanes per query.  The two are independent, each is a chain of dependent memory round trips that
// fills a fraction of the chip, and back to back they cost 16 + 17 us per iteration; side by side ~17 us.
// The normal blocks come first in the grid so that the longer chain starts first.
// SEL: the instantiation that can carry a selection pass on its leading blocks.  The pick of a pass is what needs the most
// registers in these kernels (87 VGPRs = 5 waves per SIMD against 64 = 8 without it), and the launches are bound by how many
// of their ~5000 short blocks are resident at a time - so the launches that carry no pass (all of them on the usual,
// speculative schedule) run the instantiation without the role.
template <bool SEL, int QG>
__global__ void __launch_bounds__(kFrontBlock) k_front(const float4* __restrict__ pat, const int* __restrict__ off, int m,
                                                       float4* __restrict__ nrm_out, int nb_nrm, GridDesc g,
                                                       const float4* __restrict__ q, int nq, int* __restrict__ idx,
                                                       float* __restrict__ d2, int nb_work, FusedSelect fs, FrontInit init) {
    __shared__ float4 tiles[(kFrontBlock / kGroup) * kTileStride];
    front_init(init);
    const int nsel = (SEL && fs.scratch) ? fs.nblk : 0;
    if (SEL && (int)blockIdx.x < nsel) {
        // leading blocks: pass 2 of the percentile selection of the PREVIOUS iteration's dense search (select_dev.h); the
        // bins live in the tile buffer
        static_assert(sizeof(tiles) >= kFsBins * sizeof(unsigned), "tile buffer too small for the selection bins");
        fs_pass_embedded<2>((unsigned*)tiles, fs, (int)blockIdx.x);
        return;
    }
    const int bid = (int)blockIdx.x - nsel;
    (void)nb_work;
    if (bid < nb_nrm) {
        const int t = bid * kFrontBlock + threadIdx.x;
        const int i = t / kGroup;
        if (i < m) patch_normal_group(pat, off, i, t % kGroup, nrm_out, tiles + (threadIdx.x / kGroup) * kTileStride);
        return;
    }
    const int t = (bid - nb_nrm) * kFrontBlock + threadIdx.x;
    const int i = t / QG, sub = t % QG;         // QG lanes per query (nn_device.h: nn_query_group)
    if (i >= nq) return;                        // a whole group is in or out of range together
    const float4 v = q[i];
    const NNBest b = nn_query_group<QG>(g, v.x, v.y, v.z, sub);
    if (sub == 0) {
        idx[i] = b.found() ? b.idx() : -1;
        d2[i] = b.d2();
    }
}

// pcl::transformPointCloud of a point array on `nb` blocks (4 coalesced requests in flight per lane)
__device__ __forceinline__ void xf_points_block(const Mat4& T, const float4* in, float4* out, int n, int bid, int nb) {
    const int stride = nb * kFrontBlock;
    for (int i = bid * kFrontBlock + (int)threadIdx.x; i < n; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n) out[i + u * stride] = xform_point(T.m, v[u]);
    }
}

// Transform update (R.cpp:943-954) AND the front of the next outer iteration (R.cpp:737-747, 824) in ONE launch.  Every role
// only needs T (read from the ICP state on the device, guarded like k_transform_all), so nothing waits for anything:
//   normal blocks   move the points of their patches (pat_in -> pat) on the way into the covariance sums;
//   query blocks    move centroids / boundary points (ctbp_in -> ctbp) and search the moved point among the target centroids;
//   cloud blocks    move the full cloud and fold its new bounding box (xf_cloud_block);
//   first blocks    passes 1 and 2 of the percentile selection when a dense search has just run (select_dev.h).
// The moved values are the same float expressions as in the stand-alone launches (xform_point), so normals, matches and
// distances are bit-identical to transform-then-front; the two launches cost 14 + 17 us back to back, this one ~24 us.
// (Alternating query and cloud blocks in the grid, or 6 / 8 waves per SIMD through launch bounds: no gain, measured.)
template <bool SEL, int QG>
__global__ void __launch_bounds__(kFrontBlock) k_xf_front(const float4* pat_in, float4* pat, const int* __restrict__ off, int m,
                                                          float4* __restrict__ nrm_out, int nb_nrm, GridDesc g,
                                                          const float4* ctbp_in, float4* ctbp, int nq, int* __restrict__ idx,
                                                          float* __restrict__ d2, int nb_nn,
                                                          const float4* cloud_in, float4* cloud, int n, int nb_cloud,
                                                          const IcpState* __restrict__ st, const unsigned* __restrict__ ns_dev,
                                                          unsigned* __restrict__ bbox_part, unsigned* __restrict__ slot, FusedSelect fs,
                                                          int nblk2, FrontInit init, const unsigned* __restrict__ guard, int npat) {
    __shared__ float4 tiles[(kFrontBlock / kGroup) * kTileStride];
    front_init(init);
    static_assert(kFrontBlock == kXfBlock, "xf_cloud_block is written for this block size");
    const int nsel = (SEL && fs.scratch) ? fs.nblk : 0;
    if (SEL && (int)blockIdx.x < nsel) {
        static_assert(sizeof(tiles) >= kFsBins * sizeof(unsigned), "tile buffer too small for the selection bins");
        fs_pass_embedded<1>((unsigned*)tiles, fs, (int)blockIdx.x, fs.mail.seq);
        return;
    }
    const int nsel2 = (SEL && fs.scratch) ? nblk2 : 0;
    if (SEL && (int)blockIdx.x < nsel + nsel2) {
        // pass 2 right behind pass 1 in the grid (its blocks wait for pass 1's tag: blocks with smaller indices, dispatched
        // before them): the percentile reaches the host while the rest of the launch is still running, so the next
        // classification - which needs it as its threshold - is enqueued without the device going idle
        FusedSelect f2 = fs;
        f2.nblk = nblk2;
        fs_pass_embedded<2>((unsigned*)tiles, f2, (int)blockIdx.x - nsel, fs.mail.seq);
        return;
    }
    // (flags, count, guard word and T are requested together: one round trip before the roles start instead of up to four)
    const int done_in = st->done;
    const unsigned ns_in = *ns_dev, guard_in = guard ? *guard : 1u;
    Mat4 T;
#pragma unroll
    for (int i = 0; i < 16; ++i) T.m[i] = st->Tfinal[i];
    if (!done_in || ns_in < 4u) return;
    if (guard_in != 1u) return;                 // enqueued on the guess that this iteration ends Stage 1: the ICP tail says otherwise
    int bid = (int)blockIdx.x - nsel - nsel2;
    if (bid < nb_nrm) {
        FT_ROLE_BEGIN(0);
        if (!nrm_out) {                 // no consumer for the source patch normals (loop.hip: source_normals()): the points only
            xf_points_block(T, pat_in, pat, npat, bid, nb_nrm);
            FT_ROLE_END(0);
            return;
        }
        const int t = bid * kFrontBlock + threadIdx.x;
        const int i = t / kGroup;
        if (i < m) patch_normal_group<true>(pat, off, i, t % kGroup, nrm_out, tiles + (threadIdx.x / kGroup) * kTileStride, pat_in, T.m, pat);
        FT_ROLE_END(0);
        return;
    }
    bid -= nb_nrm;
    if (bid < nb_nn) {
        const int t = bid * kFrontBlock + threadIdx.x;
        const int i = t / QG, sub = t % QG;
        if (i >= nq) return;                        // a whole group is in or out of range together
        FT_ROLE_BEGIN(1);
        const float4 v = xform_point(T.m, ctbp_in[i]);
        if (sub == 0) ctbp[i] = v;
        const NNBest b = nn_query_group<QG>(g, v.x, v.y, v.z, sub);
        if (sub == 0) {
            idx[i] = b.found() ? b.idx() : -1;
            d2[i] = b.d2();
        }
        FT_ROLE_END(1);
        return;
    }
    bid -= nb_nn;
    FT_ROLE_BEGIN(2);
    xf_cloud_block(T, cloud_in, cloud, n, bid, nb_cloud, bbox_part, slot, (float (*)[6])tiles);
    FT_ROLE_END(2);
}

// CT / BP / sigma of already selected patches
__global__ void k_patch_stats(const float4* __restrict__ pat, const int* __restrict__ off, int m,
                              float4* __restrict__ ct, float4* __restrict__ bp, float* __restrict__ bpstd,
                              float* __restrict__ ctstd) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float4* p = pat + off[i];
    int n = off[i + 1] - off[i];
    float4 c, b[6];
    ct_bp<false>(p, nullptr, n, n, &c, b);
    ct[i] = c;
    for (int k = 0; k < 6; ++k) bp[6 * i + k] = b[k];
    float sd = (n >= 2) ? patch_std<false>(p, nullptr, n, n) : 0.0f;
    bpstd[i] = sd;
    ctstd[i] = sd / (float)n;        // S.cpp:317-319
}

__global__ void k_gather_sorted(const float4* __restrict__ cloud, const int* __restrict__ order, int n,
                                float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cloud[order[i]];
}

__global__ void k_iota(int* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

__global__ void k_label_hist(const int* __restrict__ labels, int n, int nsv, int* __restrict__ cnt,
                             int* __restrict__ bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int l = labels[i];
    if (l < 0 || l >= nsv) { atomicAdd(bad, 1); return; }
    atomicAdd(&cnt[l], 1);
}

// One lane per supervoxel: S.cpp:107-150 (size gate, refinement, size gate, feature gate, CT/BP) + sigma.
__global__ void k_select(const float4* __restrict__ sp, const int* __restrict__ svoff, int nsv,
                         unsigned char* __restrict__ keep, int* __restrict__ kept_cnt,
                         float4* __restrict__ ct, float4* __restrict__ bp, float* __restrict__ bpstd,
                         float* __restrict__ ctstd) {
    const int minPtNum = 20;     // C.h:42
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsv) return;
    const float4* p = sp + svoff[s];
    unsigned char* kp = keep + svoff[s];
    int n = svoff[s + 1] - svoff[s];
    kept_cnt[s] = 0;
    if (n < minPtNum) return;                                   // S.cpp:109
    // PatchRefinement, S.cpp:195-228
    float abcd[4];
    pca_plane<false>(p, nullptr, n, n, abcd);
    double ss = 0.0;
    for (int i = 0; i < n; ++i) { double d = pt2plane(p[i], abcd); ss += d * d; }
    ss = sqrt(ss / (double)n);
    double thr = fabs(2.0 * ss);
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        unsigned char k = fabs(pt2plane(p[i], abcd)) < thr ? 1 : 0;
        kp[i] = k;
        cnt += k;
    }
    if (cnt < minPtNum) return;                                 // S.cpp:119
    // calPatchFeature, S.cpp:231-257
    float mean[3];
    double S[9], C[9], w[3], V[9];
    mean_and_scatter<true>(p, kp, n, cnt, mean, S);
    for (int i = 0; i < 9; ++i) C[i] = (double)((float)S[i] / (float)cnt);
    jacobi3(C, w, V);
    float e0 = (float)fabs(w[0]), e1 = (float)fabs(w[1]), e2 = (float)fabs(w[2]);
    // sort descending (2-pass bubble)
    float t;
    if (e0 < e1) { t = e0; e0 = e1; e1 = t; }
    if (e1 < e2) { t = e1; e1 = e2; e2 = t; }
    if (e0 < e1) { t = e0; e0 = e1; e1 = t; }
    float variation = e2 / (e0 + e1 + e2);
    float planarity = (e1 - e2) / e0;
    if (variation > 0.02f || planarity < 0.25f) return;         // S.cpp:127
    float4 c, b[6];
    ct_bp<true>(p, kp, n, cnt, &c, b);
    ct[s] = c;
    for (int k = 0; k < 6; ++k) bp[6 * s + k] = b[k];
    float sd = patch_std<true>(p, kp, n, cnt);
    bpstd[s] = sd;
    ctstd[s] = sd / (float)cnt;
    kept_cnt[s] = cnt;
}

// flags -> (valid ? 1 : 0) for the patch index scan
__global__ void k_valid_flags(const int* __restrict__ kept_cnt, int nsv, int* __restrict__ flag) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nsv) flag[s] = kept_cnt[s] > 0 ? 1 : 0;
    if (s == nsv) flag[s] = 0;
}

// one lane per supervoxel: copy its kept points to the final CSR position
__global__ void k_emit(const float4* __restrict__ sp, const int* __restrict__ order, const int* __restrict__ svoff,
                       int nsv, const unsigned char* __restrict__ keep, const int* __restrict__ kept_cnt,
                       const int* __restrict__ pidx, const int* __restrict__ poff,
                       const float4* __restrict__ ct, const float4* __restrict__ bp,
                       const float* __restrict__ bpstd, const float* __restrict__ ctstd,
                       float4* __restrict__ o_pat, int* __restrict__ o_off, int* __restrict__ o_src,
                       float4* __restrict__ o_ct, float4* __restrict__ o_bp, float* __restrict__ o_bpstd,
                       float* __restrict__ o_ctstd) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsv || kept_cnt[s] <= 0) return;
    int j = pidx[s], w = poff[s];
    o_off[j] = w;
    const int base = svoff[s], n = svoff[s + 1] - base;
    for (int i = 0; i < n; ++i)
        if (keep[base + i]) {
            float4 v = sp[base + i];
            v.w = 1.0f;
            o_pat[w] = v;
            o_src[w] = order[base + i];
            ++w;
        }
    o_ct[j] = ct[s];
    for (int k = 0; k < 6; ++k) o_bp[6 * j + k] = bp[6 * s + k];
    o_bpstd[j] = bpstd[s];
    o_ctstd[j] = ctstd[s];
}

__global__ void k_point_patch_ids(const int* __restrict__ off, int m, int* __restrict__ pid) {
    int j = blockIdx.x;
    for (int i = off[j] + threadIdx.x; i < off[j + 1]; i += blockDim.x) pid[i] = j;
}

}  // namespace

// ======================================================================================================
int pw_patch_normals_launch(pwicp_context* ctx, const float4* d_pat, const int* d_off, int m, float4* d_nrm) {
    if (m <= 0) return PWICP_OK;
    hipLaunchKernelGGL(k_patch_normals, dim3(div_up((long long)m * kGroup, kFrontBlock)), dim3(kFrontBlock), 0, ctx->stream,
                       d_pat, d_off, m, d_nrm);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

// Lanes per query of the front launches' searches (nn_device.h: nn_query_group<G>).  Measured on the 1 M-point pair: k_front
// (queries + normals: the chain of round trips decides) 17.0 us with 8 lanes, 18.0 with 4; k_xf_front (the cloud's transform
// shares the chip: resident blocks decide) 23.2 us with 8, 21.7 with 4, 25.6 with 2.  PWICP_FRONT_QUERY_LANES = 4 / 8 forces one for both.
// Few queries (the reference's own scans: 1.8 k patches, 14 k queries) leave the chip to the chains whatever else runs: 8 there
// too (their loop on Epoch_012: 0.80 -> 0.78 ms).
static int front_query_lanes(bool with_cloud, int nq) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PWICP_FRONT_QUERY_LANES"); v = e ? ((atoi(e) == 4 || atoi(e) == 8) ? atoi(e) : 0) : 0; }
    return v ? v : ((with_cloud && nq >= 50000) ? 4 : 8);
}

int pw_front_launch(pwicp_context* ctx, const float4* d_pat, const int* d_off, int m, float4* d_nrm, const GridDesc& g,
                    const float4* d_q, int nq, int* d_idx, float* d_d2, const FusedSelect* fs, const FrontInit* init) {
    if (m <= 0 || nq <= 0) {
        if (fs && fs->scratch) return pw_fs_pass_launch(ctx, 2, *fs);
        return PWICP_OK;
    }
    const int nb_nrm = d_nrm ? div_up((long long)m * kGroup, kFrontBlock) : 0;      // nullptr: the queries only
    const int qg = front_query_lanes(false, nq);
    const int nb_nn = div_up((long long)nq * qg, kFrontBlock);
    FusedSelect none{};
    const bool sel = fs && fs->scratch;
    const FrontInit fi = init ? *init : FrontInit{};
#define PW_FRONT(SEL_, QG_, NSEL_, FS_)                                                                                            \
    hipLaunchKernelGGL((k_front<SEL_, QG_>), dim3(nb_nrm + nb_nn + (NSEL_)), dim3(kFrontBlock), 0, ctx->stream, d_pat, d_off, m, d_nrm,   \
                       nb_nrm, g, d_q, nq, d_idx, d_d2, nb_nrm + nb_nn, FS_, fi)
    if (sel) { if (qg == 4) PW_FRONT(true, 4, fs->nblk, *fs); else PW_FRONT(true, 8, fs->nblk, *fs); }
    else { if (qg == 4) PW_FRONT(false, 4, 0, none); else PW_FRONT(false, 8, 0, none); }
#undef PW_FRONT
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_xf_front_launch(pwicp_context* ctx, const float4* d_pat_in, float4* d_pat, const int* d_off, int m, float4* d_nrm,
                       const GridDesc& g, const float4* d_ctbp_in, float4* d_ctbp, int nq, int* d_idx, float* d_d2,
                       const float4* d_cloud_in, float4* d_cloud, int n, const IcpState* d_state, const unsigned* d_ns,
                       unsigned* d_bbox_part, unsigned* d_slot, const FusedSelect* fs, const FrontInit* init, const unsigned* d_guard,
                       int npat) {
    // d_nrm == nullptr: the patch points are only moved (npat of them), on as many blocks as the cloud's share per point
    const int nb_nrm = d_nrm ? div_up((long long)m * kGroup, kFrontBlock) : std::max(1, std::min(div_up(npat, kFrontBlock), ctx->n_cu * 4));
    const int qg = front_query_lanes(true, nq);
    const int nb_nn = div_up((long long)nq * qg, kFrontBlock);
    const int nb_cloud = std::min(div_up(n, kFrontBlock), ctx->n_cu * 8);
    FusedSelect none{};
    const bool sel = fs && fs->scratch;
    const FrontInit fi = init ? *init : FrontInit{};
#define PW_XF_FRONT(SEL_, QG_, NSEL_, FS_)                                                                                          \
    hipLaunchKernelGGL((k_xf_front<SEL_, QG_>), dim3(nb_nrm + nb_nn + nb_cloud + (NSEL_)), dim3(kFrontBlock), 0, ctx->stream, d_pat_in,    \
                       d_pat, d_off, m, d_nrm, nb_nrm, g, d_ctbp_in, d_ctbp, nq, d_idx, d_d2, nb_nn, d_cloud_in, d_cloud, n, nb_cloud,    \
                       d_state, d_ns, d_bbox_part, d_slot, FS_, kFsBlocks, fi, d_guard, npat)
    if (sel) { if (qg == 4) PW_XF_FRONT(true, 4, fs->nblk + kFsBlocks, *fs); else PW_XF_FRONT(true, 8, fs->nblk + kFsBlocks, *fs); }
    else { if (qg == 4) PW_XF_FRONT(false, 4, 0, none); else PW_XF_FRONT(false, 8, 0, none); }
#undef PW_XF_FRONT
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_patch_stats_launch(pwicp_context* ctx, const float4* d_pat, const int* d_off, int m, float4* ct, float4* bp,
                          float* bpstd, float* ctstd) {
    if (m <= 0) return PWICP_OK;
    hipLaunchKernelGGL(k_patch_stats, dim3(div_up(m, 64)), dim3(64), 0, ctx->stream, d_pat, d_off, m, ct, bp,
                       bpstd, ctstd);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_select_patches_dev(pwicp_context* ctx, const float4* d_cloud, int n, const int* d_labels, int nsv,
                          PatchSet* out) {
    out->m = 0;
    out->tot = 0;
    if (n <= 0 || nsv <= 0) {
        HIPCHK(ctx, out->off.reserve(1));
        HIPCHK(ctx, hipMemsetAsync(out->off.p, 0, sizeof(int), ctx->stream));
        return PWICP_OK;
    }
    // supervoxel sizes and offsets
    DevBuf<int> svoff, bad, tmp;
    HIPCHK(ctx, svoff.reserve((size_t)nsv + 1));
    HIPCHK(ctx, bad.reserve(1));
    HIPCHK(ctx, hipMemsetAsync(svoff.p, 0, ((size_t)nsv + 1) * sizeof(int), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(bad.p, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_label_hist, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_labels, n, nsv,
                       svoff.p, bad.p);
    PWCHK(pw_exclusive_scan(ctx, svoff.p, (long long)nsv + 1, &tmp));
    int hbad = 0;
    HIPCHK(ctx, hipMemcpyAsync(&hbad, bad.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (hbad) { ctx->set_err("pwicp_select_patches: label out of range [0, n_supervoxels)"); return PWICP_E_INVALID; }

    // stable grouping by label (point order inside a supervoxel, S.cpp:99-103): LSD radix sort of (label, index)
    DevBuf<int> keys_out, order_in, order;
    HIPCHK(ctx, keys_out.reserve((size_t)n));
    HIPCHK(ctx, order_in.reserve((size_t)n));
    HIPCHK(ctx, order.reserve((size_t)n));
    hipLaunchKernelGGL(k_iota, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, order_in.p, n);
    int end_bit = 1;
    while (end_bit < 31 && (1ll << end_bit) < (long long)nsv) ++end_bit;
    size_t tbytes = 0;
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tbytes, d_labels, keys_out.p, order_in.p, order.p, n, 0,
                                                   end_bit, ctx->stream));
    DevBuf<unsigned char> tsort;
    HIPCHK(ctx, tsort.reserve(tbytes));
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(tsort.p, tbytes, d_labels, keys_out.p, order_in.p, order.p, n, 0,
                                                   end_bit, ctx->stream));
    DevBuf<float4> sp;
    HIPCHK(ctx, sp.reserve((size_t)n));
    hipLaunchKernelGGL(k_gather_sorted, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_cloud, order.p, n,
                       sp.p);

    DevBuf<unsigned char> keep;
    DevBuf<int> kept, pidx;
    DevBuf<float4> ct, bp;
    DevBuf<float> bpstd, ctstd;
    HIPCHK(ctx, keep.reserve((size_t)n));
    HIPCHK(ctx, kept.reserve((size_t)nsv + 1));
    HIPCHK(ctx, pidx.reserve((size_t)nsv + 1));
    HIPCHK(ctx, ct.reserve((size_t)nsv));
    HIPCHK(ctx, bp.reserve((size_t)nsv * 6));
    HIPCHK(ctx, bpstd.reserve((size_t)nsv));
    HIPCHK(ctx, ctstd.reserve((size_t)nsv));
    HIPCHK(ctx, hipMemsetAsync(kept.p, 0, ((size_t)nsv + 1) * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_select, dim3(div_up(nsv, 64)), dim3(64), 0, ctx->stream, sp.p, svoff.p, nsv, keep.p, kept.p,
                       ct.p, bp.p, bpstd.p, ctstd.p);
    hipLaunchKernelGGL(k_valid_flags, dim3(div_up(nsv + 1, kBlock)), dim3(kBlock), 0, ctx->stream, kept.p, nsv, pidx.p);
    // kept_cnt of rejected supervoxels is 0, so its exclusive scan is the output point offset
    DevBuf<int> poff;
    HIPCHK(ctx, poff.reserve((size_t)nsv + 1));
    HIPCHK(ctx, hipMemcpyAsync(poff.p, kept.p, ((size_t)nsv + 1) * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    PWCHK(pw_exclusive_scan(ctx, poff.p, (long long)nsv + 1, &tmp));
    PWCHK(pw_exclusive_scan(ctx, pidx.p, (long long)nsv + 1, &tmp));
    int hm = 0, htot = 0;
    HIPCHK(ctx, hipMemcpyAsync(&hm, pidx.p + nsv, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&htot, poff.p + nsv, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    out->m = hm;
    out->tot = htot;
    HIPCHK(ctx, out->pat.reserve((size_t)std::max(htot, 1)));
    HIPCHK(ctx, out->src.reserve((size_t)std::max(htot, 1)));
    HIPCHK(ctx, out->off.reserve((size_t)hm + 1));
    HIPCHK(ctx, out->ct.reserve((size_t)std::max(hm, 1)));
    HIPCHK(ctx, out->bp.reserve((size_t)std::max(hm, 1) * 6));
    HIPCHK(ctx, out->bpstd.reserve((size_t)std::max(hm, 1)));
    HIPCHK(ctx, out->ctstd.reserve((size_t)std::max(hm, 1)));
    hipLaunchKernelGGL(k_emit, dim3(div_up(nsv, 64)), dim3(64), 0, ctx->stream, sp.p, order.p, svoff.p, nsv, keep.p,
                       kept.p, pidx.p, poff.p, ct.p, bp.p, bpstd.p, ctstd.p, out->pat.p, out->off.p, out->src.p,
                       out->ct.p, out->bp.p, out->bpstd.p, out->ctstd.p);
    HIPCHK(ctx, hipMemcpyAsync(out->off.p + hm, &htot, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_point_patch_ids_launch(pwicp_context* ctx, const int* d_off, int m, int* d_pid) {
    if (m <= 0) return PWICP_OK;
    hipLaunchKernelGGL(k_point_patch_ids, dim3(m), dim3(64), 0, ctx->stream, d_off, m, d_pid);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}
