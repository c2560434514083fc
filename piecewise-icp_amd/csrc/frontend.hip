// Segmentation front end on the device: supervoxel labels of a cloud from its k-NN graph.
//
// Reference: PatchGenerationAndRefinement, src/Segmentation.cpp:18-68, driving codelibrary/geometry/point_cloud/
// supervoxel_segmentation.h:65-265 (fusion by doubling lambda :104-170, boundary refinement :172-236, relabel :238-248)
// with the metric of include/Segmentation.h:362-375.
//
// Both passes of the reference are SERIAL and ORDER DEPENDENT (one FIFO each).  They run here as speculative fixed-point
// iterations that converge to exactly the serial result:
//
//   * every unit of the serial order (a point popped from the refinement queue / a fusion centre) is evaluated in parallel
//     from the state the units BEFORE it left in the previous sweep ("state as of my turn": a value written by an earlier
//     unit is read from the previous sweep's outcome of that unit, everything else from the state at the start);
//   * sweeps repeat until no outcome changes.  The outcome of the first unit never depends on a guess, so by induction over
//     the serial order the fixed point IS the serial result; order-dependent side products (which point enters the next
//     queue generation first) are rebuilt afterwards from (position, neighbour) keys with an atomic min.
//
// All decisions use the reference's double arithmetic (no contraction: the library is built with -ffp-contract=off;
// sqrt and the division are IEEE on gfx950), so labels are identical to the host pipeline's, which the tests assert.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "../host/frontend.h"

namespace {

using pwhost::FePt;
constexpr int kNone = INT_MAX;
constexpr unsigned long long kNoKey = ~0ull;

// Segmentation.h:362-375
__device__ __forceinline__ double sv_metric(const FePt& p, const FePt& q, double resolution) {
    const double dot = p.nx * q.nx + p.ny * q.ny + p.nz * q.nz;
    const double t1 = p.x - q.x, t2 = p.y - q.y, t3 = p.z - q.z;
    const double dist = sqrt(t1 * t1 + t2 * t2 + t3 * t3);
    return 1.0 - fabs(dot) + dist / resolution * 0.4;
}

template <typename T>
__global__ void k_fill(T* p, long long n, T v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ================================================================================================================
// Boundary refinement (supervoxel_segmentation.h:172-236)
// ================================================================================================================
// The reference's FIFO is processed generation by generation: generation 0 = the seeds in scan order, generation g+1 =
// the points pushed while generation g was popped, in push order.  Inside a generation the point at position p sees, for
// a neighbour j, the label j had when p was popped: the new label of j when j sits at an earlier position of the same
// generation, else the label at the start of the generation.

// dis[i] = metric(i, label[i])  (:174-176)
__global__ void k_ref_dis(const FePt* __restrict__ P, const int* __restrict__ lab, int n, double res, double* __restrict__ dis) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dis[i] = sv_metric(P[i], P[lab[i]], res);
}

// seeds (:181-195): point i with a differently labelled neighbour pushes itself, then those neighbours in row order; a point
// is pushed once.  key = 65 * i + (0 for i itself | e + 1 for its neighbour e): the queue order is the order of the keys.
__global__ void k_ref_seed_keys(const int* __restrict__ nb, int k, int n, const int* __restrict__ lab, unsigned long long* __restrict__ key) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int* row = nb + (size_t)i * k;
    const int li = lab[i];
    bool any = false;
    for (int e = 0; e < k; ++e) {
        const int j = row[e];
        if (lab[j] != li) {
            any = true;
            atomicMin(&key[j], 65ull * (unsigned long long)i + (unsigned long long)(e + 1));
        }
    }
    if (any) atomicMin(&key[i], 65ull * (unsigned long long)i);
}

// number of queue entries point i contributes (its own + the neighbours whose first push it is); scatter = same walk
template <bool SCATTER>
__global__ void k_ref_seed_emit(const int* __restrict__ nb, int k, int n, const int* __restrict__ lab,
                                const unsigned long long* __restrict__ key, int* __restrict__ cnt, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int* row = nb + (size_t)i * k;
    const int li = lab[i];
    int c = 0;
    const int base = SCATTER ? cnt[i] : 0;
    if (key[i] == 65ull * (unsigned long long)i) {
        if (SCATTER) out[base] = i;
        ++c;
    }
    for (int e = 0; e < k; ++e) {
        const int j = row[e];
        if (lab[j] != li && key[j] == 65ull * (unsigned long long)i + (unsigned long long)(e + 1)) {
            if (SCATTER) out[base + c] = j;
            ++c;
        }
    }
    if (!SCATTER) cnt[i] = c;
}

__global__ void k_ref_begin_generation(const int* __restrict__ L, int m, const int* __restrict__ lab, int* __restrict__ pos,
                                       int* __restrict__ nl, unsigned long long* __restrict__ key) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int i = L[p];
    pos[i] = p;
    nl[p] = lab[i];
    key[i] = kNoKey;
}

// one sweep over a generation (:198-214): label and distance the point at position p ends its visit with
__global__ void k_ref_sweep(const int* __restrict__ L, int m, const int* __restrict__ nb, int k, const int* __restrict__ pos,
                            const int* __restrict__ lab, const double* __restrict__ dis, const FePt* __restrict__ P, double res,
                            const int* __restrict__ nl_prev, int* __restrict__ nl_new, double* __restrict__ nd,
                            int* __restrict__ changed) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int i = L[p];
    const int* row = nb + (size_t)i * k;
    const FePt me = P[i];
    int a = lab[i];
    double d = dis[i];
    int t0 = a, t1 = a, t2 = a;                          // labels already evaluated (a rejected label stays rejected: d only decreases)
    for (int e = 0; e < k; ++e) {
        const int j = row[e];
        const int pj = pos[j];
        const int b = pj < p ? nl_prev[pj] : lab[j];
        if (b == a || b == t0 || b == t1 || b == t2) continue;
        t2 = t1; t1 = t0; t0 = b;
        const double dd = sv_metric(me, P[b], res);
        if (dd < d) { a = b; d = dd; }
    }
    nl_new[p] = a;
    nd[p] = d;
    if (a != nl_prev[p]) *changed = 1;
}

// pushes of a converged generation (:216-230): a point whose label changed pushes every neighbour that now differs from
// it and is not in the queue (= not waiting at a later position of this generation, not pushed before: first key wins)
template <int MODE>   // 0: keys, 1: count, 2: scatter
__global__ void k_ref_push(const int* __restrict__ L, int m, const int* __restrict__ nb, int k, const int* __restrict__ pos,
                           const int* __restrict__ lab, const int* __restrict__ nl, unsigned long long* __restrict__ key,
                           int* __restrict__ cnt, int* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int i = L[p];
    int c = 0;
    if (nl[p] != lab[i]) {
        const int* row = nb + (size_t)i * k;
        const int mine = nl[p];
        const int base = MODE == 2 ? cnt[p] : 0;
        for (int e = 0; e < k; ++e) {
            const int j = row[e];
            if (j == i) continue;
            const int pj = pos[j];
            const int b = pj < p ? nl[pj] : lab[j];
            if (b == mine) continue;
            if (pj != kNone && pj > p) continue;
            const unsigned long long kk = 64ull * (unsigned long long)p + (unsigned long long)e;
            if (MODE == 0) atomicMin(&key[j], kk);
            else if (key[j] == kk) {
                if (MODE == 2) out[base + c] = j;
                ++c;
            }
        }
    }
    if (MODE == 1) cnt[p] = c;
}

__global__ void k_ref_commit(const int* __restrict__ L, int m, const int* __restrict__ nl, const double* __restrict__ nd,
                             int* __restrict__ lab, double* __restrict__ dis, int* __restrict__ pos) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int i = L[p];
    lab[i] = nl[p];
    dis[i] = nd[p];
    pos[i] = kNone;
}

// relabel (:238-248): root point -> index of the root in ascending order
__global__ void k_mark_roots(const int* __restrict__ roots, int nr, int* __restrict__ map) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nr) map[roots[r]] = r;
}
__global__ void k_relabel(int* __restrict__ lab, int n, const int* __restrict__ map) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lab[i] = map[lab[i]];
}

struct FeTrace {
    pwicp_context* ctx;
    const bool on = getenv("PWICP_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[pwicp front end/dev] %-30s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

inline dim3 grid1(long long n, int block = 256) { return dim3((unsigned)std::max<long long>(1, (n + block - 1) / block)); }

// labels (root point per point) -> refined labels, in place
int refine_device(pwicp_context* ctx, const FePt* dP, const int* d_nb, int k, int n, double res, int* d_lab) {
    hipStream_t st = ctx->stream;
    DevBuf<double> dis, nd;
    DevBuf<unsigned long long> key;
    DevBuf<int> pos, cnt, La, Lb, nla, nlb, flag, tmp;
    HIPCHK(ctx, dis.reserve((size_t)n));
    HIPCHK(ctx, nd.reserve((size_t)n));
    HIPCHK(ctx, key.reserve((size_t)n));
    HIPCHK(ctx, pos.reserve((size_t)n));
    HIPCHK(ctx, cnt.reserve((size_t)n + 1));
    HIPCHK(ctx, La.reserve((size_t)n));
    HIPCHK(ctx, Lb.reserve((size_t)n));
    HIPCHK(ctx, nla.reserve((size_t)n));
    HIPCHK(ctx, nlb.reserve((size_t)n));
    HIPCHK(ctx, flag.reserve(1));
    hipLaunchKernelGGL(k_ref_dis, grid1(n), dim3(256), 0, st, dP, d_lab, n, res, dis.p);
    hipLaunchKernelGGL(k_fill<unsigned long long>, grid1(n), dim3(256), 0, st, key.p, (long long)n, kNoKey);
    hipLaunchKernelGGL(k_fill<int>, grid1(n), dim3(256), 0, st, pos.p, (long long)n, kNone);
    hipLaunchKernelGGL(k_ref_seed_keys, grid1(n), dim3(256), 0, st, d_nb, k, n, d_lab, key.p);
    HIPCHK(ctx, hipMemsetAsync(cnt.p, 0, sizeof(int) * ((size_t)n + 1), st));
    hipLaunchKernelGGL(k_ref_seed_emit<false>, grid1(n), dim3(256), 0, st, d_nb, k, n, d_lab, key.p, cnt.p, (int*)nullptr);
    PWCHK(pw_exclusive_scan(ctx, cnt.p, (long long)n + 1, &tmp));
    int m = 0;
    HIPCHK(ctx, hipMemcpyAsync(&m, cnt.p + n, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    hipLaunchKernelGGL(k_ref_seed_emit<true>, grid1(n), dim3(256), 0, st, d_nb, k, n, d_lab, key.p, cnt.p, La.p);
    int* L = La.p;
    int* Lnext = Lb.p;
    const bool trace = getenv("PWICP_TRACE") != nullptr;
    long long pops = 0;
    int generations = 0, sweeps = 0;
    while (m > 0) {
        ++generations;
        pops += m;
        hipLaunchKernelGGL(k_ref_begin_generation, grid1(m), dim3(256), 0, st, L, m, d_lab, pos.p, nla.p, key.p);
        int* nl_prev = nla.p;
        int* nl_new = nlb.p;
        for (;;) {
            ++sweeps;
            HIPCHK(ctx, hipMemsetAsync(flag.p, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_ref_sweep, grid1(m), dim3(256), 0, st, L, m, d_nb, k, pos.p, d_lab, dis.p, dP, res, nl_prev, nl_new,
                               nd.p, flag.p);
            int changed = 0;
            HIPCHK(ctx, hipMemcpyAsync(&changed, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            std::swap(nl_prev, nl_new);
            if (!changed) break;
        }
        // (key[] of every point is kNoKey here: reset for the points of L at the start of the generation, for all at the seeds)
        hipLaunchKernelGGL(k_ref_push<0>, grid1(m), dim3(256), 0, st, L, m, d_nb, k, pos.p, d_lab, nl_prev, key.p, (int*)nullptr, (int*)nullptr);
        HIPCHK(ctx, hipMemsetAsync(cnt.p + m, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_ref_push<1>, grid1(m), dim3(256), 0, st, L, m, d_nb, k, pos.p, d_lab, nl_prev, key.p, cnt.p, (int*)nullptr);
        PWCHK(pw_exclusive_scan(ctx, cnt.p, (long long)m + 1, &tmp));
        int m_next = 0;
        HIPCHK(ctx, hipMemcpyAsync(&m_next, cnt.p + m, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        if (m_next > n) { ctx->set_err("front end: refinement queue overflow"); return PWICP_E_INTERNAL; }
        hipLaunchKernelGGL(k_ref_push<2>, grid1(m), dim3(256), 0, st, L, m, d_nb, k, pos.p, d_lab, nl_prev, key.p, cnt.p, Lnext);
        hipLaunchKernelGGL(k_ref_commit, grid1(m), dim3(256), 0, st, L, m, nl_prev, nd.p, d_lab, dis.p, pos.p);
        std::swap(L, Lnext);
        m = m_next;
    }
    if (trace) fprintf(stderr, "[pwicp front end/dev]   refinement: %lld pops, %d generations, %d sweeps\n", pops, generations, sweeps);
    return PWICP_OK;
}

}  // namespace

// Device pipeline from the k-NN graph on the host (n rows of k indices, the point itself first): PCA normals and the fusion
// on the host (stages of host/frontend.cpp), boundary refinement and relabelling on the device.
int pw_frontend_labels(pwicp_context* ctx, const float* cloud_xyz4, int n, const int32_t* nb, int k, float sv_resolution,
                       int32_t* labels, int* n_supervoxels) {
    if (k > 64) { ctx->set_err("front end: k > 64 neighbours not supported on the device"); return PWICP_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    FeTrace tr{ctx};
    std::vector<FePt> P((size_t)n);
    pwhost::fe_points_and_normals(cloud_xyz4, n, nb, k, P.data());
    tr.lap("pca normals (host)");
    const double res = (double)sv_resolution;
    const int n_sv = pwhost::fe_count_occupied_cells(P.data(), n, res);
    std::vector<int> root_of, roots;
    if (pwhost::fe_fusion_host(P.data(), nb, k, n, res, n_sv, &root_of, &roots) < 0) return PWICP_E_NOMEM;
    tr.lap("cells + fusion (host)");
    DevBuf<FePt> dP;
    DevBuf<int> d_nb, d_lab, d_roots, d_map;
    HIPCHK(ctx, dP.reserve((size_t)n));
    HIPCHK(ctx, d_nb.reserve((size_t)n * k));
    HIPCHK(ctx, d_lab.reserve((size_t)n));
    HIPCHK(ctx, d_roots.reserve(roots.size()));
    HIPCHK(ctx, d_map.reserve((size_t)n));
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(dP.p, P.data(), sizeof(FePt) * (size_t)n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(d_nb.p, nb, sizeof(int) * (size_t)n * k, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(d_lab.p, root_of.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(d_roots.p, roots.data(), sizeof(int) * roots.size(), hipMemcpyHostToDevice, st));
    tr.lap("upload");
    PWCHK(refine_device(ctx, dP.p, d_nb.p, k, n, res, d_lab.p));
    tr.lap("boundary refinement");
    hipLaunchKernelGGL(k_mark_roots, grid1((long long)roots.size()), dim3(256), 0, st, d_roots.p, (int)roots.size(), d_map.p);
    hipLaunchKernelGGL(k_relabel, grid1(n), dim3(256), 0, st, d_lab.p, n, d_map.p);
    HIPCHK(ctx, hipMemcpyAsync(labels, d_lab.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    *n_supervoxels = (int)roots.size();
    tr.lap("relabel + download");
    return PWICP_OK;
}
