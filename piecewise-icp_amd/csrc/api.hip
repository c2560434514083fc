// C ABI entry points of libpwicp.so: context management and the stand-alone building blocks.
// The pair-level loop (pwicp_pair_*) lives in loop.hip.
#include <cmath>
#include <new>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "icp.h"
#include "patch.h"

namespace {
// Streams of destroyed contexts, per device, for the next pwicp_create.  The runtime gives every NEW stream the next hardware
// queue (up to $GPU_MAX_HW_QUEUES) and keeps the queue of a destroyed stream allocated: after four contexts had come and gone, the
// streams of a series sat on queues 4 - 7 with queues 0 - 3 idle but alive, and every launch of the front ends (thousands of small
// dependent ones) was slower - 8 x 1 M points 0.41 -> 0.45 s (tools/series_repeat.py, SERIES_REPEAT_PRE=ctx).  A context that
// takes over a parked stream sits on the queue the old one had.  (Idle streams are never destroyed: a few hundred bytes each.)
struct StreamPool {
    std::mutex mu;
    std::map<int, std::vector<hipStream_t>> idle;
    static StreamPool& get() { static StreamPool* p = new StreamPool; return *p; }     // (never destructed: no HIP call at exit)
};
// The runtime serves a process's streams from FOUR hardware queues unless $GPU_MAX_HW_QUEUES says otherwise, and it reads the
// variable when it starts.  A series worker runs five streams (its own and four front ends), pairs side by side one each: on four
// queues two of them share one and wait for each other - 8 x 1 M points 0.40 - 0.43 s instead of 0.37 s.  The library does NOT
// edit the host's environment (setenv races with any getenv of a multithreaded host and changes the runtime for the host's
// other streams): the callers that own their process set the variable before the first HIP call (bench.py, pwicp_demo, the
// Python binding at import), and a series that finds fewer than five queues configured says so once (hw_queue_hint,
// host/registration.cpp).
}  // namespace

extern "C" {

const char* pwicp_version(void) { return "pwicp-mi355x 0.1 (gfx950, HIP)"; }

int pwicp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pwicp_create(pwicp_context** out, int device_id) {
    if (!out) return PWICP_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PWICP_E_NO_DEVICE;   // fail loudly: no CPU fallback
    if (device_id < 0 || device_id >= n) return PWICP_E_INVALID;
    pwicp_context* ctx = new (std::nothrow) pwicp_context();
    if (!ctx) return PWICP_E_NOMEM;
    ctx->device = device_id;
    ctx->pool->device = device_id;
    PwPoolRegistry::get().add(ctx->pool);
    if (hipSetDevice(device_id) != hipSuccess) { delete ctx; return PWICP_E_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    {
        StreamPool& sp = StreamPool::get();
        std::lock_guard<std::mutex> g(sp.mu);
        auto& v = sp.idle[device_id];
        if (!v.empty()) { ctx->stream = v.back(); v.pop_back(); }
    }
    if (!ctx->stream && hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return PWICP_E_NO_DEVICE; }
    *out = ctx;
    return PWICP_OK;
}

void pwicp_destroy(pwicp_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->scratch.reset();
    if (pw_tls_pool.get() == ctx->pool.get()) pw_tls_pool.reset();
    if (ctx->pool) ctx->pool->trim();           // (the pool itself lives as long as a buffer of this context does)
    if (ctx->stream) {                           // (synchronised above: nothing of this context is left on it)
        StreamPool& sp = StreamPool::get();
        std::lock_guard<std::mutex> g(sp.mu);
        sp.idle[ctx->device].push_back(ctx->stream);
        ctx->stream = nullptr;
    }
    delete ctx;
}

}  // extern "C"

// host/registration.cpp (WorkerParking): what an allocation that is out of device memory may release as its last resort
bool pw_set_release_parked_hook(bool (*fn)(int)) { PwPoolRegistry::get().release_parked = fn; return true; }

extern "C" {

int pwicp_context_device(const pwicp_context* ctx) { return ctx ? ctx->device : -1; }

const char* pwicp_last_error(const pwicp_context* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

}  // extern "C"

namespace {

// cell edge for a stand-alone search: ~2x the mean spacing of a surface-sampled cloud, from its extent
float estimate_cell_edge(const float* t4, int nt) {
    float edge = 0.f;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < nt; ++i)
        for (int k = 0; k < 3; ++k) {
            float v = t4[4 * (size_t)i + k];
            if (v < mn[k]) mn[k] = v;
            if (v > mx[k]) mx[k] = v;
        }
    if (nt > 0) {
        double ex = (double)mx[0] - mn[0], ey = (double)mx[1] - mn[1], ez = (double)mx[2] - mn[2];
        double a = std::max(ex * ey, std::max(ex * ez, ey * ez));     // largest face ~ surface area
        double l = std::max(ex, std::max(ey, ez));
        double spacing = a > 0 ? std::sqrt(a / (double)nt) : (l > 0 ? l / (double)nt : 1.0);
        edge = (float)(2.0 * spacing);
    }
    if (!(edge > 0.f) || !std::isfinite(edge)) edge = 1.f;
    return edge;
}

int upload(pwicp_context* ctx, const float* h, int n, DevBuf<float4>* d) {
    HIPCHK(ctx, d->reserve((size_t)(n > 0 ? n : 1)));
    if (n > 0) HIPCHK(ctx, hipMemcpyAsync(d->p, h, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    return PWICP_OK;
}

// NN of cloud2 (queries) in cloud1 (targets) -> device d2 (and optionally idx)
int nn_host_clouds(pwicp_context* ctx, const float* t4, int nt, const float* q4, int nq, DevBuf<int>* idx,
                   DevBuf<float>* d2) {
    if (!ctx) return PWICP_E_INVALID;
    if (!t4 || !q4 || nt < 0 || nq < 0) { ctx->set_err("null pointer / negative size"); return PWICP_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf<float4> dt, dq;
    PWCHK(upload(ctx, t4, nt, &dt));
    PWCHK(upload(ctx, q4, nq, &dq));
    Grid g;
    const float edge = estimate_cell_edge(t4, nt);
    PWCHK(pw_grid_build(ctx, dt.p, nt, edge, &g));
    PWCHK(pw_check_finite(ctx, dq.p, nq));
    HIPCHK(ctx, d2->reserve((size_t)(nq > 0 ? nq : 1)));
    if (idx) HIPCHK(ctx, idx->reserve((size_t)(nq > 0 ? nq : 1)));
    PWCHK(pw_nn_launch(ctx, g.d, dq.p, nq, idx ? idx->p : nullptr, d2->p, nullptr));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PWICP_OK;
}

}  // namespace

extern "C" {

int pwicp_nn_search(pwicp_context* ctx, const float* target_xyz4, int n_target, const float* query_xyz4,
                    int n_query, int32_t* index_match, float* sq_distance) {
    if (!ctx) return PWICP_E_INVALID;
    if (!index_match || !sq_distance) { ctx->set_err("null output"); return PWICP_E_INVALID; }
    DevBuf<int> idx;
    DevBuf<float> d2;
    PWCHK(nn_host_clouds(ctx, target_xyz4, n_target, query_xyz4, n_query, &idx, &d2));
    if (n_query > 0) {
        HIPCHK(ctx, hipMemcpy(index_match, idx.p, (size_t)n_query * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(sq_distance, d2.p, (size_t)n_query * sizeof(float), hipMemcpyDeviceToHost));
    }
    return PWICP_OK;
}

int pwicp_percentile_dist(pwicp_context* ctx, const float* cloud1_xyz4, int n1, const float* cloud2_xyz4, int n2,
                          float percentile, double* dist_out) {
    if (!ctx) return PWICP_E_INVALID;
    if (!dist_out || n2 <= 0 || n1 <= 0) { ctx->set_err("empty cloud / null output"); return PWICP_E_INVALID; }
    DevBuf<float> d2;
    PWCHK(nn_host_clouds(ctx, cloud1_xyz4, n1, cloud2_xyz4, n2, nullptr, &d2));
    DevBuf<unsigned> scratch;
    DevBuf<float> out;
    HIPCHK(ctx, scratch.reserve(8 + 3 * 2048));
    HIPCHK(ctx, out.reserve(1));
    int k = (int)((float)n2 * percentile);      // C.cpp:177: int leftnum = n * percentile
    if (k >= n2) k = n2 - 1;
    if (k < 0) k = 0;
    PWCHK(pw_select_kth_launch(ctx, d2.p, n2, k, scratch.p, out.p));
    float v = 0.f;
    HIPCHK(ctx, hipMemcpyAsync(&v, out.p, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *dist_out = (double)sqrtf(v);               // C.cpp:277: sqrt(float) then widened
    return PWICP_OK;
}

int pwicp_overlap_ratio(pwicp_context* ctx, const float* cloud1_xyz4, int n1, const float* cloud2_xyz4, int n2,
                        float DTinit, float* ratio_out) {
    if (!ctx) return PWICP_E_INVALID;
    if (!ratio_out || n2 <= 0 || n1 <= 0) { ctx->set_err("empty cloud / null output"); return PWICP_E_INVALID; }
    DevBuf<float> d2;
    PWCHK(nn_host_clouds(ctx, cloud1_xyz4, n1, cloud2_xyz4, n2, nullptr, &d2));
    DevBuf<unsigned> cnt;
    HIPCHK(ctx, cnt.reserve(1));
    PWCHK(pw_count_below_launch(ctx, d2.p, n2, DTinit, cnt.p));
    unsigned c = 0;
    HIPCHK(ctx, hipMemcpyAsync(&c, cnt.p, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *ratio_out = (float)c / (float)n2;          // R.cpp:613
    return PWICP_OK;
}

}  // extern "C"

extern "C" {

}  // extern "C"
float pw_estimate_cell_edge(const float* xyz4, int n) { return estimate_cell_edge(xyz4, n); }
extern "C" {

int pwicp_knn(pwicp_context* ctx, const float* cloud_xyz4, int n, int k, float cell_edge, int32_t* neighbors) {
    if (!ctx) return PWICP_E_INVALID;
    if (!cloud_xyz4 || !neighbors || n <= 0 || k <= 0 || k > n) { ctx->set_err("pwicp_knn: invalid argument"); return PWICP_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool trace = getenv("PWICP_TRACE") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[pwicp knn] %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    DevBuf<float4> pts;
    PWCHK(upload(ctx, cloud_xyz4, n, &pts));
    lap("upload");
    Grid g;
    PWCHK(pw_grid_build(ctx, pts.p, n, cell_edge > 0.f ? cell_edge : estimate_cell_edge(cloud_xyz4, n), &g));
    lap("grid");
    DevBuf<int> nb;
    HIPCHK(ctx, nb.reserve((size_t)n * k));
    PWCHK(pw_knn_launch(ctx, g.d, k, nb.p));
    lap("alloc + kernel");
    HIPCHK(ctx, hipMemcpy(neighbors, nb.p, (size_t)n * k * sizeof(int), hipMemcpyDeviceToHost));
    lap("download");
    return PWICP_OK;
}

// calPCresolution (C.cpp:239-263): float mean of the distance of every point to its nearest other point.  The
// per-point distances come from the device (float-metric 2-NN, the point itself first); they are summed on the host
// in point order, in float, as the reference does.
int pwicp_pc_resolution_dev(pwicp_context* ctx, const float* cloud_xyz4, int n, float* resolution) {
    if (!ctx) return PWICP_E_INVALID;
    if (!cloud_xyz4 || !resolution || n < 2) { ctx->set_err("pwicp_pc_resolution_dev: invalid argument"); return PWICP_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf<float4> pts;
    PWCHK(upload(ctx, cloud_xyz4, n, &pts));
    Grid g;
    PWCHK(pw_grid_build(ctx, pts.p, n, estimate_cell_edge(cloud_xyz4, n), &g));
    DevBuf<float> d;
    HIPCHK(ctx, d.reserve((size_t)n));
    PWCHK(pw_knn_mean_dist_launch(ctx, g.d, 1, d.p));
    std::vector<float> h((size_t)n);
    HIPCHK(ctx, hipMemcpy(h.data(), d.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    float res = 0.0f;
    for (int i = 0; i < n; ++i) res += h[(size_t)i];
    *resolution = res / (float)n;
    return PWICP_OK;
}

}  // extern "C"

// ---- patch-level and ICP building blocks -------------------------------------------------------------------
extern "C" {

int pwicp_patch_normals(pwicp_context* ctx, const float* patch_xyz4, const int32_t* offsets, int n_patches,
                        float* normals4, uint8_t* ok) {
    if (!ctx) return PWICP_E_INVALID;
    if (!patch_xyz4 || !offsets || !normals4 || n_patches < 0) { ctx->set_err("null pointer / negative size"); return PWICP_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int m = n_patches, tot = m > 0 ? offsets[m] : 0;
    DevBuf<float4> pat, nrm;
    DevBuf<int> off;
    PWCHK(upload(ctx, patch_xyz4, tot, &pat));
    HIPCHK(ctx, off.reserve((size_t)m + 1));
    HIPCHK(ctx, nrm.reserve((size_t)std::max(m, 1)));
    HIPCHK(ctx, hipMemcpyAsync(off.p, offsets, ((size_t)m + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    PWCHK(pw_patch_normals_launch(ctx, pat.p, off.p, m, nrm.p));
    std::vector<float> h((size_t)std::max(m, 1) * 4);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), nrm.p, (size_t)m * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < m; ++i) {
        const bool good = h[4 * (size_t)i + 3] != 0.0f;
        if (ok) ok[i] = good ? 1 : 0;
        normals4[4 * (size_t)i + 0] = h[4 * (size_t)i + 0];
        normals4[4 * (size_t)i + 1] = h[4 * (size_t)i + 1];
        normals4[4 * (size_t)i + 2] = h[4 * (size_t)i + 2];
        normals4[4 * (size_t)i + 3] = 0.0f;
    }
    return PWICP_OK;
}

// Per-patch centroid, six boundary points and sigmas of GIVEN patches: calPatchCTandBP (S.cpp:260-303), calPatchSTD
// (C.cpp:336-354; decl C.h:150) and calBPandCTSTD (S.cpp:306-321; decl S.h:451-452) for all patches at once.  Any output
// may be NULL.
int pwicp_patch_stats(pwicp_context* ctx, const float* patch_xyz4, const int32_t* offsets, int n_patches, float* centroid_xyz4,
                      float* boundary_xyz4, float* std_bp, float* std_ct) {
    if (!ctx) return PWICP_E_INVALID;
    if (!patch_xyz4 || !offsets || n_patches < 0) { ctx->set_err("null pointer / negative size"); return PWICP_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int m = n_patches, tot = m > 0 ? offsets[m] : 0;
    if (m == 0) return PWICP_OK;
    DevBuf<float4> pat, ct, bp;
    DevBuf<int> off;
    DevBuf<float> sb, sc;
    PWCHK(upload(ctx, patch_xyz4, tot, &pat));
    HIPCHK(ctx, off.reserve((size_t)m + 1));
    HIPCHK(ctx, ct.reserve((size_t)m));
    HIPCHK(ctx, bp.reserve((size_t)m * 6));
    HIPCHK(ctx, sb.reserve((size_t)m));
    HIPCHK(ctx, sc.reserve((size_t)m));
    HIPCHK(ctx, hipMemcpyAsync(off.p, offsets, ((size_t)m + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    PWCHK(pw_patch_stats_launch(ctx, pat.p, off.p, m, ct.p, bp.p, sb.p, sc.p));
    if (centroid_xyz4) HIPCHK(ctx, hipMemcpyAsync(centroid_xyz4, ct.p, (size_t)m * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (boundary_xyz4) HIPCHK(ctx, hipMemcpyAsync(boundary_xyz4, bp.p, (size_t)m * 6 * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (std_bp) HIPCHK(ctx, hipMemcpyAsync(std_bp, sb.p, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (std_ct) HIPCHK(ctx, hipMemcpyAsync(std_ct, sc.p, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PWICP_OK;
}

int pwicp_select_patches(pwicp_context* ctx, const float* cloud_xyz4, int n, const int32_t* labels, int n_supervoxels,
                         int* n_patches, int* n_patch_points, float* patch_xyz4, int32_t* offsets, int32_t* src_index,
                         float* centroid_xyz4, float* boundary_xyz4, float* std_bp, float* std_ct) {
    if (!ctx) return PWICP_E_INVALID;
    if (!cloud_xyz4 || !labels || n < 0 || n_supervoxels < 0 || !n_patches || !n_patch_points) {
        ctx->set_err("null pointer / negative size"); return PWICP_E_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf<float4> cloud;
    DevBuf<int> lab;
    PWCHK(upload(ctx, cloud_xyz4, n, &cloud));
    HIPCHK(ctx, lab.reserve((size_t)std::max(n, 1)));
    if (n > 0) HIPCHK(ctx, hipMemcpyAsync(lab.p, labels, (size_t)n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    PatchSet P;
    PWCHK(pw_select_patches_dev(ctx, cloud.p, n, lab.p, n_supervoxels, &P));
    *n_patches = P.m;
    *n_patch_points = P.tot;
    if (!patch_xyz4) return PWICP_OK;
    if (!offsets || !centroid_xyz4 || !boundary_xyz4 || !std_bp || !std_ct) { ctx->set_err("null output"); return PWICP_E_INVALID; }
    const size_t m = (size_t)P.m, tot = (size_t)P.tot;
    HIPCHK(ctx, hipMemcpyAsync(offsets, P.off.p, (m + 1) * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (tot) HIPCHK(ctx, hipMemcpyAsync(patch_xyz4, P.pat.p, tot * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (tot && src_index) HIPCHK(ctx, hipMemcpyAsync(src_index, P.src.p, tot * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (m) {
        HIPCHK(ctx, hipMemcpyAsync(centroid_xyz4, P.ct.p, m * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(boundary_xyz4, P.bp.p, m * 6 * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(std_bp, P.bpstd.p, m * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(std_ct, P.ctstd.p, m * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PWICP_OK;
}

int pwicp_p2p_icp(pwicp_context* ctx, const float* target_xyz4, const float* target_normal4, int n_target,
                  const float* source_xyz4, const float* source_normal4, int n_source, double euclid_eps, float* T16,
                  int* n_iterations) {
    if (!ctx) return PWICP_E_INVALID;
    if (!target_xyz4 || !target_normal4 || !source_xyz4 || !source_normal4 || !T16 || n_target < 0 || n_source < 0) {
        ctx->set_err("null pointer / negative size"); return PWICP_E_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf<float4> tgt, tgtn;
    PWCHK(upload(ctx, target_xyz4, n_target, &tgt));
    PWCHK(upload(ctx, target_normal4, n_target, &tgtn));
    Grid g;
    PWCHK(pw_grid_build(ctx, tgt.p, n_target, estimate_cell_edge(target_xyz4, n_target), &g));
    IcpWork w;
    PWCHK(w.reserve(ctx, n_source));
    if (n_source > 0) {
        HIPCHK(ctx, hipMemcpyAsync(w.src.p, source_xyz4, (size_t)n_source * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(w.srcn.p, source_normal4, (size_t)n_source * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    }
    return pw_icp_run(ctx, g.d, tgt.p, tgtn.p, &w, n_source, euclid_eps, T16, n_iterations);
}

int pwicp_trans_para_vcm(pwicp_context* ctx, const float* target_xyz4, const float* target_normal4, int n_target,
                         const float* source_stable_xyz4, int n_source, double* VCM36) {
    if (!ctx) return PWICP_E_INVALID;
    if (!target_xyz4 || !target_normal4 || !source_stable_xyz4 || !VCM36 || n_target <= 0 || n_source < 0) {
        ctx->set_err("null pointer / bad size"); return PWICP_E_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf<float4> tgt, tgtn, src;
    PWCHK(upload(ctx, target_xyz4, n_target, &tgt));
    PWCHK(upload(ctx, target_normal4, n_target, &tgtn));
    PWCHK(upload(ctx, source_stable_xyz4, n_source, &src));
    Grid g;
    PWCHK(pw_grid_build(ctx, tgt.p, n_target, estimate_cell_edge(target_xyz4, n_target), &g));
    IcpWork w;
    PWCHK(w.reserve(ctx, n_source));
    return pw_vcm_run(ctx, g.d, tgt.p, tgtn.p, &w, src.p, n_source, VCM36);
}

}  // extern "C"
