// Shared host-side definitions for libpwicp.so (HIP, gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "pwicp.h"

// All translation units are compiled with -ffp-contract=off: every float/double operation
// rounds once, so the decision-relevant arithmetic (squared distances, point-to-plane
// distances, thresholds, covariance sums) is bit-identical to the reference's x64 SSE2 build.

// Per-context cache of device allocations.  hipFree waits for the whole device and hipMalloc is not cheap either: a streamed
// 5 M-point epoch spent 2.7 ms destroying its pair and ~2 ms allocating the next one's ~40 buffers, next to a 0.9 ms loop
// (tools/stream_pair_costs.py).  A buffer that a DevBuf releases goes back to the pool of the context it was allocated under and
// is only ever handed out again under that context - i.e. to work on the same stream, which runs after whatever still uses
// it.  Blocks come in size classes (<= 12.5 % above the request); the cache is capped (PWICP_POOL_MB, default 8192 per
// context), beyond that - and when the device runs out of memory - blocks are really freed.  PWICP_POOL_MB=0: no caching.
// Several contexts share a device (the auxiliary front-end contexts of a series worker, a streaming second context, pairs side
// by side): every pool is listed per device (PwPoolRegistry), and an allocation that still fails after its own pool was
// trimmed gives the cached blocks of ALL pools of that device back before it reports PWICP_E_NOMEM - memory idle in a sibling's
// cache must not fail a request.
struct PwPool;
struct PwPoolRegistry {
    std::mutex mu;
    std::vector<std::weak_ptr<PwPool>> pools;
    static PwPoolRegistry& get() { static PwPoolRegistry r; return r; }
    void add(const std::shared_ptr<PwPool>& p) {
        std::lock_guard<std::mutex> g(mu);
        size_t k = 0;
        for (auto& w : pools) if (!w.expired()) pools[k++] = w;      // (dead entries go as new ones come)
        pools.resize(k);
        pools.push_back(p);
    }
    inline void trim_device(int device);
    // last resort of an allocation that still fails after every cache of the device has been trimmed: frees what closed series /
    // pair calls have left PARKED for the next one - whole contexts with their front-end work spaces, gigabytes that no cache trim
    // reaches (set by host/registration.cpp: WorkerParking; returns whether anything was released)
    bool (*release_parked)(int device) = nullptr;    // host/registration.cpp: frees what is parked for THAT device only
};
struct PwPool {
    std::mutex mu;
    std::multimap<size_t, void*> free_blocks;       // capacity -> block
    size_t cached = 0, cap = 0;
    int device = -1;                                // set when the pool's context is created (pwicp_create)
    PwPool() {
        const char* e = getenv("PWICP_POOL_MB");
        cap = (size_t)(e ? std::max(atol(e), 0L) : 8192L) << 20;
    }
    PwPool(const PwPool&) = delete;
    PwPool& operator=(const PwPool&) = delete;
    ~PwPool() { trim(); }
    static size_t size_class(size_t bytes) {
        if (bytes < 4096) return 4096;
        size_t step = (size_t)1 << 9;               // granule = 2^(floor(log2 bytes) - 3), at least 512 B
        while ((step << 4) <= bytes) step <<= 1;
        return (bytes + step - 1) / step * step;
    }
    void trim() {
        std::lock_guard<std::mutex> g(mu);
        for (auto& kv : free_blocks) (void)hipFree(kv.second);
        free_blocks.clear();
        cached = 0;
    }
    hipError_t take(size_t bytes, void** out, size_t* cap_out) {
        const size_t c = size_class(bytes);
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = free_blocks.find(c);
            if (it != free_blocks.end()) {
                *out = it->second; *cap_out = c;
                cached -= c;
                free_blocks.erase(it);
                return hipSuccess;
            }
        }
        hipError_t e = hipMalloc(out, c);
        if (e == hipErrorOutOfMemory) {             // give the cached blocks back and try once more
            (void)hipGetLastError();
            trim();
            e = hipMalloc(out, c);
            if (e == hipErrorOutOfMemory) {         // ... then what the other contexts of this device keep cached
                (void)hipGetLastError();
                PwPoolRegistry::get().trim_device(device);
                e = hipMalloc(out, c);
                if (e == hipErrorOutOfMemory && PwPoolRegistry::get().release_parked) {     // ... then the parked contexts
                    (void)hipGetLastError();
                    // (destroying a parked context selects ITS device: the caller's device is put back before the retry)
                    int cur = device;
                    (void)hipGetDevice(&cur);
                    const bool freed = PwPoolRegistry::get().release_parked(device);
                    (void)hipSetDevice(cur);
                    if (freed) e = hipMalloc(out, c);
                }
            }
        }
        *cap_out = c;
        return e;
    }
    void give(void* p, size_t c) {
        {
            std::lock_guard<std::mutex> g(mu);
            if (cached + c <= cap) { free_blocks.emplace(c, p); cached += c; return; }
        }
        (void)hipFree(p);
    }
};

inline void PwPoolRegistry::trim_device(int device) {
    std::vector<std::shared_ptr<PwPool>> live;
    {
        std::lock_guard<std::mutex> g(mu);
        for (auto& w : pools)
            if (auto p = w.lock()) if (p->device == device || device < 0) live.push_back(p);
    }
    for (auto& p : live) p->trim();                 // (a cached block is not in use by definition; hipFree waits for the device)
}

struct pwicp_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::shared_ptr<PwPool> pool = std::make_shared<PwPool>();
    std::string err;
    int n_cu = 256;
    std::shared_ptr<void> scratch;        // grow-only work buffers a stage keeps between calls (csrc/frontend.hip), freed with the context
    std::shared_ptr<void> host_slot;      // what the host stages keep with a context between calls (host/registration.cpp: the
                                          // auxiliary contexts of the front ends), freed with the context
    void set_err(const char* where, hipError_t e) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s: %s", where, hipGetErrorString(e));
        err = buf;
    }
    void set_err(const char* msg) { err = msg; }
};


#define PW_STR2(x) #x
#define PW_STR(x) PW_STR2(x)
// (the allocation pool a thread is working under: DevBuf::reserve takes blocks from it.  Every entry point passes through
// HIPCHK with its context before it allocates.  The thread holds the POOL, shared, not the context: a context that another
// thread destroys leaves nothing dangling here - its pool lives until the last holder lets go.)
inline thread_local std::shared_ptr<PwPool> pw_tls_pool;
inline void pw_tls_enter(pwicp_context* ctx) {
    if (pw_tls_pool.get() != ctx->pool.get()) pw_tls_pool = ctx->pool;
}
#define HIPCHK(ctx, expr)                                                   \
    do {                                                                    \
        pw_tls_enter(ctx);                                                  \
        hipError_t e__ = (expr);                                            \
        if (e__ != hipSuccess) {                                            \
            (ctx)->set_err(__FILE__ ":" PW_STR(__LINE__) " " #expr, e__);   \
            return e__ == hipErrorOutOfMemory ? PWICP_E_NOMEM : PWICP_E_NO_DEVICE; \
        }                                                                   \
    } while (0)
#define PWCHK(expr)                        \
    do {                                   \
        int s__ = (expr);                  \
        if (s__ != PWICP_OK) return s__;   \
    } while (0)

// Device buffer with explicit lifetime (no exceptions across the C ABI).
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    std::shared_ptr<PwPool> pool;        // where the block came from and goes back to (null: plain hipMalloc / hipFree)
    size_t cap_bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            if (pool) pool->give(p, cap_bytes);
            else (void)hipFree(p);
        }
        p = nullptr;
        n = 0;
        cap_bytes = 0;
        pool.reset();
    }
    void swap(DevBuf& o) {
        T* tp = p; p = o.p; o.p = tp;
        size_t tn = n; n = o.n; o.n = tn;
        pool.swap(o.pool);
        size_t tc = cap_bytes; cap_bytes = o.cap_bytes; o.cap_bytes = tc;
    }
    // grows only; contents are NOT preserved
    hipError_t reserve(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p && count * sizeof(T) <= cap_bytes) { n = count; return hipSuccess; }     // (the block's size class already holds it)
        release();
        if (count == 0) count = 1;
        hipError_t e;
        if (pw_tls_pool && pw_tls_pool->cap > 0) {
            pool = pw_tls_pool;
            e = pool->take(count * sizeof(T), (void**)&p, &cap_bytes);
            if (e != hipSuccess) { p = nullptr; pool.reset(); cap_bytes = 0; }
        } else {
            e = hipMalloc((void**)&p, count * sizeof(T));
        }
        if (e == hipSuccess) n = count;
        return e;
    }
};

static inline int div_up(long long a, int b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// Uniform-grid search structure over a static point set (device resident).
// Points are bucketed by cell with a counting sort; a cell row along x is contiguous in memory,
// so the 3 x-neighbour cells of a stencil row are ONE contiguous range of points.
// ---------------------------------------------------------------------------------------------
struct GridLevel {
    float ox, oy, oz;        // origin = min corner of the bounding box
    float h, inv_h;          // cell edge
    float inv_hy, inv_hz;    // = inv_h, or 0 for a level of COLUMNS along that axis (ny or nz = 1), see pw_grid_build
    float slack;             // bound on |computed cell boundary - true boundary| (rounding)
    int nx, ny, nz;
    int n;                   // number of points
    int perm;                // axis roles of this level: 0 = (x, y, z); 1 = the level is built on (y, z, x), 2 = on (z, x, y) —
                             // pts hold the coordinates in THAT order, origin / counts refer to it, queries are permuted on
                             // entry and squared distances are still summed in the original x, y, z order (nn_device.h).
                             // Only the small-cell levels of the dense search use 1 / 2 (pw_grid_add_dense).
    const int* cell_start;   // nx*ny*nz + 1 entries
    const float4* pts;       // sorted by cell; w = __int_as_float(original index)
    const float* pts3;       // the same points packed x, y, z (12 B each): levels of the dense search only, else nullptr
};

// Two levels over the same points: `fine` serves the common 27-cell stencil, `coarse` (2x the edge) resolves
// far queries without walking many empty fine cells.
struct GridDesc {
    GridLevel fine, coarse;
};

struct Grid {
    GridDesc d{};
    DevBuf<int> cell_start, ccell_start;
    DevBuf<float4> pts, cpts;
    double kbar27 = 0.0;     // mean #points in the fine 27-cell stencil around an occupied cell, point weighted
    // optional third level of SMALL cells for the disc-pruned dense search (pw_grid_add_dense), same layout as `fine`
    GridLevel dense{};
    bool has_dense = false;
    DevBuf<int> dcell_start;
    DevBuf<float4> dpts;
    DevBuf<float> dpts3;
    // when `fine` is a grid of cells: the same small cells as columns, for pairs whose queries lie on gentle parts
    GridLevel dense_alt{};
    bool has_dense_alt = false;
    DevBuf<int> acell_start;
    DevBuf<float4> apts;
    DevBuf<float> apts3;
    // ... and with other axis roles (rows along y or z, columns along x): a face that no layout on (x, y, z) fits
    struct Extra {
        GridLevel lv{};
        bool has = false;
        DevBuf<int> cell_start;
        DevBuf<float4> pts;
        DevBuf<float> pts3;
    } extra[3];
};

// grid.hip
int pw_grid_build(pwicp_context* ctx, const float4* d_pts, int n, float cell_edge, Grid* g);
// exact 1-NN of d_q[0..nq) ; d_idx may be null; d_examined (optional) accumulates #points examined
int pw_nn_launch(pwicp_context* ctx, const GridDesc& g, const float4* d_q, int nq, int* d_idx, float* d_d2,
                 unsigned long long* d_examined);
// buffers of the far-query hand-over of a dense launch (grid.hip: k_nn_dense_far); far_bufs != nullptr selects it
struct DenseFarBuffers {
    DevBuf<float4> q;
    DevBuf<int> slot;
    DevBuf<unsigned> count;
};
// dense NN of patch points: query i = patch point qorder[i] (skipped, sentinel written, unless stable[pt_patch[.]])
// d_qpatch (optional with `dense`): d_pt_patch[d_qorder[i]] precomputed; dense != nullptr selects the disc-pruned kernel
int pw_nn_dense_launch(pwicp_context* ctx, const GridDesc& g, const float4* d_pat, const int* d_qorder,
                           const int* d_pt_patch, const int* d_stable, int nq, float* d_d2,
                           unsigned long long* d_examined, const GridLevel* dense = nullptr, const int* d_qpatch = nullptr,
                           const struct FusedSelect* fs = nullptr, struct DenseFarBuffers* far_bufs = nullptr,
                           const float4* d_patq = nullptr);
// d_patq (optional, disc-pruned kernel): d_pat[d_qorder[i]] precomputed - the queries themselves in launch order
int pw_gather_f4_launch(pwicp_context* ctx, const float4* d_src, const int* d_order, int n, float4* d_out);
// passes 1 / 2 of the fused percentile selection (select_dev.h) as launches of their own
int pw_fs_pass_launch(pwicp_context* ctx, int pass, const struct FusedSelect& fs);
int pw_grid_add_dense(pwicp_context* ctx, const float4* d_pts, int n, float cell_edge, Grid* g);
// far_frac (optional): share of the probed queries that leave the small-cell search at their initial position
int pw_dense_level_for(pwicp_context* ctx, const Grid& g, const float4* d_q, int nq, const GridLevel** out, double* far_frac = nullptr);
// out[i] = src[order[i]]
int pw_gather_int_launch(pwicp_context* ctx, const int* d_src, const int* d_order, int n, int* d_out);
int pw_morton_order(pwicp_context* ctx, const GridDesc& g, const float4* d_pts, int n, DevBuf<int>* order, const GridLevel* strip_lv = nullptr);
int pw_bbox(pwicp_context* ctx, const float4* d_pts, int n, float mn[3], float mx[3]);
int pw_check_finite(pwicp_context* ctx, const float4* d_pts, int n);
int pw_knn_launch(pwicp_context* ctx, const GridDesc& g, int k, int* d_nb, int* d_rev_count = nullptr);
float pw_estimate_cell_edge(const float* xyz4, int n);     // cell edge of a stand-alone search grid (~2x the point spacing)
int pw_knn_mean_dist_launch(pwicp_context* ctx, const GridDesc& g, int mean_k, float* d_mean);
// k-th smallest (0-based) of the non-sentinel entries of n non-negative floats; result written to d_out[0]; scratch >= 3*2048+8 uints
// optional host-mailbox message of a selection: the selected value's bits -> dst[0], then seq (system-scope release)
// ---- device -> host mailbox hand-over (pinned, host-coherent memory) --------------------------------------------------------
// The payload words go out as system-scope write-through stores, every lane that stored drains its store queue (gfx9 counts
// stores in vmcnt; a system-scope store is acknowledged by the fabric), then ONE lane stores the sequence word.  That is the
// part of __threadfence_system() a mailbox needs; the other part - writing the whole L2 back and invalidating the L1 - costs
// 1.5-3 us behind a launch that has just rewritten the clouds, for words that never were in a cache.  gfx950 only (see icp.hip).
#if defined(__HIPCC__)
__device__ __forceinline__ void mail_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void mail_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }     // every lane that called mail_store
__device__ __forceinline__ void mail_publish(unsigned* seq_ptr, unsigned seq) {                        // one lane, after the drain
    __hip_atomic_store(seq_ptr, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

struct SelectMail {
    unsigned* dst = nullptr;
    unsigned* seq_ptr = nullptr;
    unsigned seq = 0;
};
int pw_select_kth_launch(pwicp_context* ctx, const float* d_vals, int n, int k, unsigned* d_scratch, float* d_out,
                         bool armed = false, const SelectMail* mail = nullptr);
// count of values with sqrtf(v) < thr  -> d_count[0]
int pw_count_below_launch(pwicp_context* ctx, const float* d_d2, int n, float thr, unsigned* d_count);
// exclusive scan in place of n ints (n may be large); d_tmp >= div_up(n,4096)+1 ints
int pw_exclusive_scan(pwicp_context* ctx, int* d_data, long long n, DevBuf<int>* tmp);
