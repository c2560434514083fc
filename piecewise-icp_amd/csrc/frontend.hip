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
//     queue generation first) are rebuilt afterwards from (position, neighbour) keys with an atomic min;
//   * which units a sweep re-runs, and whether a unit may already see what its neighbours in the order decided in the SAME
//     sweep (chunks of consecutive fusion centres inside a wavefront), only changes the number of sweeps: any mixture of
//     guesses is admissible because a fusion round is only closed by a sweep over all units, from the standing state alone,
//     that changes nothing (the certificate in fusion_device).
//
// File map: boundary refinement (k_ref_*, refine_device) | fusion (FusState, k_fus_run and the bookkeeping kernels of a
// sweep, hand-over between rounds) | normals scatter, occupied cells, lambda0 select | FeWorkspace | fusion_device |
// pw_frontend_segment_device.
//
// All decisions use the reference's double arithmetic (no contraction: the library is built with -ffp-contract=off;
// sqrt and the division are IEEE on gfx950), so labels are identical to the host pipeline's, which the tests assert.
#include <atomic>
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <climits>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
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
    for (int e0 = 0; e0 < k; e0 += 8) {                  // (neighbours and their labels eight at a time)
        int j[8], lj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) j[u] = (e0 + u < k) ? row[e0 + u] : i;
#pragma unroll
        for (int u = 0; u < 8; ++u) lj[u] = lab[j[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u < k && lj[u] != li) {
                any = true;
                atomicMin(&key[j[u]], 65ull * (unsigned long long)i + (unsigned long long)(e0 + u + 1));
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
    for (int e0 = 0; e0 < k; e0 += 8) {                  // (neighbours and their labels eight at a time)
        int j[8], lj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) j[u] = (e0 + u < k) ? row[e0 + u] : i;
#pragma unroll
        for (int u = 0; u < 8; ++u) lj[u] = lab[j[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u;
            if (e < k && lj[u] != li && key[j[u]] == 65ull * (unsigned long long)i + (unsigned long long)(e + 1)) {
                if (SCATTER) out[base + c] = j[u];
                ++c;
            }
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
    // the neighbours' labels eight at a time - indices, positions, labels: three round trips per eight neighbours instead of
    // three per neighbour (nothing the loop reads is written by this launch) - then the visit in neighbour order as before
    constexpr int C = 8;
    for (int e0 = 0; e0 < k; e0 += C) {
        int j[C], pj[C], b[C];
#pragma unroll
        for (int u = 0; u < C; ++u) j[u] = (e0 + u < k) ? row[e0 + u] : i;
#pragma unroll
        for (int u = 0; u < C; ++u) pj[u] = pos[j[u]];
#pragma unroll
        for (int u = 0; u < C; ++u) b[u] = pj[u] < p ? nl_prev[pj[u]] : lab[j[u]];
#pragma unroll
        for (int u = 0; u < C; ++u) {
            if (e0 + u >= k) break;
            const int bb = b[u];
            if (bb == a || bb == t0 || bb == t1 || bb == t2) continue;
            t2 = t1; t1 = t0; t0 = bb;
            const double dd = sv_metric(me, P[bb], res);
            if (dd < d) { a = bb; d = dd; }
        }
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
        for (int e0 = 0; e0 < k; e0 += 8) {              // (indices, positions and labels of eight neighbours at a time)
            int j8[8], pj8[8], b8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) j8[u] = (e0 + u < k) ? row[e0 + u] : i;
#pragma unroll
            for (int u = 0; u < 8; ++u) pj8[u] = pos[j8[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) b8[u] = pj8[u] < p ? nl[pj8[u]] : lab[j8[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u, j = j8[u], pj = pj8[u];
                if (e >= k || j == i) continue;
                if (b8[u] == mine) continue;
                if (pj != kNone && pj > p) continue;
                const unsigned long long kk = 64ull * (unsigned long long)p + (unsigned long long)e;
                if (MODE == 0) atomicMin(&key[j], kk);
                else if (key[j] == kk) {
                    if (MODE == 2) out[base + c] = j;
                    ++c;
                }
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

// ================================================================================================================
// Supervoxel fusion (supervoxel_segmentation.h:104-170)
// ================================================================================================================
// One round (one value of lambda) of the reference visits the current supervoxel roots ("centres") in ascending order;
// centre i runs a breadth-first search over the roots adjacent to it, absorbs every root j with
// lambda - size(j) * metric(i, j) > 0 (its neighbours join the search) and keeps the others as its new adjacency list.
// What centre i sees depends on the centres before it: a root absorbed earlier is seen through its absorber, an earlier
// centre has its new size and adjacency.  State as of "time i" for a node x, from the standing outcomes of the last sweep:
//     absorber   ab[x]  (the smallest centre whose outcome absorbs x), followed while the absorbers are < i and increasing
//     size       x < i ? size after x's own turn : size at the start of the round
//     adjacency  x < i and x ran ? its new list : its list at the start of the round
// A sweep re-runs the centres on the work list in parallel (one wavefront per centre: lanes = candidates / list entries,
// the search order of the reference is kept with ballots, so the lists come out in the reference's order); then the nodes
// whose state changed wake the centres that may have read them (reverse index of the base lists + the absorber chains).
// No work left <=> every centre's outcome is consistent with the outcomes before it <=> the serial result.

struct FusState {
    // static in a round
    const FePt* P;
    const float4* Pf;          // [2 n] the same points in single precision, 32 B each: (x, y, z, nx) (ny, nz, -, -) - fus_loss_decides
    float inv_res_f;           // (float)(1 / res)
    double res, lambda;
    const int* root0;          // [n] root of a point at the start of the round (the entries of the base lists ARE roots at the start
                               // of the round - k_fus_next_lists resolves them on the way into the arena - so no search reads it)
    const int* s0;             // [n] size of a root
    const int* len0;           // [n] length of its adjacency list (0: absorbed in an earlier round / isolated)
    const long long* off0;     // [n] offset of the list in arena0
    const int* arena0;
    const int* revoff;         // [n + 1] reverse index: owners of base entries whose root0 is x
    const int* revown;
    // standing outcomes
    int* ab;                   // [n] absorber, kNone
    int* ab_prev;              // [n] absorber as of the previous sweep
    int* rec_sz;               // [n] size after the node's own turn (= s0 when it did not run)
    int* rec_ran;              // [n]
    int* rec_absn;             // [n] absorbed nodes, in order, at sa[rec_ptr ..], followed by rec_adjn adjacent nodes
    int* rec_adjn;
    long long* rec_ptr;
    int* sa;                   // list arena of the round (bump allocated; nothing is freed inside a round)
    unsigned long long* sa_top;
    unsigned long long sa_cap;
    // work lists and per-slot outcomes of the running sweep
    const int* W;
    int* slot_of;              // [n] slot of a centre on W (only valid when W[slot_of[x]] == x)
    int* o_sz; int* o_ran; int* o_absn; int* o_adjn; long long* o_ptr; int* o_dirty; long long* o_oldptr; int* o_oldabsn;
    int* wake;                 // [n] 1: already on the next work list
    int* Wnext; int* nWnext;
    int* dflag;                // [n] 1: already on a dirty list
    int* cflag;                // [n] 1: already handled as a node absorbed by a changed node
    int* dtmin;                // [n] a changed node concerns the centres after dtmin only (kNone outside a sweep)
    int* status;               // [0] queue overflow, [1] arena overflow, [2] dirty closure deeper than the levels run,
                               // [3] too many changed nodes: everybody runs again
    int wake_all_above;
    // short work lists: several sweeps are enqueued before the host reads anything back; the number of slots then lives on the
    // device (nW_dev; nullptr: the launch argument counts) and `stop` != 0 makes the remaining launches of the batch return
    const int* nW_dev;
    int* stop;
    int queue_limit;           // <= kFusQueue ($PWICP_FUSION_QUEUE: smaller, to exercise the fallback)
    int slot0;                 // first slot of the part of the work list this launch takes (a sweep in two colours: fusion_device)
    int* changed;              // [16 x 32] centres whose outcome changed in the running sweep, spread over 16 lines (k_fus_run adds,
                               // k_fus_dirty0 reads); nullptr: not counted (batched sweeps)
};

// search queue / visited hash of a wavefront: the common case in LDS small enough for 5 blocks of 4 wavefronts per CU (the kernel
// is bound by the latency of dependent gathers: occupancy is throughput); a centre whose search outgrows it runs again in
// the same sweep on a wavefront with the large configuration
#ifndef PW_FUS_QUEUE_S
#define PW_FUS_QUEUE_S 256
#endif
#ifndef PW_FUS_MIN_WAVES
#define PW_FUS_MIN_WAVES 6            // wavefronts per SIMD asked of the compiler for the sweeps' kernel: 6 is what the LDS of a block allows
                                      // (24 per CU), 80 registers with one spilled instead of 84 and five wavefronts: fusion 38.9 -> 37.5 ms
                                      // per 1 M points (build parameter; 1: the compiler's choice; 7 / 8 with smaller tables: slower)
#endif
constexpr int kFusQueueS = PW_FUS_QUEUE_S, kFusHashS = 2 * PW_FUS_QUEUE_S, kFusQueue = 2048, kFusHash = 4096;
constexpr int kFusArenas = 256;

// batched sweeps (FusState::nW_dev): the number of slots lives on the device, a raised stop flag ends the batch
#define FUS_BATCHED_NW(s, nW)            \
    do {                                 \
        if ((s).nW_dev) {                \
            if (*(s).stop) return;       \
            nW = *(s).nW_dev;            \
        }                                \
    } while (0)
#define WSYNC()                                               \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
    } while (0)

// A wavefront takes a CHUNK of consecutive work-list slots and runs them one after the other: a centre then sees what the
// centres of its own chunk have just decided (their outcomes and claims are kept in LDS) instead of their outcomes of the
// previous sweep - Gauss-Seidel inside the chunk, Jacobi between chunks.  Neighbouring centres are neighbours in the visiting
// order, which is where most dependencies are: the worst round needs ~3x fewer sweeps.  Any mixture of old and new guesses
// converges to the same fixed point; a sweep without changes still certifies it.
#ifndef PW_FUS_CHUNK
#define PW_FUS_CHUNK 16
#endif
constexpr int kFusChunk = PW_FUS_CHUNK, kFusFresh = 8 * PW_FUS_CHUNK;
// wavefronts per block of the sweeps' launch (the wavefronts of a block share nothing: a block only holds its slots until its
// slowest wavefront is done) and the most blocks of one launch (beyond that a wavefront takes several chunks)
#ifndef PW_FUS_WAVES
#define PW_FUS_WAVES 4
#endif
#ifndef PW_FUS_GRID_CAP
#define PW_FUS_GRID_CAP 8192
#endif
constexpr int kFusWaves = PW_FUS_WAVES, kFusGridCap = PW_FUS_GRID_CAP;

struct FusWave {           // per-wavefront scratch (LDS)
    int* keys; int* vals; int* queue;
    int qn, gcount, qcap;
    bool overflow;
    // centres this wavefront has already run in this sweep (chunk), and the nodes they absorbed
    int* cid; int* csz; int* cran; int* cabsn; int* cadjn; long long* cptr;
    int cidv;              // lane t: the t-th centre of the chunk that has run (the copy fus_chunk_index scans: a v_readlane per
                           // entry instead of an LDS read and its wait - the scan runs three or four times per centre)
    int* fkey; int* fval;
    int ndone, nfresh;
};

__device__ __forceinline__ int fus_chunk_index(const FusWave& w, int c) {
    int q = -1;
    const int nd = __builtin_amdgcn_readfirstlane(w.ndone);
    for (int t = 0; t < nd; ++t)
        if (__builtin_amdgcn_readlane(w.cidv, t) == c) q = t;
    return q;
}

// absorber of x as the running sweep knows it: the standing one, unless it belongs to a centre of the chunk that has run
// again (then only the claims that centre has just made count)
__device__ __forceinline__ int fus_absorber_from(const FusState& s, const FusWave& w, int x, int c) {
    if (w.ndone) {
        if (c != kNone && fus_chunk_index(w, c) >= 0) c = kNone;
        if (w.nfresh) {
            int sl = (int)(((unsigned)x * 2654435761u) >> (32 - __builtin_ctz(kFusFresh)));
            for (;;) {
                const int key = w.fkey[sl];
                if (key == -1) break;
                if (key == x) { c = min(c, w.fval[sl]); break; }
                sl = (sl + 1) & (kFusFresh - 1);
            }
        }
    }
    return c;
}
__device__ __forceinline__ int fus_absorber(const FusState& s, const FusWave& w, int x) { return fus_absorber_from(s, w, x, s.ab[x]); }

__device__ __forceinline__ int fus_root_at(const FusState& s, const FusWave& w, int y, int t) {
    int x = y, g = -1;                 // (y is a root of the round's start: base lists are resolved, outcome lists hold roots)
    for (;;) {
        const int c = fus_absorber(s, w, x);
        if (c >= t || c <= g) break;
        g = c; x = c;
    }
    return x;
}

// roots of the entries of a list join the search, in list order, each once
template <int HCAP>
__device__ __forceinline__ void fus_expand(const FusState& s, FusWave& w, const int* __restrict__ lp, int len, int i, int lane) {
    for (int base = 0; base < len && !w.overflow; base += 64) {
        const int e = base + lane;
        const bool valid = e < len;
        int slot = 0;
        int r = 0;
        const int gidx = w.gcount + lane;
        if (valid) {
            r = fus_root_at(s, w, lp[e], i);
            slot = (int)(((unsigned)r * 2654435761u) >> (32 - __builtin_ctz(HCAP)));
            for (;;) {
                const int prev = atomicCAS(&w.keys[slot], -1, r);
                if (prev == -1 || prev == r) break;
                slot = (slot + 1) & (HCAP - 1);
            }
            atomicMin(&w.vals[slot], gidx);
        }
        WSYNC();
        const bool first = valid && w.vals[slot] == gidx;
        const unsigned long long m = __ballot(first);
        const int add = __popcll(m);
        if (w.qn + add > w.qcap) { w.overflow = true; break; }
        if (first) w.queue[w.qn + __popcll(m & ((1ull << lane) - 1ull))] = r;
        w.qn += add;
        w.gcount += 64;
        WSYNC();
    }
}

// Does centre `me` absorb a root j of size sj?  The reference's test is lambda - sj * metric(i, j) > 0 in double (:120-123), and the
// double metric - a square root, a division - is ~90 of the ~410 vector instructions of a run.  The same expression in single
// precision (v_sqrt_f32, a multiplication by 1 / res) is within kFusEps * (1 + distance term) of it: Pf holds the double
// coordinates (exact: they were floats) and normals (relative 2^-24) rounded once; the dot product of two (near-)unit normals
// is then off by < 6 * 2^-24, the distance term by < 9 * 2^-24 relative, the two sums by 2 * 2^-24 each - under 11 * 2^-24 *
// (1 + term) = 6.6e-7 * (1 + term) in all; kFusEps is six times that.  Whenever lambda is farther from the single-precision loss
// than that bound the sign of the double expression is known; the (rare: ~1e-5 of the candidates) others take the double path.
// So the decision is the reference's, bit for bit, and the double points are only gathered for the ambiguous candidates.
constexpr float kFusEps = 4.0e-6f;
__device__ __forceinline__ bool fus_loss_decides(const FusState& s, const float4 m0, const float4 m1, int j, int sj, bool& absorb) {
    const float4 a0 = s.Pf[2 * (size_t)j], a1 = s.Pf[2 * (size_t)j + 1];
    const float dotf = m0.w * a0.w + m1.x * a1.x + m1.y * a1.y;
    const float t1 = m0.x - a0.x, t2 = m0.y - a0.y, t3 = m0.z - a0.z;
    const float term = __builtin_amdgcn_sqrtf(t1 * t1 + t2 * t2 + t3 * t3) * s.inv_res_f * 0.4f;
    const float mf = 1.0f - fabsf(dotf) + term;
    const double lossf = (double)sj * (double)mf;
    const double bound = (double)sj * (double)(kFusEps * (1.0f + term));
    const double imp = s.lambda - lossf;
    absorb = imp > 0.0;
    return fabs(imp) > bound;
}

// The words a run starts from - the centre's standing outcome (rec_*), its size, list and absorber - one per lane (lane 0 ... 10), so
// that they can be requested for the NEXT centre of the chunk while this one runs: the work-list entry and these words were two
// round trips at the head of every run (16 % of it, tools/fus_knockout.sh).  None of them is written inside the kernel.
__device__ __forceinline__ int fus_head_words(const FusState& s, int lane, int i) {
    const int* p = s.rec_ran + i;
    p = lane == 1 ? s.rec_sz + i : p;
    p = lane == 2 ? s.rec_absn + i : p;
    p = lane == 3 ? s.rec_adjn + i : p;
    p = lane == 4 ? reinterpret_cast<const int*>(s.rec_ptr + i) : p;
    p = lane == 5 ? reinterpret_cast<const int*>(s.rec_ptr + i) + 1 : p;
    p = lane == 6 ? s.s0 + i : p;
    p = lane == 7 ? s.len0 + i : p;
    p = lane == 8 ? reinterpret_cast<const int*>(s.off0 + i) : p;
    p = lane == 9 ? reinterpret_cast<const int*>(s.off0 + i) + 1 : p;
    p = lane == 10 ? s.ab + i : p;
    return lane <= 10 ? *p : 0;
}

// QCAP / HCAP: capacity of the search queue / visited hash; WAVES wavefronts per block.  list == nullptr: the work list W in
// chunks; else the slots on `list` (the centres whose search outgrew the small configuration), one at a time.
template <int QCAP, int HCAP, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, (QCAP <= 512 ? PW_FUS_MIN_WAVES : 1)) k_fus_run(FusState s, int nW, int chunk, const int* __restrict__ list,
                                                         const int* __restrict__ n_list, int* __restrict__ ovf, int* __restrict__ n_ovf) {
    __shared__ __attribute__((aligned(16))) int s_keys[WAVES][HCAP];
    __shared__ __attribute__((aligned(16))) int s_vals[WAVES][HCAP];
    __shared__ int s_queue[WAVES][QCAP];
    __shared__ int s_chunk[WAVES][5][kFusChunk];
    __shared__ long long s_cptr[WAVES][kFusChunk];
    __shared__ int s_fresh[WAVES][2][kFusFresh];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    FusWave w;
    w.keys = s_keys[wave]; w.vals = s_vals[wave]; w.queue = s_queue[wave];
    w.cid = s_chunk[wave][0]; w.csz = s_chunk[wave][1]; w.cran = s_chunk[wave][2]; w.cabsn = s_chunk[wave][3]; w.cadjn = s_chunk[wave][4];
    w.cptr = s_cptr[wave];
    w.fkey = s_fresh[wave][0]; w.fval = s_fresh[wave][1];
    w.qcap = min(QCAP, s.queue_limit);
    FUS_BATCHED_NW(s, nW);
    if (list) { nW = *n_list; chunk = 1; }
    const int n_chunks = (nW + chunk - 1) / chunk;
    int n_changed = 0;                                  // (wave-uniform)
    for (int ci = blockIdx.x * WAVES + wave; ci < n_chunks; ci += gridDim.x * WAVES) {
      w.ndone = 0; w.nfresh = 0; w.cidv = -1;
#ifdef PW_FUS_JACOBI_CHUNK          // (diagnostic build: the chunk's centres do not see each other - what the fresh view costs per run)
      bool chunk_live = false;
#else
      bool chunk_live = chunk > 1;
#endif
      if (chunk > 1) {
          for (int t = lane; t < kFusFresh; t += 64) w.fkey[t] = -1;
          WSYNC();
      }
      const int slot_end = min(nW, (ci + 1) * chunk);
#ifndef PW_FUS_NO_PREFETCH
      // the chunk's work-list entries in one load (lane t: the t-th), the head words of its first centre
      const int chunk_n = slot_end - ci * chunk;
      int slot_v = 0, cen_v = 0;
      if (lane < chunk_n) {
          slot_v = list ? list[ci * chunk + lane] : s.slot0 + ci * chunk + lane;
          cen_v = s.W[slot_v];
      }
      int head_next = fus_head_words(s, lane, __builtin_amdgcn_readlane(cen_v, 0));
#endif
      for (int sl_i = ci * chunk; sl_i < slot_end; ++sl_i) {
#ifndef PW_FUS_NO_PREFETCH
        const int t_in = sl_i - ci * chunk;
        const int slot = __builtin_amdgcn_readlane(slot_v, t_in);
        const int i = __builtin_amdgcn_readlane(cen_v, t_in);
        const int head = head_next;
        if (sl_i + 1 < slot_end) head_next = fus_head_words(s, lane, __builtin_amdgcn_readlane(cen_v, t_in + 1));   // (in flight during this run)
        if (lane == 0) { s.slot_of[i] = slot; s.wake[i] = 0; }
        const int old_ran = __builtin_amdgcn_readlane(head, 0), old_sz = __builtin_amdgcn_readlane(head, 1);
        const int old_absn = __builtin_amdgcn_readlane(head, 2), old_adjn = __builtin_amdgcn_readlane(head, 3);
        const long long old_ptr = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(head, 5) << 32) |
                                              (unsigned long long)(unsigned)__builtin_amdgcn_readlane(head, 4));
        const int s0_i = __builtin_amdgcn_readlane(head, 6), len0_i = __builtin_amdgcn_readlane(head, 7);
        const long long off0_i = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(head, 9) << 32) |
                                             (unsigned long long)(unsigned)__builtin_amdgcn_readlane(head, 8));
        const int ab_i = __builtin_amdgcn_readlane(head, 10);
#else
        const int slot = list ? list[sl_i] : s.slot0 + sl_i;
        const int i = s.W[slot];
        if (lane == 0) { s.slot_of[i] = slot; s.wake[i] = 0; }
        const int old_ran = s.rec_ran[i], old_sz = s.rec_sz[i], old_absn = s.rec_absn[i], old_adjn = s.rec_adjn[i];
        const long long old_ptr = s.rec_ptr[i];
        const int s0_i = s.s0[i], len0_i = s.len0[i];
        const long long off0_i = s.off0[i];
        const int ab_i = s.ab[i];
#endif
        int ran = 0, size_i = s0_i, nabs = 0;
        w.qn = 1; w.gcount = 0; w.overflow = false;
#if defined(PW_FUS_KO) && PW_FUS_KO == 1      // (knock-out profiling, tools/fus_knockout.sh: the run ends after its prologue)
        if (old_ran + old_sz + old_absn + old_adjn + (int)old_ptr != -12345) { if (lane == 0) { s.o_sz[slot] = old_sz; s.o_ran[slot] = old_ran; s.o_absn[slot] = old_absn; s.o_adjn[slot] = old_adjn; s.o_ptr[slot] = old_ptr; s.o_dirty[slot] = 0; s.o_oldptr[slot] = old_ptr; s.o_oldabsn[slot] = old_absn; } continue; }
#endif
        if (len0_i != 0 && !(fus_absorber_from(s, w, i, ab_i) < i)) {
            ran = 1;
            {   // (four entries per store)
                int4* k4 = reinterpret_cast<int4*>(w.keys);
                int4* v4 = reinterpret_cast<int4*>(w.vals);
                for (int t = lane; t < HCAP / 4; t += 64) { k4[t] = make_int4(-1, -1, -1, -1); v4[t] = make_int4(INT_MAX, INT_MAX, INT_MAX, INT_MAX); }
            }
            WSYNC();
            if (lane == 0) {
                const int sl = (int)(((unsigned)i * 2654435761u) >> (32 - __builtin_ctz(HCAP)));
                w.keys[sl] = i; w.vals[sl] = -1;
                w.queue[0] = i;
            }
            WSYNC();
            fus_expand<HCAP>(s, w, s.arena0 + off0_i, len0_i, i, lane);
            const float4 me0 = s.Pf[2 * (size_t)i], me1 = s.Pf[2 * (size_t)i + 1];
            int front = 1;
#if defined(PW_FUS_KO) && PW_FUS_KO == 2      // (... after the search has taken in the centre's own list)
            front = w.qn;
#endif
            while (front < w.qn && !w.overflow) {
                const int stop = w.qn;
                for (int base = front; base < stop && !w.overflow; base += 64) {
                    const int idx = base + lane;
                    const bool valid = idx < stop;
                    const int j = valid ? w.queue[idx] : 0;
                    int sj = 0;
                    bool absorb = false, sure = true;
                    if (valid) {
                        if (j < i) {
                            const int q = fus_chunk_index(w, j);
                            sj = q >= 0 ? w.csz[q] : s.rec_sz[j];
                        } else {
                            sj = s.s0[j];
                        }
#ifdef PW_FUS_DOUBLE_ONLY          // (the reference's expression for every candidate: A/B and cross-check build)
                        const double loss = (double)sj * sv_metric(s.P[i], s.P[j], s.res);
                        absorb = s.lambda - loss > 0.0;
#else
                        sure = fus_loss_decides(s, me0, me1, j, sj, absorb);
#endif
                    }
#ifndef PW_FUS_DOUBLE_ONLY
                    if (__ballot(valid && !sure)) {                     // (wave-uniform, rare)
                        if (valid && !sure) {
                            const double loss = (double)sj * sv_metric(s.P[i], s.P[j], s.res);
                            absorb = s.lambda - loss > 0.0;
                        }
                    }
#endif
                    if (valid && absorb) w.queue[idx] = j | (int)0x80000000;
                    unsigned long long A = __ballot(absorb);
                    while (A && !w.overflow) {
                        const int l = __builtin_ctzll(A);
                        A &= A - 1ull;
                        const int jj = __builtin_amdgcn_readlane(j, l);
                        size_i += __builtin_amdgcn_readlane(sj, l);
                        ++nabs;
                        const int q = jj < i ? fus_chunk_index(w, jj) : -1;
                        if (q >= 0) {
                            if (w.cran[q]) fus_expand<HCAP>(s, w, s.sa + w.cptr[q] + w.cabsn[q], w.cadjn[q], i, lane);
                            else fus_expand<HCAP>(s, w, s.arena0 + s.off0[jj], s.len0[jj], i, lane);
                        } else if (jj < i && s.rec_ran[jj]) {
                            fus_expand<HCAP>(s, w, s.sa + s.rec_ptr[jj] + s.rec_absn[jj], s.rec_adjn[jj], i, lane);
                        } else {
                            fus_expand<HCAP>(s, w, s.arena0 + s.off0[jj], s.len0[jj], i, lane);
                        }
                    }
                }
                front = stop;
            }
        }
        if (w.overflow) {
            if (lane == 0) {
                if (ovf && w.qcap == QCAP) { s.o_dirty[slot] = 2; ovf[atomicAdd(n_ovf, 1)] = slot; }     // again, with the large configuration
                else { s.o_dirty[slot] = 2; s.status[0] = 1; }
            }
            continue;
        }
#if defined(PW_FUS_KO) && (PW_FUS_KO == 2 || PW_FUS_KO == 3)      // (... before the outcome is compared and written)
        if (lane == 0) { s.o_sz[slot] = old_sz; s.o_ran[slot] = old_ran; s.o_absn[slot] = old_absn; s.o_adjn[slot] = old_adjn; s.o_ptr[slot] = old_ptr; s.o_dirty[slot] = 0; s.o_oldptr[slot] = old_ptr; s.o_oldabsn[slot] = old_absn; }
        if (w.qn != -12345) continue;
#endif
        // outcome: absorbed nodes then adjacent nodes, each in search order; unchanged lists keep their place in the arena
        const int total = ran ? w.qn - 1 : 0;
        const int nadj = total - nabs;
        bool same = ran == old_ran && size_i == old_sz && nabs == old_absn && nadj == old_adjn;
        long long new_ptr = old_ptr;
        for (int pass = 0; pass < 2; ++pass) {
            long long ptr = old_ptr;
            if (pass == 0 && !same) continue;
            if (pass == 1) {
                if (same) break;
                // (the arena is split into kFusArenas regions with a bump pointer each: same-address atomics serialise)
                const unsigned long long region = (unsigned long long)(blockIdx.x & (kFusArenas - 1));
                const unsigned long long rcap = s.sa_cap / kFusArenas;
                unsigned long long at = 0;
                if (lane == 0) at = atomicAdd(&s.sa_top[region * 16], (unsigned long long)total);
                at = __shfl(at, 0);
                if (at + (unsigned long long)total > rcap) {
                    if (lane == 0) { s.o_dirty[slot] = 2; s.status[1] = 1; }
                    same = true;          // (nothing is written; the host aborts)
                    break;
                }
                ptr = (long long)(region * rcap + at);
                new_ptr = ptr;
            }
            int na = 0, nd = 0;
            bool diff = false;
            for (int base = 1; base <= total; base += 64) {
                const int idx = base + lane;
                const bool valid = idx <= total;
                const int q = valid ? w.queue[idx] : 0;
                const bool isab = valid && q < 0;
                const unsigned long long ma = __ballot(isab), mv = __ballot(valid);
                const unsigned long long lt = (1ull << lane) - 1ull;
                const int node = q & 0x7fffffff;
                const long long at = isab ? (long long)(na + __popcll(ma & lt)) : (long long)(nabs + nd + __popcll((mv & ~ma) & lt));
                if (valid) {
                    if (pass == 0) diff |= s.sa[ptr + at] != node;
                    else s.sa[ptr + at] = node;
                }
                na += __popcll(ma);
                nd += __popcll(mv & ~ma);
            }
            if (pass == 0) {
                if (same && __ballot(diff)) same = false;
            } else if (lane == 0) {
                s.o_ptr[slot] = ptr;
            }
        }
        if (lane == 0) {
            s.o_sz[slot] = size_i; s.o_ran[slot] = ran; s.o_absn[slot] = nabs; s.o_adjn[slot] = nadj;
            if (same) s.o_ptr[slot] = old_ptr;
            s.o_dirty[slot] = same ? 0 : 1;
            s.o_oldptr[slot] = old_ptr; s.o_oldabsn[slot] = old_absn;
        }
        n_changed += same ? 0 : 1;
        // what the later centres of the chunk see of this one
        if (chunk_live && sl_i + 1 < slot_end && w.ndone < kFusChunk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // its lists are in memory before they are read back (measured: free)
            if (lane == 0) {
                const int q = w.ndone;
                w.cid[q] = i; w.csz[q] = size_i; w.cran[q] = ran; w.cabsn[q] = nabs; w.cadjn[q] = nadj; w.cptr[q] = new_ptr;
            }
            if (lane == w.ndone) w.cidv = i;
            if (nabs > 0 && w.nfresh + nabs > kFusFresh / 2) {
                // no room for its claims: the rest of the chunk reads the standing state only (a view that is neither the old
                // nor the new state would not be flagged as a change)
                chunk_live = false;
                w.ndone = 0; w.nfresh = 0;
                WSYNC();
                continue;
            }
            if (nabs > 0) {
                for (int base = 1; base <= total; base += 64) {
                    const int idx = base + lane;
                    if (idx <= total && w.queue[idx] < 0) {
                        const int node = w.queue[idx] & 0x7fffffff;
                        int sl = (int)(((unsigned)node * 2654435761u) >> (32 - __builtin_ctz(kFusFresh)));
                        for (;;) {
                            const int prev = atomicCAS(&w.fkey[sl], -1, node);
                            if (prev == -1) { w.fval[sl] = i; break; }
                            if (prev == node) { atomicMin(&w.fval[sl], i); break; }
                            sl = (sl + 1) & (kFusFresh - 1);
                        }
                    }
                }
                w.nfresh += nabs;
            }
            ++w.ndone;
            WSYNC();
        }
      }
    }
    if (s.changed && lane == 0 && n_changed) atomicAdd(&s.changed[(blockIdx.x & 15) * 32], n_changed);
}

// the claims of the outcomes that changed: first all old ones are withdrawn, then the new ones are made (two launches)
__global__ void k_fus_retract(FusState s, int nW) {
    FUS_BATCHED_NW(s, nW);
    const int slot = s.slot0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= s.slot0 + nW || s.o_dirty[slot] != 1) return;
    const int c = s.W[slot];
    const long long p = s.o_oldptr[slot];
    for (int e = 0, m = s.o_oldabsn[slot]; e < m; ++e) {
        const int j = s.sa[p + e];
        if (s.ab[j] == c) s.ab[j] = kNone;
    }
}
__global__ void k_fus_claim(FusState s, int nW) {
    FUS_BATCHED_NW(s, nW);
    const int slot = s.slot0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= s.slot0 + nW || s.o_dirty[slot] == 2) return;
    const int c = s.W[slot];
    if (s.o_dirty[slot] == 1) {
        s.rec_sz[c] = s.o_sz[slot]; s.rec_ran[c] = s.o_ran[slot]; s.rec_absn[c] = s.o_absn[slot]; s.rec_adjn[c] = s.o_adjn[slot];
        s.rec_ptr[c] = s.o_ptr[slot];
    }
    // the claims of EVERY centre that ran are made again: a standing claim that was hidden behind a smaller one (ab keeps only
    // the smallest) would otherwise be lost when the smaller one is withdrawn while its owner, seeing the node free, stays as it is
    const long long p = s.o_ptr[slot];
    for (int e = 0, m = s.o_absn[slot]; e < m; ++e) atomicMin(&s.ab[s.sa[p + e]], c);
}
// all standing claims (the certificate of a round rebuilds the absorbers from scratch)
__global__ void k_fus_claim_all(FusState s, const int* __restrict__ cen, int nc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc) return;
    const int c = cen[t];
    if (!s.rec_ran[c]) return;
    const long long p = s.rec_ptr[c];
    for (int e = 0, m = s.rec_absn[c]; e < m; ++e) atomicMin(&s.ab[s.sa[p + e]], c);
}
__global__ void k_fus_ab_changed(const int* __restrict__ ab, int* __restrict__ ab_prev, int n, int* __restrict__ changed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (ab[i] != ab_prev[i]) { ab_prev[i] = ab[i]; *changed = 1; }
}

// append to a list with ONE atomic per wavefront (the lanes that are active here and want to; same-address atomics serialise)
__device__ __forceinline__ void wave_append(int* ctr, int* arr, bool want, int v) {
    const unsigned long long m = __ballot(want);
    if (!m) return;
    const int lane = threadIdx.x & 63, leader = __builtin_ctzll(m);
    int base = 0;
    if (lane == leader) base = atomicAdd(ctr, __popcll(m));
    base = __shfl(base, leader);
    if (want) arr[base + __popcll(m & ((1ull << lane) - 1ull))] = v;
}

// A changed node x only matters to the centres AFTER t = dtmin[x]: its own outcome is read by later centres only (t = x),
// a change of its absorber from c to c' is invisible to the centres up to min(c, c') (for them x is a free root either way).
__device__ __forceinline__ void fus_mark_dirty(const FusState& s, bool on, int x, int tmin, int* dq, int* ndq) {
    bool fresh = false;
    if (on) {
        atomicMin(&s.dtmin[x], tmin);
        fresh = atomicExch(&s.dflag[x], 1) == 0;
    }
    wave_append(ndq, dq, fresh, x);
}

// level 0 of the dirty list: centres whose outcome changed, nodes whose absorber changed
__global__ void k_fus_dirty0(FusState s, int nW, int* dq, int* ndq) {
    FUS_BATCHED_NW(s, nW);
    // More centres changed their outcome in this sweep than the wake-up looks at changed nodes (every such centre is one): the
    // list would be thrown away - k_fus_wake asks for everybody to run again - so it is not made.  The count is final (k_fus_run
    // has ended); the host then takes the sweep's absorbers as the previous ones wholesale.
    if (s.changed) {
        int nch = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) nch += s.changed[r * 32];
        if (nch > s.wake_all_above) {
            if (blockIdx.x == 0 && threadIdx.x == 0) *ndq = nch;
            return;
        }
    }
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = slot < nW && s.o_dirty[slot] != 2;
    const bool on = live && s.o_dirty[slot] == 1;
    const int c = live ? s.W[slot] : 0;
    fus_mark_dirty(s, on, c, c, dq, ndq);
    if (!live) return;
    for (int pass = on ? 0 : 1; pass < 2; ++pass) {
        const long long p = pass ? s.o_ptr[slot] : s.o_oldptr[slot];
        const int m = pass ? s.o_absn[slot] : s.o_oldabsn[slot];
        for (int e = 0; e < m; ++e) {
            const int j = s.sa[p + e];
            const int a0 = s.ab_prev[j], a1 = s.ab[j];
            fus_mark_dirty(s, a0 != a1, j, min(a0, a1), dq, ndq);
        }
    }
}

// `on` lanes wake their owner and whoever absorbed it, transitively, along the new absorbers; where the old absorber of
// a node on the way differs, the old chain is followed from there as well (rare)
__device__ __forceinline__ void fus_wake_chain(const FusState& s, bool on, int owner, int tmin) {
    int x = owner, g = -1;
    bool go = on;
    while (__ballot(go)) {
        const bool fresh = go && x > tmin && s.wake[x] == 0 && atomicExch(&s.wake[x], 1) == 0;
        wave_append(s.nWnext, s.Wnext, fresh, x);
        int y = kNone, gy = -1;
        if (go) {
            const int nx = s.ab[x], ox = s.ab_prev[x];
            if (ox != nx && ox != kNone && ox > g) { y = ox; gy = ox; }
            if (nx == kNone || nx <= g) go = false;
            else { g = nx; x = nx; }
        }
        while (__ballot(y != kNone)) {                 // old chain from here on
            const bool on2 = y != kNone;
            const bool fresh2 = on2 && y > tmin && s.wake[y] == 0 && atomicExch(&s.wake[y], 1) == 0;
            wave_append(s.nWnext, s.Wnext, fresh2, y);
            if (on2) {
                const int ny = s.ab_prev[y];
                if (ny == kNone || ny <= gy) y = kNone;
                else { gy = ny; y = ny; }
            }
        }
    }
}

// One wavefront per changed node x: it wakes the owners of the base entries that lead to x (and whoever absorbed them,
// transitively).  The nodes x has absorbed (old and new outcome) are read THROUGH x, so they are handled like x - once per
// sweep (dflag), for all centres (bound -1: the node may hang below several changed nodes) - from a small stack of the
// wavefront (wider than the stack: status[2], everybody runs again).  They go on dq2 only to have their flag reset.
constexpr int kWakeStack = 256;
__global__ void __launch_bounds__(256) k_fus_wake(FusState s, const int* dq, const int* ndq, int* dq2, int* ndq2) {
    __shared__ int s_node[4][kWakeStack];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    if (s.nW_dev && *s.stop) return;
    const int n_dirty = *ndq;
    // Many changed nodes: finding who read them costs more than running every centre again (a run is ~2 ns of device
    // time, a changed node ~10 ns of walks) - the host puts all centres on the next work list (status[3]).
    if (n_dirty > s.wake_all_above) {
        if (blockIdx.x == 0 && threadIdx.x == 0) s.status[3] = 1;
        return;
    }
    int* node = s_node[wv];
    for (int t = wave; t < n_dirty; t += nwaves) {
        int top = 1;
        const int root = dq[t];
        if (lane == 0) node[0] = root;
        WSYNC();
        while (top > 0) {
            --top;
            const int x = node[top];
            const int tmin = top == 0 && x == root ? s.dtmin[root] : -1;
            WSYNC();
            const int e0 = s.revoff[x] - 1, m0 = s.revoff[x + 1];      // entry e0 stands for x itself
            for (int base = e0; base < m0; base += 64) {
                const int e = base + lane;
                const bool on = e < m0;
                fus_wake_chain(s, on, on ? (e == e0 ? x : s.revown[e]) : 0, tmin);
            }
            for (int pass = 0; pass < 2; ++pass) {
                long long p; int m;
                if (pass == 0) { p = s.rec_ptr[x]; m = s.rec_absn[x]; }
                else {
                    const int slot = s.slot_of[x];
                    if (slot < 0 || s.W[slot] != x || s.o_dirty[slot] != 1) break;     // (slot_of is only meaningful for this sweep's W)
                    p = s.o_oldptr[slot]; m = s.o_oldabsn[slot];
                }
                for (int base = 0; base < m; base += 64) {
                    const bool on = base + lane < m;
                    const int ch = on ? s.sa[p + base + lane] : 0;
                    const bool fresh = on && atomicExch(&s.cflag[ch], 1) == 0;
                    wave_append(ndq2, dq2, fresh, ch);
                    const unsigned long long mk = __ballot(fresh);
                    if (top + __popcll(mk) > kWakeStack) {
                        if (lane == 0) s.status[2] = 1;
                    } else {
                        if (fresh) node[top + __popcll(mk & ((1ull << lane) - 1ull))] = ch;
                        top += __popcll(mk);
                    }
                    WSYNC();
                }
            }
        }
    }
}

__global__ void k_fus_sweep_end(FusState s, const int* dq, const int* ndq, const int* dq2, const int* ndq2) {
    if (s.nW_dev && *s.stop) return;
    if (s.changed) {                                   // (k_fus_dirty0 made no list: nothing was flagged, *ndq is only a count)
        int nch = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) nch += s.changed[r * 32];
        if (nch > s.wake_all_above) return;
    }
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < *ndq; t += gridDim.x * blockDim.x) {
        const int x = dq[t];
        s.dflag[x] = 0;
        s.dtmin[x] = kNone;
        s.ab_prev[x] = s.ab[x];
    }
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < *ndq2; t += gridDim.x * blockDim.x) s.cflag[dq2[t]] = 0;
}

// Batched sweeps (<= 1024 slots): what k_fus_retract, k_fus_claim and k_fus_dirty0 do, as the phases of ONE block - between
// dependent launches the device idles ~5 us, which is what a sweep over a few hundred centres is made of.
// counters of a batch of sweeps: all zero but the number of slots (a kernel argument instead of a 64-byte copy from pageable host memory)
__global__ void k_fus_batch_init(int* __restrict__ ctr, int nW) {
    if (threadIdx.x < 16) ctr[threadIdx.x] = threadIdx.x == 4 ? nW : 0;
}
__global__ void __launch_bounds__(1024) k_fus_post(FusState s, int* dq, int* ndq) {
    if (*s.stop) return;
    const int nW = *s.nW_dev;
    const int slot = threadIdx.x;
    const bool live = slot < nW && s.o_dirty[slot] != 2;
    const bool on = live && s.o_dirty[slot] == 1;
    const int c = live ? s.W[slot] : 0;
    if (on) {                                          // withdraw
        const long long p = s.o_oldptr[slot];
        for (int e = 0, m = s.o_oldabsn[slot]; e < m; ++e) atomicCAS(&s.ab[s.sa[p + e]], c, kNone);
    }
    __syncthreads();
    if (on) {
        s.rec_sz[c] = s.o_sz[slot]; s.rec_ran[c] = s.o_ran[slot]; s.rec_absn[c] = s.o_absn[slot]; s.rec_adjn[c] = s.o_adjn[slot];
        s.rec_ptr[c] = s.o_ptr[slot];
    }
    if (live) {                                        // claim (every centre that ran)
        const long long p = s.o_ptr[slot];
        for (int e = 0, m = s.o_absn[slot]; e < m; ++e) atomicMin(&s.ab[s.sa[p + e]], c);
    }
    __syncthreads();
    fus_mark_dirty(s, on, c, c, dq, ndq);              // changed nodes
    if (!live) return;
    for (int pass = on ? 0 : 1; pass < 2; ++pass) {
        const long long p = pass ? s.o_ptr[slot] : s.o_oldptr[slot];
        const int m = pass ? s.o_absn[slot] : s.o_oldabsn[slot];
        for (int e = 0; e < m; ++e) {
            const int j = s.sa[p + e];
            const int a0 = s.ab_prev[j], a1 = __hip_atomic_load(&s.ab[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fus_mark_dirty(s, a0 != a1, j, min(a0, a1), dq, ndq);
        }
    }
}

// Batched sweeps only: the hand-over to the next sweep that the host does otherwise.  ctr: [0] nWnext, [1] ndq, [2] ndq2,
// [3] overflow slots, [4] nW, [5] stop, [6] sweeps executed, [7] runs; [8..11] status.
__global__ void __launch_bounds__(1024) k_fus_advance(FusState s, int* __restrict__ W, const int* __restrict__ Wnext, int* __restrict__ ctr, int cap,
                                                      const int* __restrict__ dq, const int* __restrict__ dq2) {
    __shared__ int s_next;
    if (ctr[5]) return;
    // (k_fus_sweep_end: flags of the changed nodes, absorbers of this sweep become the previous ones)
    for (int t = threadIdx.x, m = ctr[1]; t < m; t += blockDim.x) {
        const int x = dq[t];
        s.dflag[x] = 0;
        s.dtmin[x] = kNone;
        s.ab_prev[x] = __hip_atomic_load(&s.ab[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int t = threadIdx.x, m = ctr[2]; t < m; t += blockDim.x) s.cflag[dq2[t]] = 0;
    if (threadIdx.x == 0) s_next = ctr[0];
    __syncthreads();
    const int next = s_next, now = ctr[4];
    if (now == 0) return;                                                       // nothing was left: idle launch of the batch
    const bool halt = ctr[3] || ctr[8] || ctr[9] || ctr[10] || ctr[11] || next > cap;      // the host decides how to go on
                                                                                             // ([3]: a search outgrew the small queue)
    if (!halt)
        for (int t = threadIdx.x; t < next; t += blockDim.x) W[t] = Wnext[t];
    __syncthreads();
    if (threadIdx.x == 0) {
        ctr[6] += 1;
        ctr[7] += now;
        if (halt) { ctr[5] = 1; return; }
        ctr[4] = next;
        ctr[0] = 0; ctr[1] = 0; ctr[2] = 0; ctr[3] = 0;
    }
}

// ---- set-up and hand-over between rounds ----------------------------------------------------------------------------
// smallest metric to a neighbour (:91-102); lambda0 = its median
// key of a centre's spatial tile (two axes a, b of the cloud's box, tiles of edge 1 / inv): the full sweeps of a round take the
// centres tile by tile - a wavefront's chunk of consecutive slots is then a PATCH of the surface (its centres still in ascending
// order: the sort is stable), and what a centre has just decided reaches its neighbours across the scan lines in the same sweep
__global__ void k_fus_tile_keys(const FePt* __restrict__ P, const int* __restrict__ cen, int nc, int a, int b, double mna, double mnb,
                                double inv, unsigned ntx, unsigned ntiles, int ncol, unsigned* __restrict__ keys) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc) return;
    const FePt& p = P[cen[t]];
    const double ua = a == 0 ? p.x : (a == 1 ? p.y : p.z), ub = b == 0 ? p.x : (b == 1 ? p.y : p.z);
    const unsigned tx = (unsigned)fmin(fmax((ua - mna) * inv, 0.0), (double)(ntx - 1));
    const unsigned ty = (unsigned)fmax((ub - mnb) * inv, 0.0);
    // (colours: ncol = 2 the tiles of a checkerboard, 4 the 2 x 2 pattern - all tiles of colour 0 first, then colour 1, ...)
    const unsigned colour = ncol == 4 ? (tx & 1u) + 2u * (ty & 1u) : (ncol == 2 ? (tx + ty) & 1u : 0u);
    keys[t] = colour * ntiles + ty * ntx + tx;
}
// number of keys below `limit` in a sorted array (= the centres of the first colour)
__global__ void k_fus_count_below(const unsigned* __restrict__ keys, int n, unsigned limit, int* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && keys[t] < limit && (t + 1 == n || keys[t + 1] >= limit)) *out = t + 1;
}
// (... and the entries per node of the first round's reverse index counted on the way, rev_count: k_fus_reverse<0>'s job)
__global__ void k_fus_min_metric(const FePt* __restrict__ P, const int* __restrict__ nb, int k, int n, double res, double* __restrict__ out,
                                 int* __restrict__ rev_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int* row = nb + (size_t)i * k;
    const FePt me = P[i];
    double d = DBL_MAX;
    for (int e0 = 0; e0 < k; e0 += 8) {                  // (eight neighbours in flight)
        int j[8];
        FePt pj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) j[u] = (e0 + u < k) ? row[e0 + u] : i;
#pragma unroll
        for (int u = 0; u < 8; ++u) pj[u] = P[j[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (rev_count && e0 + u < k) atomicAdd(&rev_count[j[u]], 1);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j[u] != i) d = fmin(d, sv_metric(me, pj[u], res));
    }
    out[i] = d;
}
// the points once more in single precision (fus_loss_decides)
__global__ void k_fus_pack_single(const FePt* __restrict__ P, int n, float4* __restrict__ Pf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FePt p = P[i];
    Pf[2 * (size_t)i] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.nx);
    Pf[2 * (size_t)i + 1] = make_float4((float)p.ny, (float)p.nz, 0.f, 0.f);
}
__global__ void k_fus_first_round(int n, int k, int* root0, int* s0, int* len0, long long* off0, int* cen) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    root0[i] = i; s0[i] = 1; len0[i] = k; off0[i] = (long long)i * k; cen[i] = i;
}
__global__ void k_fus_reset(int n, const int* __restrict__ s0, int* ab, int* ab_prev, int* rec_sz, int* rec_ran, int* rec_absn, int* rec_adjn,
                            long long* rec_ptr, int* wake, int* dflag, int* cflag, int* dtmin, int* slot_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ab[i] = kNone; ab_prev[i] = kNone; rec_sz[i] = s0[i]; rec_ran[i] = 0; rec_absn[i] = 0; rec_adjn[i] = 0; rec_ptr[i] = 0;
    wake[i] = 0; dflag[i] = 0; cflag[i] = 0; dtmin[i] = kNone; slot_of[i] = -1;
}
// reverse index of the base lists: MODE 0 counts, MODE 1 scatters (cursor = running offsets)
template <int MODE>
__global__ void k_fus_reverse(const int* __restrict__ cen, int nc, const int* __restrict__ root0, const int* __restrict__ len0,
                              const long long* __restrict__ off0, const int* __restrict__ arena0, int* __restrict__ cnt_or_cursor,
                              int* __restrict__ revown) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc) return;
    const int c = cen[t];
    const int* a = arena0 + off0[c];
    // (eight entries at a time - entries, their roots, the counters: three round trips per eight entries instead of per entry;
    // the order inside a root's range is the atomics' order of arrival either way)
    const int m = len0[c];
    for (int e0 = 0; e0 < m; e0 += 8) {
        int x0[8], at[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x0[u] = (e0 + u < m) ? a[e0 + u] : 0;     // (entries are roots: k_fus_next_lists)
#pragma unroll
        for (int u = 0; u < 8; ++u) at[u] = (e0 + u < m) ? atomicAdd(&cnt_or_cursor[x0[u]], 1) : 0;
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u < m) revown[at[u]] = c;
        }
    }
}
__global__ void k_fus_total_absorbed(const int* __restrict__ cen, int nc, const int* __restrict__ rec_absn, int* __restrict__ per_centre,
                                     unsigned long long* __restrict__ total) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc) return;
    const int a = rec_absn[cen[t]];
    per_centre[t] = a;
    if (a) atomicAdd(total, (unsigned long long)a);
}
// end of a round: every point follows the absorbers of the round (valid: see k_fus_final for the round that stops early)
__global__ void k_fus_new_roots(int n, int* __restrict__ root0, const int* __restrict__ ab) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int x = root0[p], g = -1;
    for (;;) {
        const int c = ab[x];
        if (c == kNone || c <= g) break;
        g = c; x = c;
    }
    root0[p] = x;
}
// per centre (by index on cen): still a root? length of its list in the next round
__global__ void k_fus_next_sizes(const int* __restrict__ cen, int nc, const int* __restrict__ ab, const int* __restrict__ rec_ran,
                                 const int* __restrict__ rec_adjn, const int* __restrict__ len0, int* __restrict__ alive, int* __restrict__ newlen) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc) return;
    const int c = cen[t];
    const bool root = ab[c] == kNone;
    alive[t] = root ? 1 : 0;
    newlen[t] = !root ? 0 : (rec_ran[c] ? rec_adjn[c] : len0[c]);
}
// one wavefront per centre: its list moves into the next round's arena; sizes / offsets of the next round
__global__ void k_fus_next_lists(const int* __restrict__ cen, int nc, const int* __restrict__ alive_scan, const int* __restrict__ len_scan,
                                 const int* __restrict__ ab, const int* __restrict__ rec_ran, const int* __restrict__ rec_sz,
                                 const int* __restrict__ rec_absn, const int* __restrict__ rec_adjn, const long long* __restrict__ rec_ptr,
                                 const int* __restrict__ sa, const int* __restrict__ arena_old, const long long* __restrict__ off_old,
                                 const int* __restrict__ len_old, int* __restrict__ arena_new, long long* __restrict__ off_new,
                                 int* __restrict__ len_new, int* __restrict__ s0, int* __restrict__ cen_new,
                                 const int* __restrict__ root_new, int* __restrict__ rev_count) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (t >= nc) return;
    const int c = cen[t];
    if (ab[c] != kNone) {
        if (lane == 0) { len_new[c] = 0; off_new[c] = 0; }
        return;
    }
    const int m = len_scan[t + 1] - len_scan[t];
    const int* src = rec_ran[c] ? sa + rec_ptr[c] + rec_absn[c] : arena_old + off_old[c];
    int* dst = arena_new + len_scan[t];
    // every entry as the root it has at the start of the next round (k_fus_new_roots has run): the searches and the reverse index
    // of that round then read the list and nothing else - one dependent gather less per entry, in every run of every sweep
    // ... and counted for the next round's reverse index on the way (what k_fus_reverse<0> would read the lists again for)
    for (int e = lane; e < m; e += 64) {
        const int r = root_new[src[e]];
        dst[e] = r;
        atomicAdd(&rev_count[r], 1);
    }
    if (lane == 0) {
        len_new[c] = m; off_new[c] = len_scan[t];
        s0[c] = rec_sz[c];
        cen_new[alive_scan[t]] = c;
    }
}
// the round that reaches the target count stops inside centre `stop_c` after `stop_n` absorptions (:139-141): only the
// claims made before that moment hold
__global__ void k_fus_cut_positions(const int* __restrict__ sa, long long ptr, int m, int* __restrict__ cut) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < m) cut[sa[ptr + e]] = e;
}
__global__ void k_fus_final(int n, const int* __restrict__ root0, const int* __restrict__ ab, int stop_c, int stop_n,
                            const int* __restrict__ cut, int* __restrict__ lab) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int x = root0[p], g = -1;
    for (;;) {
        const int c = ab[x];
        if (c == kNone || c <= g) break;
        if (!(c < stop_c || (c == stop_c && cut[x] < stop_n))) break;
        g = c; x = c;
    }
    lab[p] = x;
}
__global__ void k_fus_is_root(const int* __restrict__ cen, int nc, const int* __restrict__ lab, int* __restrict__ flag) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nc) flag[t] = lab[cen[t]] == cen[t] ? 1 : 0;
}
__global__ void k_fus_compact(const int* __restrict__ cen, int nc, const int* __restrict__ scan, int* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nc && scan[t + 1] != scan[t]) out[scan[t]] = cen[t];
}

// ================================================================================================================
// PCA normals (pca_estimate_normals.h:42-108): the neighbourhood scatter here, the closed-form eigen step on the host
// ================================================================================================================
// Mean and second moments about the mean of the k neighbours of every point, accumulated in double in neighbour order
// exactly as the reference does (unit weights; sums divided by the neighbour count).  The smallest eigenvector is taken
// on the host (pwhost::fe_normals_from_scatter): it needs pow / acos / cos, whose last bit differs between libm and the
// device library, and a different bit there can move a label.
__global__ void k_fe_scatter(const float4* __restrict__ cloud, const int* __restrict__ nb, int k, int i0, int n, double* __restrict__ S6) {
    const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;       // (points i0 .. n - 1 of the cloud: the sums come down in pieces)
    if (i >= n) return;
    const int* row = nb + (size_t)i * k;
    double m0 = 0, m1 = 0, m2 = 0, count = 0;
    for (int e0 = 0; e0 < k; e0 += 8) {                  // (eight neighbours in flight; the sums in neighbour order)
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u < k) v[u] = cloud[row[e0 + u]];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u < k) {
                m0 += (double)v[u].x; m1 += (double)v[u].y; m2 += (double)v[u].z;
                count += 1.0;
            }
    }
    const double to_mean = 1.0 / count;
    m0 *= to_mean; m1 *= to_mean; m2 *= to_mean;
    double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0, weight = 0;
    for (int e0 = 0; e0 < k; e0 += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u < k) v[u] = cloud[row[e0 + u]];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u < k) {
                const double d0 = (double)v[u].x - m0, d1 = (double)v[u].y - m1, d2 = (double)v[u].z - m2;
                xx += d0 * d0; xy += d0 * d1; xz += d0 * d2;
                yy += d1 * d1; yz += d1 * d2; zz += d2 * d2;
                weight += 1.0;
            }
    }
    const double scale = 1.0 / weight;
    double* o = S6 + (size_t)i * 6;
    o[0] = xx * scale; o[1] = xy * scale; o[2] = xz * scale; o[3] = yy * scale; o[4] = yz * scale; o[5] = zz * scale;
}

// number of occupied cells of edge `resolution` (grid_sample.h:30-75): distinct cell keys through an open-addressing table
// (from the float cloud: its coordinates as doubles are FePt's, k_fe_assemble - the count does not wait for the normals)
__global__ void k_fe_count_cells(const float4* __restrict__ cloud, int n, double mn0, double mn1, double mn2, double resolution, int s1, int s2,
                                 int s3, unsigned long long* __restrict__ table, unsigned long long mask, int* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool fresh = false;
    if (i < n) {
        const float4 v = cloud[i];
        int x = (int)(((double)v.x - mn0) / resolution), y = (int)(((double)v.y - mn1) / resolution), z = (int)(((double)v.z - mn2) / resolution);
        x = min(max(x, 0), s1 - 1); y = min(max(y, 0), s2 - 1); z = min(max(z, 0), s3 - 1);
        const unsigned long long key = ((unsigned long long)(unsigned)x << 42) ^ ((unsigned long long)(unsigned)y << 21) ^ (unsigned long long)(unsigned)z;
        unsigned long long h = (key * 0x9E3779B97F4A7C15ull) >> 20 & mask;
        for (;;) {
            const unsigned long long prev = atomicCAS(&table[h], ~0ull, key);
            if (prev == ~0ull) { fresh = true; break; }
            if (prev == key) break;
            h = (h + 1) & mask;
        }
    }
    const unsigned long long m = __ballot(fresh);
    if (m && (threadIdx.x & 63) == __builtin_ctzll(m)) atomicAdd(count, __popcll(m));
}

// Device (and pinned host) buffers of the front end, kept by the context between calls and only ever grown: a front end is
// ~50 buffers; allocating and freeing them per cloud costs milliseconds, and hipFree waits for the whole device - which
// stalls the front ends of other clouds running on other streams (host/registration.cpp: AuxContexts).
struct FeWorkspace {
    // pipeline
    DevBuf<float4> pts;
    Grid grid;
    DevBuf<int> d_nb, d_lab, lab0, d_roots, d_map, cell_cnt;
    DevBuf<double> dS, dN;
    DevBuf<FePt> dP;
    DevBuf<float4> Pf;                  // the points in single precision (fusion_device)
    DevBuf<unsigned long long> table;
    double* hS = nullptr;               // pinned: scatter down, normals up
    double* hN = nullptr;
    size_t h_n = 0;
    int* h_ctr = nullptr;               // pinned: counters read back after every sweep
    unsigned* mail_h = nullptr;         // pinned, mapped, host-coherent: [0..16) words of a small read-back, [16] its sequence number
    unsigned* mail_d = nullptr;         // the same memory as the device sees it
    unsigned mail_seq = 0;
    // refinement
    DevBuf<double> dis, nd;
    DevBuf<unsigned long long> key;
    DevBuf<int> pos, cnt, La, Lb, nla, nlb, flag, tmp;
    // fusion
    DevBuf<double> dmin;
    DevBuf<unsigned long long> sel_state;
    DevBuf<unsigned> sel_hist;
    DevBuf<int> root0, s0, lenA, lenB, cenA, cenB, revoff, revown, cursor, ab, ab_prev, rec_sz, rec_ran, rec_absn, rec_adjn, slot_of, wake,
        dflag, cflag, dtmin, Wa, Wb, dq, dq2, ovf, o_sz, o_ran, o_absn, o_adjn, o_dirty, o_oldabsn, alive, newlen, cut, arenaA, arenaB, sa, ctr;
    DevBuf<long long> offA, offB, rec_ptr, o_ptr, o_oldptr;
    DevBuf<unsigned long long> big;     // [0] absorbed in the round, [16 * (1 + r)] bump pointer of arena region r
    DevBuf<int> cen_t;                  // the centres of a round in TILE order (fusion_device: full sweeps)
    DevBuf<unsigned> tkey, tkey2;
    DevBuf<unsigned char> tsort;
    double bb_mn[3] = {0, 0, 0}, bb_mx[3] = {0, 0, 0};      // bounding box of the cloud (set by the driver of the pipeline)
    bool bb_set = false;
    static constexpr int kPieces = 8;
    hipEvent_t ev_down[kPieces] = {};   // piece c of the scatter sums is in host memory (the stream goes on with work that needs no normals)
    bool rev0_counted = false;          // ws.revoff holds the entries per point of the graph's reverse index (counted by the k-NN launch)
    bool rev0_ready = false;            // the first round's reverse index (and k_fus_first_round's arrays) stand for a cloud of
    int rev0_n = 0, rev0_k = 0;         // rev0_n points, rev0_k neighbours: fusion_prepare_first_round ran ahead of the normals
    int sa_factor = 3;                  // list arena = sa_factor * n * k entries (doubled, once, when a round overflows it)
    bool sa_overflow = false;           // set by fusion_device when it gave up because of the arena
    FeWorkspace() = default;
    FeWorkspace(const FeWorkspace&) = delete;
    FeWorkspace& operator=(const FeWorkspace&) = delete;
    ~FeWorkspace() {
        if (hS) (void)hipHostFree(hS);
        if (hN) (void)hipHostFree(hN);
        if (h_ctr) (void)hipHostFree(h_ctr);
        if (mail_h) (void)hipHostFree(mail_h);
        for (hipEvent_t e : ev_down)
            if (e) (void)hipEventDestroy(e);
    }
    hipError_t host_reserve(size_t n) {
        if (!h_ctr) {
            const hipError_t e = hipHostMalloc((void**)&h_ctr, sizeof(int) * 16, hipHostMallocDefault);
            if (e != hipSuccess) return e;
        }
        if (!mail_h) {
            hipError_t e = hipHostMalloc((void**)&mail_h, sizeof(unsigned) * 32, hipHostMallocMapped | hipHostMallocCoherent);
            if (e != hipSuccess) return e;
            memset(mail_h, 0, sizeof(unsigned) * 32);
            e = hipHostGetDevicePointer((void**)&mail_d, mail_h, 0);
            if (e != hipSuccess) return e;
        }
        if (n <= h_n) return hipSuccess;
        // (with headroom: the clouds of a series differ by a few thousand points after the outlier removal, and a pinned
        // re-allocation of 72 bytes per point is 10 - 25 ms on the stream's host thread - it hit every other cloud of a series)
        n = (n + n / 8 + 65535) & ~(size_t)65535;
        if (hS) (void)hipHostFree(hS);
        if (hN) (void)hipHostFree(hN);
        hS = hN = nullptr;
        h_n = 0;
        hipError_t e = hipHostMalloc((void**)&hS, sizeof(double) * 6 * n, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        e = hipHostMalloc((void**)&hN, sizeof(double) * 3 * n, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        h_n = n;
        return hipSuccess;
    }
};

FeWorkspace* workspace_of(pwicp_context* ctx) {
    if (!ctx->scratch) ctx->scratch = std::shared_ptr<void>(new FeWorkspace, [](void* p) { delete static_cast<FeWorkspace*>(p); });
    return static_cast<FeWorkspace*>(ctx->scratch.get());
}

// Small device -> host read-backs (a counter, a flag, the 16 counters of a sweep) - a front end does ~300 of them, each a
// copy into host memory plus a hipStreamSynchronize (the host thread sleeps and is woken by an interrupt: 30-60 us).  They go
// through a mailbox in pinned host-coherent memory instead, as the registration loop's hand-overs do (common.h: mail_store):
// a one-wave launch behind the producers copies the words and then a sequence number, the host spins on that word.
// PWICP_FE_MAILBOX=0: copy + synchronise.
// (clear / n_clear: words zeroed once the source words are read - the counters of a fusion sweep are re-armed for the next one by the
// launch that reads them, not by a hipMemsetAsync of its own in front of every sweep)
__global__ void __launch_bounds__(64) k_fe_mail(const unsigned* src, int n, unsigned* __restrict__ dst, unsigned seq, unsigned* clear, int n_clear) {
    const int t = threadIdx.x;
    const unsigned word = t < n ? src[t] : 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int u = t; u < n_clear; u += 64) clear[u] = 0u;
    if (t < n) mail_store(&dst[t], word);
    mail_drain();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (t == 0) mail_publish(&dst[16], seq);
}

// two words from two places in one read-back (the counts a round of the fusion hands to the next one)
__global__ void __launch_bounds__(64) k_fe_gather2(const unsigned* __restrict__ a, const unsigned* __restrict__ b, unsigned* __restrict__ out) {
    if (threadIdx.x == 0) out[0] = *a;
    if (threadIdx.x == 1) out[1] = *b;
}

int fe_read_words(pwicp_context* ctx, FeWorkspace& ws, const void* d_src, int n_words, void* h_out, void* d_clear = nullptr, int n_clear = 0) {
    static const bool use_mail = !(getenv("PWICP_FE_MAILBOX") && atoi(getenv("PWICP_FE_MAILBOX")) == 0);
    if (!use_mail || !ws.mail_h || n_words > 16) {
        HIPCHK(ctx, hipMemcpyAsync(h_out, d_src, sizeof(unsigned) * (size_t)n_words, hipMemcpyDeviceToHost, ctx->stream));
        if (n_clear > 0) HIPCHK(ctx, hipMemsetAsync(d_clear, 0, sizeof(unsigned) * (size_t)n_clear, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return PWICP_OK;
    }
    const unsigned seq = ++ws.mail_seq;
    hipLaunchKernelGGL(k_fe_mail, dim3(1), dim3(64), 0, ctx->stream, (const unsigned*)d_src, n_words, ws.mail_d, seq, (unsigned*)d_clear, n_clear);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(&ws.mail_h[16], __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0x3ff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (__atomic_load_n(&ws.mail_h[16], __ATOMIC_ACQUIRE) != seq) {
                ctx->set_err("pwicp front end: device mailbox never signalled");
                return PWICP_E_INTERNAL;
            }
            break;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    memcpy(h_out, ws.mail_h, sizeof(unsigned) * (size_t)n_words);
    return PWICP_OK;
}

// ---- k-th smallest of n doubles (exact; lambda0 = the median of the smallest neighbour metric, :98-102) ---------------------
// MSB-first radix select on the order-preserving 64-bit image of a double, 8 passes of 8 bits: a histogram pass over the
// values that match the digits found so far, then a one-block pick.  (std::nth_element over 1 M doubles costs 6 ms on the
// host plus the 8 MB download; this is ~0.2 ms.)
__device__ __forceinline__ unsigned long long sel_key(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
// state: [0] prefix, [1] mask of the digits already fixed, [2] rank remaining
__global__ void k_sel_hist(const double* __restrict__ v, int n, int shift, const unsigned long long* __restrict__ state,
                           unsigned* __restrict__ hist) {
    __shared__ unsigned s_h[256];
    s_h[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned long long prefix = state[0], mask = state[1];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long key = sel_key(v[i]);
        if ((key & mask) == prefix) atomicAdd(&s_h[(unsigned)(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    const unsigned c = s_h[threadIdx.x];
    if (c) atomicAdd(&hist[threadIdx.x], c);
}
__global__ void k_sel_pick(unsigned* __restrict__ hist, unsigned long long* __restrict__ state, int shift) {
    __shared__ unsigned s_h[256];
    s_h[threadIdx.x] = hist[threadIdx.x];
    hist[threadIdx.x] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long k = state[2], run = 0;
        int digit = 255;
        for (int b = 0; b < 256; ++b) {
            if (k < run + s_h[b]) { digit = b; break; }
            run += s_h[b];
        }
        state[0] |= (unsigned long long)digit << shift;
        state[1] |= 255ull << shift;
        state[2] = k - run;
    }
}
__global__ void k_sel_result(const unsigned long long* __restrict__ state, double* __restrict__ out) {
    const unsigned long long key = state[0];
    const unsigned long long u = (key >> 63) ? (key & 0x7fffffffffffffffull) : ~key;
    *out = __longlong_as_double((long long)u);
}

// EXPERIMENT ($PWICP_NORMALS=device): the eigen step with the device library's pow / acos / cos / sqrt.  Kept to document why
// the default takes it on the host: the result differs from libm's in the last bit of a fraction of the normals, and such a
// bit can move a label.
__global__ void k_fe_eigen_device(const double* __restrict__ S6, int n, double* __restrict__ normals3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* c = S6 + (size_t)i * 6;
    const double xx = c[0], xy = c[1], xz = c[2], yy = c[3], yz = c[4], zz = c[5];
    const double shift = (xx + yy + zz) / 3.0;
    const double bx = xx - shift, by = yy - shift, bz = zz - shift;
    const double p = sqrt((bx * bx + by * by + bz * bz + 2.0 * (xy * xy + xz * xz + yz * yz)) / 6.0);
    const double inv_p3 = pow(1.0 / p, 3.0);
    const double det = inv_p3 * (bx * (by * bz - yz * yz) - xy * (xy * bz - yz * xz) + xz * (xy * yz - by * xz));
    const double half = 0.5 * det;
    const double kPi = 3.14159265358979323846;
    const double angle = half <= -1.0 ? kPi / 3.0 : (half >= 1.0 ? 0.0 : acos(half) / 3.0);
    const double lam = shift + 2.0 * p * cos(angle + kPi * (2.0 / 3.0));
    const double n0 = xy * yz - xz * (yy - lam);
    const double n1 = xy * xz - yz * (xx - lam);
    const double n2 = (xx - lam) * (yy - lam) - xy * xy;
    const double len = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    double* o = normals3 + (size_t)i * 3;
    if (len == 0.0) { o[0] = 0.0; o[1] = 0.0; o[2] = 1.0; return; }
    const double unit = 1.0 / len;
    o[0] = n0 * unit; o[1] = n1 * unit; o[2] = n2 * unit;
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
// *gave_up: a generation did not settle within the sweep cap ($PWICP_REFINE_SWEEPS, default 4096; 2-5 sweeps are normal) - the
// caller restarts the pass on the host from the labels it kept
int refine_device(pwicp_context* ctx, const FePt* dP, const int* d_nb, int k, int n, double res, int* d_lab, bool* gave_up) {
    *gave_up = false;
    const int sweep_cap = getenv("PWICP_REFINE_SWEEPS") ? std::max(atoi(getenv("PWICP_REFINE_SWEEPS")), 1) : 4096;
    hipStream_t st = ctx->stream;
    FeWorkspace& ws = *workspace_of(ctx);
    DevBuf<double>&dis = ws.dis, &nd = ws.nd;
    DevBuf<unsigned long long>& key = ws.key;
    DevBuf<int>&pos = ws.pos, &cnt = ws.cnt, &La = ws.La, &Lb = ws.Lb, &nla = ws.nla, &nlb = ws.nlb, &flag = ws.flag, &tmp = ws.tmp;
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
    PWCHK(fe_read_words(ctx, *workspace_of(ctx), cnt.p + n, 1, &m));
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
        for (int in_generation = 0;; ++in_generation) {
            if (in_generation >= sweep_cap) {
                if (trace) fprintf(stderr, "[pwicp front end/dev]   refinement gives up (generation %d not settled after %d sweeps)\n", generations, in_generation);
                *gave_up = true;
                return PWICP_OK;
            }
            ++sweeps;
            HIPCHK(ctx, hipMemsetAsync(flag.p, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_ref_sweep, grid1(m), dim3(256), 0, st, L, m, d_nb, k, pos.p, d_lab, dis.p, dP, res, nl_prev, nl_new,
                               nd.p, flag.p);
            int changed = 0;
            PWCHK(fe_read_words(ctx, *workspace_of(ctx), flag.p, 1, &changed));
            std::swap(nl_prev, nl_new);
            if (!changed) break;
        }
        // (key[] of every point is kNoKey here: reset for the points of L at the start of the generation, for all at the seeds)
        hipLaunchKernelGGL(k_ref_push<0>, grid1(m), dim3(256), 0, st, L, m, d_nb, k, pos.p, d_lab, nl_prev, key.p, (int*)nullptr, (int*)nullptr);
        HIPCHK(ctx, hipMemsetAsync(cnt.p + m, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_ref_push<1>, grid1(m), dim3(256), 0, st, L, m, d_nb, k, pos.p, d_lab, nl_prev, key.p, cnt.p, (int*)nullptr);
        PWCHK(pw_exclusive_scan(ctx, cnt.p, (long long)m + 1, &tmp));
        int m_next = 0;
        PWCHK(fe_read_words(ctx, *workspace_of(ctx), cnt.p + m, 1, &m_next));
        if (m_next > n) { ctx->set_err("front end: refinement queue overflow"); return PWICP_E_INTERNAL; }
        hipLaunchKernelGGL(k_ref_push<2>, grid1(m), dim3(256), 0, st, L, m, d_nb, k, pos.p, d_lab, nl_prev, key.p, cnt.p, Lnext);
        hipLaunchKernelGGL(k_ref_commit, grid1(m), dim3(256), 0, st, L, m, nl_prev, nd.p, d_lab, dis.p, pos.p);
        std::swap(L, Lnext);
        m = m_next;
    }
    if (trace) fprintf(stderr, "[pwicp front end/dev]   refinement: %lld pops, %d generations, %d sweeps\n", pops, generations, sweeps);
    return PWICP_OK;
}

// What the first round needs of the k-NN graph alone - its roots, sizes and base lists (k_fus_first_round) and the reverse index of
// those lists (45 M entries at 1 M points: ~2 ms of atomics and scattered stores) - enqueued by the pipeline's driver BEHIND the
// download of the scatter sums: it runs while the host does the eigen step of the normals and the stream would otherwise idle.
int fusion_prepare_first_round(pwicp_context* ctx, const int* d_nb, int k, int n) {
    hipStream_t st = ctx->stream;
    FeWorkspace& ws = *workspace_of(ctx);
    ws.rev0_ready = false;
    if ((long long)n * k > (long long)INT_MAX - 64 || n < 1) return PWICP_OK;
    const size_t N = (size_t)n;
    for (DevBuf<int>* b : {&ws.root0, &ws.s0, &ws.lenA, &ws.cenA, &ws.cursor}) HIPCHK(ctx, b->reserve(N));
    HIPCHK(ctx, ws.revoff.reserve(N + 1));
    HIPCHK(ctx, ws.offA.reserve(N));
    HIPCHK(ctx, ws.revown.reserve(N * (size_t)k));
    hipLaunchKernelGGL(k_fus_first_round, grid1(n), dim3(256), 0, st, n, k, ws.root0.p, ws.s0.p, ws.lenA.p, ws.offA.p, ws.cenA.p);
    if (!ws.rev0_counted) {
        HIPCHK(ctx, hipMemsetAsync(ws.revoff.p, 0, sizeof(int) * (N + 1), st));
        hipLaunchKernelGGL(k_fus_reverse<0>, grid1(n), dim3(256), 0, st, ws.cenA.p, n, ws.root0.p, ws.lenA.p, ws.offA.p, d_nb, ws.revoff.p, (int*)nullptr);
    }
    ws.rev0_counted = false;
    PWCHK(pw_exclusive_scan(ctx, ws.revoff.p, (long long)n + 1, &ws.tmp));
    HIPCHK(ctx, hipMemcpyAsync(ws.cursor.p, ws.revoff.p, sizeof(int) * N, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_fus_reverse<1>, grid1(n), dim3(256), 0, st, ws.cenA.p, n, ws.root0.p, ws.lenA.p, ws.offA.p, d_nb, ws.cursor.p, ws.revown.p);
    ws.rev0_ready = true; ws.rev0_n = n; ws.rev0_k = k;
    return PWICP_OK;
}

// Fusion on the device: d_lab[p] = root point of p, *d_roots = the roots in ascending order.  *gave_up: a search queue or
// the list arena overflowed (the caller falls back to the host pass; nothing else is affected).
int fusion_device(pwicp_context* ctx, const FePt* dP, const int* d_nb, int k, int n, double res, int n_sv_target, int* d_lab,
                  DevBuf<int>* d_roots, int* n_roots, bool* gave_up) {
    hipStream_t st = ctx->stream;
    const bool trace = getenv("PWICP_TRACE") != nullptr;
    *gave_up = false;
    FeWorkspace& ws = *workspace_of(ctx);
    // (the first round's arrays and reverse index may stand already: fusion_prepare_first_round; a second call - the retry with a
    // larger arena - builds them here again)
    const bool rev0 = ws.rev0_ready && ws.rev0_n == n && ws.rev0_k == k;
    ws.rev0_ready = false;
    // lambda0 (:91-102)
    double lambda;
    {
        HIPCHK(ctx, ws.dmin.reserve((size_t)n));
        HIPCHK(ctx, ws.revoff.reserve((size_t)n + 1));
        if (!rev0) HIPCHK(ctx, hipMemsetAsync(ws.revoff.p, 0, sizeof(int) * ((size_t)n + 1), st));
        hipLaunchKernelGGL(k_fus_min_metric, grid1(n), dim3(256), 0, st, dP, d_nb, k, n, res, ws.dmin.p, rev0 ? (int*)nullptr : ws.revoff.p);
        // median = the value of rank n / 2 (what std::nth_element(v.begin() + v.size() / 2) leaves there)
        HIPCHK(ctx, ws.sel_state.reserve(4));
        HIPCHK(ctx, ws.sel_hist.reserve(256));
        const unsigned long long init[4] = {0ull, 0ull, (unsigned long long)(n / 2), 0ull};
        HIPCHK(ctx, hipMemcpyAsync(ws.sel_state.p, init, sizeof(init), hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipMemsetAsync(ws.sel_hist.p, 0, sizeof(unsigned) * 256, st));
        for (int shift = 56; shift >= 0; shift -= 8) {
            hipLaunchKernelGGL(k_sel_hist, dim3((unsigned)std::min(div_up(n, 256), 1024)), dim3(256), 0, st, ws.dmin.p, n, shift, ws.sel_state.p,
                               ws.sel_hist.p);
            hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(256), 0, st, ws.sel_hist.p, ws.sel_state.p, shift);
        }
        hipLaunchKernelGGL(k_sel_result, dim3(1), dim3(1), 0, st, ws.sel_state.p, (double*)(ws.sel_state.p + 3));
        double median = 0.0;
        HIPCHK(ctx, ws.host_reserve(0));
        PWCHK(fe_read_words(ctx, ws, ws.sel_state.p + 3, 2, &median));
        lambda = std::max(DBL_EPSILON, median);
        if (trace) fprintf(stderr, "[pwicp front end/dev]   lambda0 = %.17g\n", lambda);
    }
    const size_t N = (size_t)n;
    for (DevBuf<int>* b : {&ws.root0, &ws.s0, &ws.lenA, &ws.lenB, &ws.cenA, &ws.cenB, &ws.cursor, &ws.ab, &ws.ab_prev, &ws.rec_sz,
                           &ws.rec_ran, &ws.rec_absn, &ws.rec_adjn, &ws.slot_of, &ws.wake, &ws.dflag, &ws.cflag, &ws.dtmin, &ws.Wa,
                           &ws.Wb, &ws.dq, &ws.dq2, &ws.ovf, &ws.o_sz, &ws.o_ran, &ws.o_absn, &ws.o_adjn, &ws.o_dirty, &ws.o_oldabsn, &ws.cut})
        HIPCHK(ctx, b->reserve(N));
    for (DevBuf<int>* b : {&ws.revoff, &ws.alive, &ws.newlen}) HIPCHK(ctx, b->reserve(N + 1));
    for (DevBuf<long long>* b : {&ws.offA, &ws.offB, &ws.rec_ptr, &ws.o_ptr, &ws.o_oldptr}) HIPCHK(ctx, b->reserve(N));
    HIPCHK(ctx, ws.ctr.reserve(16 + 16 * 32));        // (the sweep's counters and, behind them, FusState::changed: one memset per sweep)
    HIPCHK(ctx, ws.big.reserve(16 * (kFusArenas + 1)));
    // lists of changed outcomes are appended, nothing is freed inside a round: 2.2 n k entries at most on the clouds measured
    // (the round after the first one).  3 n k = 0.54 GB per 1 M points; a round that overflows it is retried once with 6 n k
    // (segment_from_device_graph), beyond that the device pass gives up
    const unsigned long long sa_cap = (unsigned long long)ws.sa_factor * (unsigned long long)n * (unsigned long long)k;
    ws.sa_overflow = false;
    HIPCHK(ctx, ws.sa.reserve((size_t)sa_cap));
    if (!rev0) hipLaunchKernelGGL(k_fus_first_round, grid1(n), dim3(256), 0, st, n, k, ws.root0.p, ws.s0.p, ws.lenA.p, ws.offA.p, ws.cenA.p);
    int* len0 = ws.lenA.p; int* len1 = ws.lenB.p;
    long long* off0 = ws.offA.p; long long* off1 = ws.offB.p;
    int* cen = ws.cenA.p; int* cen1 = ws.cenB.p;
    const int* arena0 = d_nb;
    DevBuf<int>* arena_next = &ws.arenaA;
    DevBuf<int>* arena_cur = &ws.arenaB;       // (the one arena0 points into from the second round on)
    int nc = n, round = 0;
    bool rev_counted = true;               // (the first round's counts come from k_fus_min_metric, the later ones' from k_fus_next_lists)
    int list_entries = -1;                 // entries of the round's lists when the host knows them (-1: read back from the index)
    long long count = n;
    HIPCHK(ctx, ws.Pf.reserve(2 * N));
    hipLaunchKernelGGL(k_fus_pack_single, grid1(n), dim3(256), 0, st, dP, n, ws.Pf.p);
    FusState s{};
    s.P = dP; s.res = res;
    s.Pf = ws.Pf.p; s.inv_res_f = (float)(1.0 / res);
    s.root0 = ws.root0.p; s.s0 = ws.s0.p;
    s.revoff = ws.revoff.p;
    s.ab = ws.ab.p; s.ab_prev = ws.ab_prev.p; s.rec_sz = ws.rec_sz.p; s.rec_ran = ws.rec_ran.p; s.rec_absn = ws.rec_absn.p; s.rec_adjn = ws.rec_adjn.p;
    s.rec_ptr = ws.rec_ptr.p; s.sa = ws.sa.p; s.sa_top = ws.big.p + 16; s.sa_cap = sa_cap;
    s.slot_of = ws.slot_of.p; s.o_sz = ws.o_sz.p; s.o_ran = ws.o_ran.p; s.o_absn = ws.o_absn.p; s.o_adjn = ws.o_adjn.p; s.o_ptr = ws.o_ptr.p;
    s.o_dirty = ws.o_dirty.p; s.o_oldptr = ws.o_oldptr.p; s.o_oldabsn = ws.o_oldabsn.p;
    s.queue_limit = getenv("PWICP_FUSION_QUEUE") ? std::min(std::max(atoi(getenv("PWICP_FUSION_QUEUE")), 2), kFusQueue) : kFusQueue;
    s.wake = ws.wake.p; s.dflag = ws.dflag.p; s.cflag = ws.cflag.p; s.dtmin = ws.dtmin.p;
    int* const nWnext = ws.ctr.p; int* const ndq = ws.ctr.p + 1; int* const status = ws.ctr.p + 8;
    s.nWnext = nWnext; s.status = status;
    HIPCHK(ctx, ws.host_reserve(0));
    int* const h_ctr = ws.h_ctr;                    // pinned: read back after every sweep
    const int gs_chunk = getenv("PWICP_FUSION_CHUNK") ? std::max(atoi(getenv("PWICP_FUSION_CHUNK")), 1) : kFusChunk;
    const int batch_sweeps = getenv("PWICP_FUSION_BATCH") ? std::max(atoi(getenv("PWICP_FUSION_BATCH")), 1) : 4;
    const int batch_cap = getenv("PWICP_FUSION_BATCH_CAP") ? std::max(atoi(getenv("PWICP_FUSION_BATCH_CAP")), 64) : 1024;
    s.nW_dev = nullptr; s.stop = nullptr;
    const int wake_all_div = getenv("PWICP_FUSION_WAKE_DIV") ? std::max(atoi(getenv("PWICP_FUSION_WAKE_DIV")), 1) : 32;
    const int chunk_div = getenv("PWICP_FUSION_CHUNK_DIV") ? std::max(atoi(getenv("PWICP_FUSION_CHUNK_DIV")), 1) : 2048;   // chunks wanted per sweep (8192: 53.4 ms of fusion at 1 M points, 2048: 51.4, 512: 51.4)
    // sweeps over all centres: the tiles of one colour after the other (1: all at once, 2: checkerboard, 4: 2 x 2 pattern)
    const int n_colours = getenv("PWICP_FUSION_COLOURS") ? (atoi(getenv("PWICP_FUSION_COLOURS")) >= 4 ? 4 : (atoi(getenv("PWICP_FUSION_COLOURS")) >= 2 ? 2 : 1)) : 2;
    const int tile_pop = getenv("PWICP_FUSION_TILE") ? std::max(atoi(getenv("PWICP_FUSION_TILE")), 0) : 16;     // centres per tile; 0: index order
    for (;; lambda *= 2.0, ++round) {
        if (nc <= 1) {                                  // (:106) nothing left to fuse
            HIPCHK(ctx, hipMemcpyAsync(d_lab, ws.root0.p, sizeof(int) * N, hipMemcpyDeviceToDevice, st));
            HIPCHK(ctx, d_roots->reserve((size_t)std::max(nc, 1)));
            HIPCHK(ctx, hipMemcpyAsync(d_roots->p, cen, sizeof(int) * (size_t)nc, hipMemcpyDeviceToDevice, st));
            *n_roots = nc;
            break;
        }
        const auto t_round = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_fus_reset, grid1(n), dim3(256), 0, st, n, ws.s0.p, ws.ab.p, ws.ab_prev.p, ws.rec_sz.p, ws.rec_ran.p, ws.rec_absn.p, ws.rec_adjn.p,
                           ws.rec_ptr.p, ws.wake.p, ws.dflag.p, ws.cflag.p, ws.dtmin.p, ws.slot_of.p);
        // reverse index of the base lists
        if (!(rev0 && round == 0)) {
        if (!rev_counted) {                        // (from the second round on k_fus_next_lists has counted the entries per root)
            HIPCHK(ctx, hipMemsetAsync(ws.revoff.p, 0, sizeof(int) * (N + 1), st));
            hipLaunchKernelGGL(k_fus_reverse<0>, grid1(nc), dim3(256), 0, st, cen, nc, ws.root0.p, len0, off0, arena0, ws.revoff.p, (int*)nullptr);
        }
        PWCHK(pw_exclusive_scan(ctx, ws.revoff.p, (long long)n + 1, &ws.tmp));
        // (entries of the index = entries of the round's lists: the hand-over of the round before has told the host - no read-back,
        // and the host goes on enqueuing; the first round's lists are the graph's rows)
        int n_entries = list_entries;
        if (n_entries < 0) PWCHK(fe_read_words(ctx, ws, ws.revoff.p + n, 1, &n_entries));
        HIPCHK(ctx, hipMemcpyAsync(ws.cursor.p, ws.revoff.p, sizeof(int) * N, hipMemcpyDeviceToDevice, st));
        // (a block that has to grow goes back to the pool first: not while launches that read it may still be running)
        {
            const size_t want = (size_t)std::max(n_entries, 1);
            if (ws.revown.p && want > ws.revown.n && want * sizeof(int) > ws.revown.cap_bytes) HIPCHK(ctx, hipStreamSynchronize(st));
        }
        HIPCHK(ctx, ws.revown.reserve((size_t)std::max(n_entries, 1)));
        hipLaunchKernelGGL(k_fus_reverse<1>, grid1(nc), dim3(256), 0, st, cen, nc, ws.root0.p, len0, off0, arena0, ws.cursor.p, ws.revown.p);
        }
        HIPCHK(ctx, hipMemsetAsync(ws.big.p, 0, sizeof(unsigned long long) * 16 * (kFusArenas + 1), st));
        s.wake_all_above = std::max(nc / wake_all_div, 64);
        s.lambda = lambda; s.len0 = len0; s.off0 = off0; s.arena0 = arena0; s.revown = ws.revown.p;
        // full sweeps take the centres in tile order (k_fus_tile_keys); the certificate keeps the plain order
        const int* cen_full = cen;
        int n_parts = 1, part_begin[4] = {0, 0, 0, 0}, part_end[4] = {nc, 0, 0, 0};      // colours: cen_full[part_begin[c] .. part_end[c])
        if (tile_pop > 0 && ws.bb_set && nc >= 4096) {
            int ax[3] = {0, 1, 2};
            double ext[3] = {ws.bb_mx[0] - ws.bb_mn[0], ws.bb_mx[1] - ws.bb_mn[1], ws.bb_mx[2] - ws.bb_mn[2]};
            std::sort(ax, ax + 3, [&](int u, int v) { return ext[u] > ext[v]; });
            const int a = ax[0], b = ax[1];
            const double area = std::max(ext[a], 1e-30) * std::max(ext[b], 1e-30);
            const double edge = std::sqrt((double)tile_pop * area / (double)nc);
            const double ntx_d = std::floor(ext[a] / edge) + 1.0, nty_d = std::floor(ext[b] / edge) + 1.0;
            if (edge > 0.0 && ntx_d * nty_d < 2.0e9) {
                HIPCHK(ctx, ws.cen_t.reserve(N));
                HIPCHK(ctx, ws.tkey.reserve(N));
                HIPCHK(ctx, ws.tkey2.reserve(N));
                const unsigned ntiles = (unsigned)(ntx_d * nty_d);
                const int ncol = ntx_d * nty_d * n_colours < 4.0e9 ? n_colours : 1;
                hipLaunchKernelGGL(k_fus_tile_keys, grid1(nc), dim3(256), 0, st, dP, cen, nc, a, b, ws.bb_mn[a], ws.bb_mn[b], 1.0 / edge,
                                   (unsigned)ntx_d, ntiles, ncol, ws.tkey.p);
                int end_bit = 1;
                while (end_bit < 32 && (double)(1ull << end_bit) < ntx_d * nty_d * ncol) ++end_bit;
                size_t tb = 0;
                HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, ws.tkey.p, ws.tkey2.p, cen, ws.cen_t.p, nc, 0, end_bit, st));
                HIPCHK(ctx, ws.tsort.reserve(tb));
                HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(ws.tsort.p, tb, ws.tkey.p, ws.tkey2.p, cen, ws.cen_t.p, nc, 0, end_bit, st));
                cen_full = ws.cen_t.p;
                if (ncol > 1) {
                    HIPCHK(ctx, hipMemsetAsync(ws.ctr.p + 12, 0, 3 * sizeof(int), st));
                    for (int c = 1; c < ncol; ++c)
                        hipLaunchKernelGGL(k_fus_count_below, grid1(nc), dim3(256), 0, st, (const unsigned*)ws.tkey2.p, nc, (unsigned)c * ntiles,
                                           ws.ctr.p + 11 + c);
                    int below[3] = {0, 0, 0};
                    PWCHK(fe_read_words(ctx, ws, ws.ctr.p + 12, 3, below));
                    n_parts = 0;
                    int prev = 0;
                    for (int c = 1; c <= ncol; ++c) {                      // (empty colours drop out)
                        const int upto = c < ncol ? std::max(below[c - 1], prev) : nc;
                        if (upto > prev) { part_begin[n_parts] = prev; part_end[n_parts] = upto; ++n_parts; prev = upto; }
                    }
                }
            }
        }
        int* W = ws.Wa.p; int* Wn = ws.Wb.p;
        // what a sweep runs: the list in W, or - without a copy of it - all centres in tile order / in plain order (the certificate)
        const int* Wrun = cen_full;
        bool w_is_full = true;            // the sweep takes cen_full (all centres, tile order, first colour first)
        bool ctr_clean = false;           // the sweep's counters are zero (re-armed by the read-back of the sweep before)
        int nW = nc, sweeps = 0;
        long long runs = 0;
        // The round ends with a CERTIFICATE: one sweep over all centres, every one reading the standing state only, that
        // changes nothing - i.e. every centre's outcome follows from the outcomes before it, which is the definition of the
        // serial result.  Whatever the work lists missed before only costs more sweeps, never a label.
        bool certified = false, certify = false;
        while (!certified) {
            if (nW == 0) {
                Wrun = cen;
                nW = nc;
                w_is_full = false;
                certify = true;
                ctr_clean = false;
                // absorbers from scratch: the smallest centre whose standing outcome absorbs the node
                hipLaunchKernelGGL(k_fill<int>, grid1(n), dim3(256), 0, st, ws.ab.p, (long long)n, kNone);
                hipLaunchKernelGGL(k_fus_claim_all, grid1(nc), dim3(256), 0, st, s, cen, nc);
                HIPCHK(ctx, hipMemsetAsync(ws.ctr.p + 15, 0, sizeof(int), st));
                hipLaunchKernelGGL(k_fus_ab_changed, grid1(n), dim3(256), 0, st, ws.ab.p, ws.ab_prev.p, n, ws.ctr.p + 15);
                int differs = 0;
                PWCHK(fe_read_words(ctx, ws, ws.ctr.p + 15, 1, &differs));
                if (trace && differs) fprintf(stderr, "[pwicp front end/dev]   (round %d: absorbers rebuilt for the certificate differ)\n", round);
            }
            if (++sweeps > 20000) {                         // (never seen; the serial pass always terminates)
                if (trace) fprintf(stderr, "[pwicp front end/dev]   fusion gives up in round %d (no fixed point after %d sweeps)\n", round, sweeps);
                *gave_up = true;
                return PWICP_OK;
            }
            s.W = Wrun; s.Wnext = Wn;
            if (!certify && nW <= batch_cap && batch_sweeps > 1) {
                // short work list: batch_sweeps sweeps per read-back (a sweep of a few hundred centres is ~50 us of kernels; the
                // read-back, the wake-up of the host thread and the next enqueue cost as much again)
                if (Wrun != W) {                      // (k_fus_advance writes the list it is given: a tiny round's centres, copied)
                    HIPCHK(ctx, hipMemcpyAsync(W, Wrun, sizeof(int) * (size_t)nW, hipMemcpyDeviceToDevice, st));
                    Wrun = W; s.W = W;
                }
                if (!ctr_clean) HIPCHK(ctx, hipMemsetAsync(ws.ctr.p + 16, 0, sizeof(int) * 16 * 32, st));
                hipLaunchKernelGGL(k_fus_batch_init, dim3(1), dim3(64), 0, st, ws.ctr.p, nW);
                s.nW_dev = ws.ctr.p + 4; s.stop = ws.ctr.p + 5;
                s.changed = nullptr;
                for (int b = 0; b < batch_sweeps; ++b) {
                    hipLaunchKernelGGL((k_fus_run<kFusQueueS, kFusHashS, kFusWaves>), dim3((unsigned)div_up(batch_cap, kFusWaves)), dim3(64 * kFusWaves), 0, st, s, 0, 1,
                                       (const int*)nullptr, (const int*)nullptr, ws.ovf.p, ws.ctr.p + 3);
                    hipLaunchKernelGGL(k_fus_post, dim3(1), dim3(1024), 0, st, s, ws.dq.p, ndq);
                    hipLaunchKernelGGL(k_fus_wake, dim3(64), dim3(256), 0, st, s, ws.dq.p, ndq, ws.dq2.p, ndq + 1);
                    hipLaunchKernelGGL(k_fus_advance, dim3(1), dim3(1024), 0, st, s, W, Wn, ws.ctr.p, batch_cap, (const int*)ws.dq.p,
                                       (const int*)ws.dq2.p);
                }
                s.nW_dev = nullptr; s.stop = nullptr;
                PWCHK(fe_read_words(ctx, ws, ws.ctr.p, 16, h_ctr, ws.ctr.p, 16));
                ctr_clean = true;
                sweeps += h_ctr[6] - 1;
                runs += h_ctr[7];
                if (h_ctr[8] || h_ctr[9]) {
                    if (trace) fprintf(stderr, "[pwicp front end/dev]   fusion gives up in round %d (%s overflow)\n", round, h_ctr[8] ? "queue" : "arena");
                    ws.sa_overflow = !h_ctr[8] && h_ctr[9];
                    *gave_up = true;
                    return PWICP_OK;
                }
                if (!h_ctr[5]) { nW = h_ctr[4]; Wrun = W; w_is_full = false; continue; }      // the batch ran through (W holds the next list, possibly empty)
                if (h_ctr[3] || h_ctr[10] || h_ctr[11]) {                // everybody runs again ([3]: a search outgrew the small queue)
                    Wrun = cen_full;
                    nW = nc;
                    w_is_full = true;
                } else {                                                 // the next list outgrew the batch: back to single sweeps
                    std::swap(W, Wn);
                    Wrun = W;
                    nW = h_ctr[0];
                    w_is_full = false;
                }
                continue;
            }
            runs += nW;
            static const bool trace_sweeps = getenv("PWICP_TRACE_SWEEPS") != nullptr;
            const auto t_sweep = std::chrono::steady_clock::now();
            const int nW_sweep = nW;
            const bool cert_sweep = certify;
            if (!ctr_clean) HIPCHK(ctx, hipMemsetAsync(ws.ctr.p, 0, sizeof(int) * (16 + 16 * 32), st));
            s.changed = certify ? nullptr : ws.ctr.p + 16;
            const int chunk = certify ? 1 : std::max(1, std::min(std::min(nW / chunk_div, kFusChunk), gs_chunk));
            // A sweep over ALL centres in tile order runs colour by colour: the tiles of one colour (no two of them neighbours), their
            // outcomes and claims made standing, then the next colour - which sees what its neighbours of the colours before have just
            // decided (Gauss-Seidel between tiles as it is inside one).  Any mixture of old and new outcomes is a valid guess; the round's certificate is untouched.
            const bool halves = !certify && w_is_full && n_parts > 1 && nW == nc;
            for (int half = 0; half < (halves ? n_parts : 1); ++half) {
                const int h0 = halves ? part_begin[half] : 0, hn = halves ? part_end[half] - part_begin[half] : nW;
                s.slot0 = h0;
                if (half > 0) HIPCHK(ctx, hipMemsetAsync(ws.ctr.p + 3, 0, sizeof(int), st));
#ifdef PW_FUS_KO
                hipEvent_t ko_e0, ko_e1;
                (void)hipEventCreate(&ko_e0); (void)hipEventCreate(&ko_e1);
                (void)hipEventRecord(ko_e0, st);
#endif
                hipLaunchKernelGGL((k_fus_run<kFusQueueS, kFusHashS, kFusWaves>), dim3((unsigned)std::min(div_up(div_up(hn, chunk), kFusWaves), kFusGridCap)),
                                   dim3(64 * kFusWaves), 0, st,
                                   s, hn, chunk, (const int*)nullptr, (const int*)nullptr, ws.ovf.p, ws.ctr.p + 3);
#ifdef PW_FUS_KO
                (void)hipEventRecord(ko_e1, st);
                (void)hipEventSynchronize(ko_e1);
                float ko_ms = 0.f;
                (void)hipEventElapsedTime(&ko_ms, ko_e0, ko_e1);
                fprintf(stderr, "[pwicp knock-out %d] round %d sweep %d part %d: %d centres, k_fus_run %.3f ms\n", PW_FUS_KO, round, sweeps, half, hn, ko_ms);
                (void)hipEventDestroy(ko_e0); (void)hipEventDestroy(ko_e1);
                if (half + 1 == (halves ? n_parts : 1)) { HIPCHK(ctx, hipStreamSynchronize(st)); *gave_up = true; return PWICP_OK; }
#endif
                hipLaunchKernelGGL((k_fus_run<kFusQueue, kFusHash, 1>), dim3(256), dim3(64), 0, st, s, 0, 1, (const int*)ws.ovf.p,
                                   (const int*)(ws.ctr.p + 3), (int*)nullptr, (int*)nullptr);
                hipLaunchKernelGGL(k_fus_retract, grid1(hn), dim3(256), 0, st, s, hn);
                hipLaunchKernelGGL(k_fus_claim, grid1(hn), dim3(256), 0, st, s, hn);
            }
            s.slot0 = 0;
            hipLaunchKernelGGL(k_fus_dirty0, grid1(nW), dim3(256), 0, st, s, nW, ws.dq.p, ndq);
            // (grids for the changed nodes: at most a few per slot)
            hipLaunchKernelGGL(k_fus_wake, dim3((unsigned)std::min(1024, std::max(32, nW / 64))), dim3(256), 0, st, s, ws.dq.p, ndq, ws.dq2.p, ndq + 1);
            hipLaunchKernelGGL(k_fus_sweep_end, dim3((unsigned)std::min(256, std::max(8, nW / 256))), dim3(256), 0, st, s, ws.dq.p, ndq, ws.dq2.p,
                               ndq + 1);
            PWCHK(fe_read_words(ctx, ws, ws.ctr.p, 16, h_ctr, ws.ctr.p, 16 + 16 * 32));
            ctr_clean = true;
            if (trace && trace_sweeps)
                fprintf(stderr, "      sweep %d: %d centres%s, %d changed nodes, %.3f ms\n", sweeps, nW_sweep, cert_sweep ? " (certificate)" : "", h_ctr[1],
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sweep).count());
            if (h_ctr[8] || h_ctr[9]) {
                if (trace) fprintf(stderr, "[pwicp front end/dev]   fusion gives up in round %d (%s overflow)\n", round, h_ctr[8] ? "queue" : "arena");
                ws.sa_overflow = !h_ctr[8] && h_ctr[9];
                *gave_up = true;
                return PWICP_OK;
            }
            if (certify && h_ctr[1] == 0) { certified = true; break; }       // nothing changed against the rebuilt absorbers
            certify = false;
            if (h_ctr[10] || h_ctr[11]) {               // closure deeper than the levels / too many changes: everybody runs again (always sound)
                // (the changed nodes may not have been listed at all - k_fus_dirty0 - so their absorbers of this sweep become the
                // previous ones here, all of them at once, and no flag of a listed node is left standing: none was set)
                if (h_ctr[11]) HIPCHK(ctx, hipMemcpyAsync(ws.ab_prev.p, ws.ab.p, sizeof(int) * N, hipMemcpyDeviceToDevice, st));
                Wrun = cen_full;
                nW = nc;
                w_is_full = true;
            } else {
                std::swap(W, Wn);
                Wrun = W;
                nW = h_ctr[0];
                w_is_full = false;
            }
        }
        // absorbed in this round, per centre
        hipLaunchKernelGGL(k_fus_total_absorbed, grid1(nc), dim3(256), 0, st, cen, nc, ws.rec_absn.p, ws.newlen.p, ws.big.p);
        unsigned long long total = 0;
        PWCHK(fe_read_words(ctx, ws, ws.big.p, 2, &total));
        if (trace) {
            std::vector<unsigned long long> tops(16 * (kFusArenas + 1));
            (void)hipMemcpy(tops.data(), ws.big.p, sizeof(unsigned long long) * tops.size(), hipMemcpyDeviceToHost);
            unsigned long long used = 0, most = 0;
            for (int r = 0; r < kFusArenas; ++r) { used += tops[16 * (1 + r)]; most = std::max(most, tops[16 * (1 + r)]); }
            fprintf(stderr, "[pwicp front end/dev]   list arena: %.1f %% used, fullest region %.1f %%\n", 100.0 * (double)used / (double)sa_cap,
                    100.0 * (double)most / (double)(sa_cap / kFusArenas));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_round).count();
            fprintf(stderr, "[pwicp front end/dev]   round %d: %d centres, %d sweeps, %lld runs (%.1f x), absorbed %llu  %8.2f ms\n", round, nc,
                    sweeps, runs, (double)runs / nc, total, ms);
        }
        const long long need = count - (long long)n_sv_target;
        if (need >= 1 && (long long)total >= need) {
            // the round stops inside a centre (:139-141)
            std::vector<int> per((size_t)nc), hcen((size_t)nc);
            HIPCHK(ctx, hipMemcpyAsync(per.data(), ws.newlen.p, sizeof(int) * (size_t)nc, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipMemcpyAsync(hcen.data(), cen, sizeof(int) * (size_t)nc, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            long long before = 0;
            int t_stop = 0;
            for (; t_stop < nc; ++t_stop) {
                if (before + per[(size_t)t_stop] >= need) break;
                before += per[(size_t)t_stop];
            }
            const int stop_c = hcen[(size_t)t_stop], stop_n = (int)(need - before);
            long long ptr = 0;
            HIPCHK(ctx, hipMemcpyAsync(&ptr, ws.rec_ptr.p + stop_c, sizeof(ptr), hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            hipLaunchKernelGGL(k_fill<int>, grid1(n), dim3(256), 0, st, ws.cut.p, (long long)n, kNone);
            hipLaunchKernelGGL(k_fus_cut_positions, grid1(per[(size_t)t_stop]), dim3(256), 0, st, ws.sa.p, ptr, per[(size_t)t_stop], ws.cut.p);
            hipLaunchKernelGGL(k_fus_final, grid1(n), dim3(256), 0, st, n, ws.root0.p, ws.ab.p, stop_c, stop_n, ws.cut.p, d_lab);
            HIPCHK(ctx, hipMemsetAsync(ws.alive.p, 0, sizeof(int) * ((size_t)nc + 1), st));
            hipLaunchKernelGGL(k_fus_is_root, grid1(nc), dim3(256), 0, st, cen, nc, d_lab, ws.alive.p);
            PWCHK(pw_exclusive_scan(ctx, ws.alive.p, (long long)nc + 1, &ws.tmp));
            int nr = 0;
            HIPCHK(ctx, hipMemcpyAsync(&nr, ws.alive.p + nc, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            HIPCHK(ctx, d_roots->reserve((size_t)std::max(nr, 1)));
            hipLaunchKernelGGL(k_fus_compact, grid1(nc), dim3(256), 0, st, cen, nc, ws.alive.p, d_roots->p);
            *n_roots = nr;
            break;
        }
        count -= (long long)total;
        // hand-over to the next round
        hipLaunchKernelGGL(k_fus_new_roots, grid1(n), dim3(256), 0, st, n, ws.root0.p, ws.ab.p);
        HIPCHK(ctx, hipMemsetAsync(ws.alive.p + nc, 0, sizeof(int), st));
        HIPCHK(ctx, hipMemsetAsync(ws.newlen.p + nc, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_fus_next_sizes, grid1(nc), dim3(256), 0, st, cen, nc, ws.ab.p, ws.rec_ran.p, ws.rec_adjn.p, len0, ws.alive.p, ws.newlen.p);
        PWCHK(pw_exclusive_scan(ctx, ws.alive.p, (long long)nc + 1, &ws.tmp));
        PWCHK(pw_exclusive_scan(ctx, ws.newlen.p, (long long)nc + 1, &ws.tmp));
        int nc_next = 0, n_list = 0;
        {
            int two[2] = {0, 0};
            hipLaunchKernelGGL(k_fe_gather2, dim3(1), dim3(64), 0, st, (const unsigned*)(ws.alive.p + nc), (const unsigned*)(ws.newlen.p + nc),
                               (unsigned*)(ws.ctr.p + 12));
            PWCHK(fe_read_words(ctx, ws, ws.ctr.p + 12, 2, two));
            nc_next = two[0]; n_list = two[1];
        }
        list_entries = n_list;
        HIPCHK(ctx, arena_next->reserve((size_t)std::max(n_list, 1)));
        HIPCHK(ctx, hipMemsetAsync(ws.revoff.p, 0, sizeof(int) * (N + 1), st));        // (the closed round's index is no longer read)
        rev_counted = true;
        hipLaunchKernelGGL(k_fus_next_lists, grid1((long long)nc * 64), dim3(256), 0, st, cen, nc, ws.alive.p, ws.newlen.p, ws.ab.p, ws.rec_ran.p, ws.rec_sz.p,
                           ws.rec_absn.p, ws.rec_adjn.p, ws.rec_ptr.p, ws.sa.p, arena0, off0, len0, arena_next->p, off1, len1, ws.s0.p, cen1,
                           (const int*)ws.root0.p, ws.revoff.p);
        arena0 = arena_next->p;
        std::swap(arena_next, arena_cur);
        std::swap(len0, len1); std::swap(off0, off1); std::swap(cen, cen1);
        nc = nc_next;
    }
    return PWICP_OK;
}

__global__ void k_fe_assemble(const float4* __restrict__ cloud, const double* __restrict__ normals3, int n, FePt* __restrict__ P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = cloud[i];
    FePt p;
    p.x = (double)v.x; p.y = (double)v.y; p.z = (double)v.z;                 // S.cpp:18-22: float -> double
    p.nx = normals3[3 * (size_t)i]; p.ny = normals3[3 * (size_t)i + 1]; p.nz = normals3[3 * (size_t)i + 2];
    P[i] = p;
}

// How often the serial host passes took over from the device passes in this process (a silent take-over must show up):
// [0] clouds through the device front end, [1] fusions finished by the serial host pass because the device pass gave up (queue /
// arena overflow, sweep cap), [2] boundary refinements finished by the host pass, [3] list arenas doubled and the fusion restarted,
// [4] fusions / [5] refinements run on the host because the environment asked for it ($PWICP_FUSION=host, $PWICP_FRONTEND=host).
std::atomic<long long> g_fe_counts[6];
// fusion (device, or the serial host pass when asked for / when the device pass gives up), refinement, relabel, download
int segment_from_device_graph(pwicp_context* ctx, FeTrace& tr, const float* cloud_xyz4, const FePt* dP, const int* d_nb, int k, int n,
                              double res, int n_sv, int32_t* labels, int* n_supervoxels) {
    hipStream_t st = ctx->stream;
    FeWorkspace& ws = *workspace_of(ctx);
    DevBuf<int>&d_lab = ws.d_lab, &d_roots = ws.d_roots, &d_map = ws.d_map;
    HIPCHK(ctx, d_lab.reserve((size_t)n));
    HIPCHK(ctx, d_map.reserve((size_t)n));
    int n_roots = 0;
    bool host_fusion = getenv("PWICP_FUSION") && std::string(getenv("PWICP_FUSION")) == "host";
    g_fe_counts[0].fetch_add(1, std::memory_order_relaxed);
    if (host_fusion) g_fe_counts[4].fetch_add(1, std::memory_order_relaxed);
    const bool fusion_asked_host = host_fusion;
    if (!host_fusion) {
        PWCHK(fusion_device(ctx, dP, d_nb, k, n, res, n_sv, d_lab.p, &d_roots, &n_roots, &host_fusion));
        FeWorkspace& fws = *workspace_of(ctx);
        if (host_fusion && fws.sa_overflow && fws.sa_factor < 6) {       // the list arena was too small for this cloud: once more, doubled
            fws.sa_factor = 6;
            g_fe_counts[3].fetch_add(1, std::memory_order_relaxed);
            if (getenv("PWICP_TRACE")) fprintf(stderr, "[pwicp front end/dev]   list arena doubled, fusion restarted\n");
            PWCHK(fusion_device(ctx, dP, d_nb, k, n, res, n_sv, d_lab.p, &d_roots, &n_roots, &host_fusion));
        }
        tr.lap("fusion");
    }
    if (host_fusion) {                                  // (serial host pass: $PWICP_FUSION=host, or the device pass gave up)
        if (!fusion_asked_host) g_fe_counts[1].fetch_add(1, std::memory_order_relaxed);
        std::vector<FePt> P((size_t)n);
        std::vector<int> nb((size_t)n * k), root_of, roots;
        HIPCHK(ctx, hipMemcpyAsync(P.data(), dP, sizeof(FePt) * (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(nb.data(), d_nb, sizeof(int) * (size_t)n * k, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        if (pwhost::fe_fusion_host(P.data(), nb.data(), k, n, res, n_sv, &root_of, &roots) < 0) return PWICP_E_NOMEM;
        n_roots = (int)roots.size();
        HIPCHK(ctx, d_roots.reserve(roots.size()));
        HIPCHK(ctx, hipMemcpyAsync(d_lab.p, root_of.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(d_roots.p, roots.data(), sizeof(int) * roots.size(), hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        tr.lap("fusion (host)");
    }
    // (the fused labels are kept: a refinement that gives up is redone on the host from them)
    HIPCHK(ctx, ws.lab0.reserve((size_t)n));
    HIPCHK(ctx, hipMemcpyAsync(ws.lab0.p, d_lab.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, st));
    bool host_refine = false;
    PWCHK(refine_device(ctx, dP, d_nb, k, n, res, d_lab.p, &host_refine));
    if (host_refine) {
        g_fe_counts[2].fetch_add(1, std::memory_order_relaxed);
        std::vector<FePt> P((size_t)n);
        std::vector<int> nb((size_t)n * k), root_of((size_t)n);
        HIPCHK(ctx, hipMemcpyAsync(P.data(), dP, sizeof(FePt) * (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(nb.data(), d_nb, sizeof(int) * (size_t)n * k, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(root_of.data(), ws.lab0.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        pwhost::fe_refine_host(P.data(), nb.data(), k, n, res, &root_of);
        HIPCHK(ctx, hipMemcpyAsync(d_lab.p, root_of.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    tr.lap(host_refine ? "boundary refinement (host)" : "boundary refinement");
    hipLaunchKernelGGL(k_mark_roots, grid1(n_roots), dim3(256), 0, st, d_roots.p, n_roots, d_map.p);
    hipLaunchKernelGGL(k_relabel, grid1(n), dim3(256), 0, st, d_lab.p, n, d_map.p);
    HIPCHK(ctx, hipMemcpyAsync(labels, d_lab.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    *n_supervoxels = n_roots;
    tr.lap("relabel + download");
    (void)cloud_xyz4;
    return PWICP_OK;
}

}  // namespace

extern "C" void pw_frontend_count(int which) { if (which >= 0 && which < 6) g_fe_counts[which].fetch_add(1, std::memory_order_relaxed); }
extern "C" PWICP_API int pwicp_frontend_fallback_counts(long long* counts6) {
    if (!counts6) return PWICP_E_INVALID;
    for (int k = 0; k < 6; ++k) counts6[k] = g_fe_counts[k].load(std::memory_order_relaxed);
    return PWICP_OK;
}

void pw_frontend_release_workspace(pwicp_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->scratch.reset();
}

std::shared_ptr<void>* pw_context_host_slot(pwicp_context* ctx) { return ctx ? &ctx->host_slot : nullptr; }

// The whole front end of one cloud (S.cpp:18-68).  Device: k-NN graph, neighbourhood scatter, occupied cells, fusion,
// refinement, relabel.  Host: only the closed-form eigen step of the normals (libm's pow / acos / cos decide label bits).
int pw_frontend_segment_device(pwicp_context* ctx, const float* cloud_xyz4, int n, int k, float cell_edge, float sv_resolution,
                               int32_t* labels, int* n_supervoxels) {
    if (k > 64) { ctx->set_err("front end: k > 64 neighbours not supported on the device"); return PWICP_E_INVALID; }
    if ((long long)n * k > (long long)INT_MAX - 64) {          // list offsets / entry counts are 32-bit (47 M points at k = 45)
        ctx->set_err("front end: cloud too large for the device pipeline (n * k must stay below 2^31)");
        return PWICP_E_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    FeTrace tr{ctx};
    FeWorkspace& ws = *workspace_of(ctx);
    ws.rev0_ready = false; ws.rev0_counted = false;      // (whatever a call that failed half-way left standing)
    DevBuf<float4>& pts = ws.pts;
    HIPCHK(ctx, pts.reserve((size_t)n));
    HIPCHK(ctx, hipMemcpyAsync(pts.p, cloud_xyz4, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, st));
    DevBuf<int>& d_nb = ws.d_nb;
    HIPCHK(ctx, d_nb.reserve((size_t)n * k));
    {
        Grid& g = ws.grid;
        PWCHK(pw_grid_build(ctx, pts.p, n, cell_edge > 0.f ? cell_edge : pw_estimate_cell_edge(cloud_xyz4, n), &g));
        // (the sizes of the graph's reverse index are counted on the way: fusion_prepare_first_round)
        const bool ahead_k = !(getenv("PWICP_FE_AHEAD") && atoi(getenv("PWICP_FE_AHEAD")) == 0);
        int* rev_count = nullptr;
        if (ahead_k && n >= 1 && n > k) {
            HIPCHK(ctx, ws.revoff.reserve((size_t)n + 1));
            HIPCHK(ctx, hipMemsetAsync(ws.revoff.p, 0, sizeof(int) * ((size_t)n + 1), st));
            rev_count = ws.revoff.p;
        }
        ws.rev0_counted = rev_count != nullptr;
        PWCHK(pw_knn_launch(ctx, g.d, k, d_nb.p, rev_count));                    // S.cpp:30-41
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    tr.lap("k-NN graph");
    // normals: scatter on the device, eigen step on the host
    DevBuf<double>&dS = ws.dS, &dN = ws.dN;
    DevBuf<FePt>& dP = ws.dP;
    HIPCHK(ctx, ws.host_reserve((size_t)n));
    if (getenv("PWICP_TRACE_NORMALS")) tr.lap("  normals: pinned staging");
    HIPCHK(ctx, dS.reserve((size_t)n * 6));
    HIPCHK(ctx, dN.reserve((size_t)n * 3));
    HIPCHK(ctx, dP.reserve((size_t)n));
    double* const S6 = ws.hS;
    double* const N3 = ws.hN;
    // The sums come down in pieces - scatter of piece c, its copy, an event - and the host takes the eigen step of a piece as soon as it
    // is there, sending its normals back up at once: the eigen step (4.5 ms per 1 M points on the host threads) runs beside the scatter
    // and the copies (3.9 ms) instead of after them.  The host waits for THOSE copies only; behind them the stream goes on with what
    // needs the graph and the points but no normals - the first round's reverse index, the occupied cells.
    // $PWICP_FE_AHEAD=0: one piece, everything else after the normals.
    const bool ahead = !(getenv("PWICP_FE_AHEAD") && atoi(getenv("PWICP_FE_AHEAD")) == 0);
    const int pieces_env = getenv("PWICP_FE_PIECES") ? std::min(std::max(atoi(getenv("PWICP_FE_PIECES")), 1), (int)FeWorkspace::kPieces) : (int)FeWorkspace::kPieces;
    // (pieces of at least ~120 k points: a 140 k-point scan in eight pieces is 1 ms slower than in one)
    const int pieces = ahead ? std::max(1, std::min(pieces_env, n / 120000)) : 1;
    auto piece_lo = [&](int c) { return (int)((long long)n * c / pieces); };
    for (int c = 0; c < pieces; ++c) {
        const int lo = piece_lo(c), hi = piece_lo(c + 1);
        hipLaunchKernelGGL(k_fe_scatter, grid1(hi - lo), dim3(256), 0, st, pts.p, d_nb.p, k, lo, hi, dS.p);
        HIPCHK(ctx, hipMemcpyAsync(S6 + 6 * (size_t)lo, dS.p + 6 * (size_t)lo, sizeof(double) * 6 * (size_t)(hi - lo), hipMemcpyDeviceToHost, st));
        if (ahead) {
            if (!ws.ev_down[c]) HIPCHK(ctx, hipEventCreateWithFlags(&ws.ev_down[c], hipEventDisableTiming));
            HIPCHK(ctx, hipEventRecord(ws.ev_down[c], st));
        }
    }
    if (ahead) PWCHK(fusion_prepare_first_round(ctx, d_nb.p, k, n));
    // bounding box for the cell count meanwhile (grid_sample.h:36-44)
    double mn[3], mx[3];
    pwhost::fe_bounding_box(cloud_xyz4, n, mn, mx);
    const double res = (double)sv_resolution;
    const int s1 = (int)((mx[0] - mn[0]) / res + 1), s2 = (int)((mx[1] - mn[1]) / res + 1), s3 = (int)((mx[2] - mn[2]) / res + 1);
    const bool cells_on_host = std::max(s1, std::max(s2, s3)) > (1 << 21);
    auto count_cells = [&]() -> int {
        size_t cap = 1;
        while (cap < 2 * (size_t)n) cap <<= 1;
        DevBuf<unsigned long long>& table = ws.table;
        DevBuf<int>& cnt = ws.cell_cnt;
        HIPCHK(ctx, table.reserve(cap));
        HIPCHK(ctx, cnt.reserve(1));
        HIPCHK(ctx, hipMemsetAsync(table.p, 0xff, sizeof(unsigned long long) * cap, st));
        HIPCHK(ctx, hipMemsetAsync(cnt.p, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_fe_count_cells, grid1(n), dim3(256), 0, st, pts.p, n, mn[0], mn[1], mn[2], res, s1, s2, s3, table.p,
                           (unsigned long long)(cap - 1), cnt.p);
        return PWICP_OK;
    };
    if (ahead && !cells_on_host) PWCHK(count_cells());
    const bool eigen_on_device = getenv("PWICP_NORMALS") && std::string(getenv("PWICP_NORMALS")) == "device";
    if (!ahead) HIPCHK(ctx, hipStreamSynchronize(st));
    if (getenv("PWICP_TRACE_NORMALS")) tr.lap("  normals: buffers, scatter, sums down, box");
    for (int c = 0; c < pieces; ++c) {
        const int lo = piece_lo(c), hi = piece_lo(c + 1);
        if (ahead) HIPCHK(ctx, hipEventSynchronize(ws.ev_down[c]));
        pwhost::fe_normals_from_scatter(S6 + 6 * (size_t)lo, hi - lo, N3 + 3 * (size_t)lo);
        if (!eigen_on_device)
            HIPCHK(ctx, hipMemcpyAsync(dN.p + 3 * (size_t)lo, N3 + 3 * (size_t)lo, sizeof(double) * 3 * (size_t)(hi - lo), hipMemcpyHostToDevice, st));
    }
    if (getenv("PWICP_TRACE_NORMALS")) tr.lap("  normals: eigen step (host)");
    if (eigen_on_device) {
        // experiment: eigen step on the device; report how many normals differ from the host's in any bit
        hipLaunchKernelGGL(k_fe_eigen_device, grid1(n), dim3(256), 0, st, dS.p, n, dN.p);
        std::vector<double> dev3((size_t)n * 3);
        HIPCHK(ctx, hipMemcpyAsync(dev3.data(), dN.p, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        size_t differ = 0;
        double worst = 0.0;
        for (int i = 0; i < n; ++i) {
            bool d = false;
            for (int c = 0; c < 3; ++c) {
                const double a = dev3[3 * (size_t)i + c], b = N3[3 * (size_t)i + c];
                if (memcmp(&a, &b, sizeof(double)) != 0) { d = true; worst = std::max(worst, fabs(a - b)); }
            }
            differ += d;
        }
        fprintf(stderr, "[pwicp front end/dev]   eigen step on the device: %zu of %d normals differ from libm's (largest component difference %.3g)\n",
                differ, n, worst);
    }
    hipLaunchKernelGGL(k_fe_assemble, grid1(n), dim3(256), 0, st, pts.p, dN.p, n, dP.p);
    tr.lap("pca normals");
    int n_sv = 0;
    if (cells_on_host) {
        // the 64-bit cell key packs 21 bits per axis: beyond that (extent / resolution > 2 M cells on an axis; GridSample
        // allows INT_MAX) the cells are counted by the host pass on the downloaded points
        std::vector<FePt> hP((size_t)n);
        HIPCHK(ctx, hipMemcpyAsync(hP.data(), dP.p, sizeof(FePt) * (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        n_sv = pwhost::fe_count_occupied_cells(hP.data(), n, res);
    } else {
        if (!ahead) PWCHK(count_cells());
        PWCHK(fe_read_words(ctx, ws, ws.cell_cnt.p, 1, &n_sv));
    }
    tr.lap("occupied cells");
    for (int d = 0; d < 3; ++d) { ws.bb_mn[d] = mn[d]; ws.bb_mx[d] = mx[d]; }
    ws.bb_set = true;
    return segment_from_device_graph(ctx, tr, cloud_xyz4, dP.p, d_nb.p, k, n, res, n_sv, labels, n_supervoxels);
}
