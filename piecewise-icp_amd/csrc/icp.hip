// Inner point-to-plane ICP and the variance-covariance matrix on the device.
//
// Reference: P2PICPwithPatchNormal src/Registration.cpp:1255-1269 (= pcl::IterativeClosestPointWithNormals::
// align: correspondence search, TransformationEstimationPointToPlaneLLS, transformPointCloudWithNormals,
// DefaultConvergenceCriteria) and calTransParaVCM Registration.cpp:1273-1343.
//
// One inner iteration = ONE launch (k_icp_iter), no host round trip:
//   accumulate (grid over the S stable centroids, 8 lanes per centroid): applies the previous incremental transform to
//                the working source (points + normals), exact 1-NN in the target-centroid grid, forms the row
//                [a b c nx ny nz | d] in float exactly as PCL does, widens to double and reduces the 21+6
//                sums (+ sum of d2 for the MSE test) with wave shuffles -> LDS -> one partial per block;
//   solve      (first wave of the block that finishes last): fixed-order sum of the block partials, 6x6 LU inverse,
//                x = inv*ATb, Rz*Ry*Rx matrix in double -> float, final = T*final, convergence tests, done flag,
//                and — in the last launch of a batch — the iteration record to the host mailbox.
// Launches after convergence are no-ops (done flag), so iterations are enqueued in small batches.
// calTransParaVCM is two launches of the same shape (k_vcm_normal, k_vcm_finish).
#include "classify_dev.h"
#include "common.h"
#include "devmath.h"
#include "icp.h"
#include "nn_device.h"
#include "select_dev.h"
#include "xform_dev.h"

using namespace pwdev;

namespace {

constexpr int kBlock = 256;
constexpr int kNSums = 28;    // 21 ATA (upper triangle) + 6 ATb + 1 sum(d2)

// -DPWICP_KTRACE: wall-clock stamps (s_memrealtime, 10 ns) of the phases of the fused launch, read by tools/ktrace.py
#ifdef PWICP_KTRACE
__device__ unsigned long long pw_ktrace[32];
#define KT_STAMP(slot_) do { if ((threadIdx.x & 63) == 0) pw_ktrace[slot_] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define KT_MIN(slot_) do { if (threadIdx.x == 0) atomicMin(&pw_ktrace[slot_], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#define KT_MAX(slot_) do { if (threadIdx.x == 0) atomicMax(&pw_ktrace[slot_], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
// start / end / role of every block of the last k_xf_vcm launch (plain stores), tools/ktrace_front.py
constexpr int kVtBlocks = 4096;
__device__ unsigned long long pw_vblk[3 * kVtBlocks];
#define VT_BEGIN(r_) do { if (threadIdx.x == 0 && blockIdx.x < kVtBlocks) { pw_vblk[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); pw_vblk[3 * blockIdx.x + 2] = (r_); } } while (0)
#define VT_END() do { if (threadIdx.x == 0 && blockIdx.x < kVtBlocks) pw_vblk[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int pwicp_debug_vtrace(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pw_vblk), sizeof(unsigned long long) * 3 * kVtBlocks) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[3 * kVtBlocks];
        if (hipMemcpyToSymbol(HIP_SYMBOL(pw_vblk), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define VT_BEGIN(r_) do { } while (0)
#define VT_END() do { } while (0)
#define KT_STAMP(slot_) do { } while (0)
#define KT_MIN(slot_) do { } while (0)
#define KT_MAX(slot_) do { } while (0)
#endif

__device__ __forceinline__ void block_reduce_store(double* v, int nv, double* __restrict__ partial_out) {
    __shared__ double sh[kBlock / 64][kNSums];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < nv; ++k) {
        double x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if (lane == 0) sh[wave][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < nv) {
        double s = sh[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) s += sh[w][threadIdx.x];
        partial_out[threadIdx.x] = s;
    }
}

// Synchronisation inside ONE wave that communicates through LDS: the lanes run in lockstep and the LDS serves a wave's
// requests in order, so a fence (waits + no compiler reordering) is enough; no s_barrier, hence usable by the first
// wave of a larger block while its other waves have already left.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Device -> host mailbox message of an ICP batch (see k_mail in loop.hip): word ranges a | b | the ICP state, then the
// sequence number with a system-scope release.  Executed by the first wave of one block.
__device__ __forceinline__ void icp_send_mail(const IcpMail& m, const IcpState* st) {
    const int t = threadIdx.x;
    if (t >= 64) return;
    const unsigned* c = (const unsigned*)st;
    const int nc = (int)(sizeof(IcpState) / 4);
    for (int i = t; i < m.na; i += 64) mail_store(&m.dst[i], __hip_atomic_load(&m.a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int i = t; i < m.nb; i += 64) mail_store(&m.dst[m.na + i], __hip_atomic_load(&m.b[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int i = t; i < nc; i += 64) mail_store(&m.dst[m.na + m.nb + i], __hip_atomic_load(&c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    mail_drain();
    wave_sync();
    if (t == 0) mail_publish(m.seq_ptr, m.seq);
}

// The same message without reading anything back: the word ranges a | b are requested when a tail STARTS (they are final by
// then: slot words of earlier launches, or atomics that were performed before the last block was counted), the state words
// come from the tail's LDS copy (icp_solve_tail: state_words).  Saves the drain of the state stores and a round trip of coherent
// loads at the end of every launch that carries a message (~1.3 us each).  One wave; na + nb <= 64.
constexpr int kStateWords = (int)(sizeof(IcpState) / 4);
constexpr int kTailWords = kStateWords + 2;          // the tail's LDS copy: the state | the stage guard's two words
__device__ __forceinline__ unsigned mail_prefetch(const IcpMail& m) {
    const int t = threadIdx.x;
    unsigned v = 0;
    if (m.dst) {
        if (t < m.na) v = __hip_atomic_load(&m.a[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (t < m.na + m.nb) v = __hip_atomic_load(&m.b[t - m.na], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v;
}
__device__ __forceinline__ void icp_send_mail_fast(const IcpMail& m, unsigned pre, const unsigned* state_words, const StageGuard& sg) {
    const int t = threadIdx.x;
    if (sg.out) {                            // the guard's words lie in the range a (slot words): this tail has just written them
        const int gi = (int)(sg.out - m.a);
        if (gi >= 0 && gi + 1 < m.na) { if (t == gi) pre = state_words[kStateWords]; if (t == gi + 1) pre = state_words[kStateWords + 1]; }
    }
    if (t < m.na + m.nb) mail_store(&m.dst[t], pre);
    if (t < kStateWords) mail_store(&m.dst[m.na + m.nb + t], state_words[t]);
    mail_drain();
    wave_sync();
    if (t == 0) mail_publish(m.seq_ptr, m.seq);
}

// Group-cooperative accumulate: kGroup (8) consecutive lanes share one stable centroid.  They split the rows of its
// nearest-neighbour search (nn_query_group: the search is a chain of dependent memory round trips at this size),
// then each lane keeps 4 of the 28 sums of the point's LLS row, so the reduction over the wave's 8 points needs
// 3 shuffle steps on 4 values instead of 6 steps on 28.
constexpr int kAccBlock = 1024;                       // 128 points per block: few partials for the solve kernel
constexpr int kAccPts = kAccBlock / kGroup;

struct TailPrev { float F; int iters; double mse; };        // what a tail needs of the previous iteration's state
__device__ __forceinline__ TailPrev tail_prefetch(const IcpState* st, bool first);
__device__ __forceinline__ void icp_solve_tail(IcpState* st, const double* sums, int ns, double mse_rel, bool first, TailPrev pv, unsigned* state_words,
                                               const StageGuard& sg);
__device__ __forceinline__ void tail_sums_block(const double* partials, int nblocks, double (*segs)[32], double* sums);

// Stores that other blocks / the mailbox wave read back in the SAME launch go through device-coherent (write-through) atomics
// and are read back with device-coherent atomic loads; the writer drains its store queue (s_waitcnt vmcnt(0)) before it
// signals (block counter, mailbox sequence number).  That is what stands in for a release / acquire fence pair here: a fence
// writes the L2 back and invalidates the L1 (~3.5 us a pair on MI355X), which is as long as the rest of the tail.
// Data handled this way - and ONLY to be touched this way inside a launch: the block partials (`partials`), the block
// aggregates of the fused classification (`agg`), the slot words [0..3] and the whole IcpState.  gfx9 counts stores in vmcnt
// (no separate vscnt), which the drain relies on: this file is gfx950 only (see the static_assert below).
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "icp.hip relies on gfx9 memory-counter behaviour (stores counted by vmcnt): build for gfx950 only"
#endif
template <typename T>
__device__ __forceinline__ void coh_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T coh_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// One inner iteration in ONE launch: every block accumulates its 128 points; the block that finishes last (device
// counter) reduces the partials in a fixed order and solves the 6x6 system — the former second kernel, whose launch
// and first dependent loads cost as much as its arithmetic.  `mail` (optional, last launch of a batch) publishes the
// iteration record to the host mailbox from the same launch.
__global__ void __launch_bounds__(kAccBlock) k_icp_iter(GridDesc g, const float4* __restrict__ tgt,
                                                        const float4* __restrict__ tgt_n, float4* __restrict__ src,
                                                        float4* __restrict__ srcn, int ns_host,
                                                        const unsigned* __restrict__ ns_dev, IcpState* st,
                                                        double* __restrict__ partials, unsigned* __restrict__ counter,
                                                        double mse_rel, IcpMail mail, FusedSelect fs, int fs_pass, StageGuard sg) {
    __shared__ double sh[kAccBlock / 64][32];
    // Leading blocks (first iteration of a run only): pass 1 or 2 of the percentile selection of the dense search that was
    // enqueued just BEFORE this ICP batch (loop.hip: the search does not depend on the ICP).  The ICP leaves most of the chip
    // idle, so the passes cost it nothing, and the percentile reaches the host together with the ICP's result instead of
    // at the end of the update launch that follows.
    const int nsel = fs.scratch ? fs.nblk : 0;
    if ((int)blockIdx.x < nsel) {
        __shared__ unsigned s_hist[kFsBins];
        if (threadIdx.x >= 256) return;
        if (fs_pass == 1) fs_pass_embedded<1>(s_hist, fs, (int)blockIdx.x, 0u, 256);
        else fs_pass_embedded<2>(s_hist, fs, (int)blockIdx.x, 0u, 256);
        return;
    }
    const int bx = (int)blockIdx.x - nsel;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = threadIdx.x % kGroup;
    const int i = bx * kAccPts + threadIdx.x / kGroup;
    // Everything the launch reads before its search is requested at once - the state's flags, the count, the previous estimate
    // and the lane's point (ns_host bounds the arrays whether or not the count lives on the device) - instead of one
    // dependent round trip after the other (done -> count -> point and estimate: ~1.5 us before the first cell is looked at).
    const int done_in = st->done, iters_in = st->iters;
    const unsigned ns_in = ns_dev ? *ns_dev : (unsigned)ns_host;
    float Tp[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) Tp[e] = st->T[e];
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), nrm = p;
    if (i < ns_host) { p = src[i]; nrm = srcn[i]; }
    if (done_in) {                  // converged in an earlier launch of the batch: only the message is left to do
        if (mail.dst && bx == 0) icp_send_mail(mail, st);
        return;
    }
    const int ns = (int)ns_in;                                // the count may live on the device (no host sync)
    if (ns <= 0) {
        if (mail.dst && bx == 0) icp_send_mail(mail, st);
        return;
    }
    if ((int)(bx * kAccPts) >= ns) return;
    if (bx == 0 && threadIdx.x == 0) KT_STAMP(16);
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
    if (i < ns) {
        if (iters_in > 0) {        // transformPointCloudWithNormals with the previous estimate
            p = xform_point(Tp, p);
            nrm = xform_normal(Tp, nrm);
            if (sub == 0) { src[i] = p; srcn[i] = nrm; }
        }
        const NNBest b = nn_query_group(g, p.x, p.y, p.z, sub);
        KT_MAX(17);
        const int bi = b.idx();
        const float4 t = tgt[bi], n = tgt_n[bi];
        const float sx = p.x, sy = p.y, sz = p.z, dx = t.x, dy = t.y, dz = t.z, nx = n.x, ny = n.y, nz = n.z;
        const double a = (double)(nz * sy - ny * sz);
        const double bb = (double)(nx * sz - nz * sx);
        const double c = (double)(ny * sx - nx * sy);
        const double d = (double)(nx * dx + ny * dy + nz * dz - nx * sx - ny * sy - nz * sz);
        // sum k (0..27) lives on lane sub = k % 8 as its (k / 8)-th value
        switch (sub) {
            case 0: w0 = a * a;   w1 = bb * nx;            w2 = (double)(nx * ny); w3 = nx * d; break;   // 0, 8, 16, 24
            case 1: w0 = a * bb;  w1 = bb * ny;            w2 = (double)(nx * nz); w3 = ny * d; break;   // 1, 9, 17, 25
            case 2: w0 = a * c;   w1 = bb * nz;            w2 = (double)(ny * ny); w3 = nz * d; break;   // 2, 10, 18, 26
            case 3: w0 = a * nx;  w1 = c * c;              w2 = (double)(ny * nz); w3 = (double)b.d2(); break;   // 3, 11, 19, 27
            case 4: w0 = a * ny;  w1 = c * nx;             w2 = (double)(nz * nz); break;                // 4, 12, 20
            case 5: w0 = a * nz;  w1 = c * ny;             w2 = a * d; break;                            // 5, 13, 21
            case 6: w0 = bb * bb; w1 = c * nz;             w2 = bb * d; break;                           // 6, 14, 22
            default: w0 = bb * c; w1 = (double)(nx * nx);  w2 = c * d; break;                            // 7, 15, 23
        }
    }
    // over the 8 points of the wave (lanes with equal `sub`), fixed order
#pragma unroll
    for (int o = kGroup; o < 64; o <<= 1) {
        w0 += __shfl_xor(w0, o); w1 += __shfl_xor(w1, o); w2 += __shfl_xor(w2, o); w3 += __shfl_xor(w3, o);
    }
    if (lane < kGroup) { sh[wave][lane] = w0; sh[wave][8 + lane] = w1; sh[wave][16 + lane] = w2; sh[wave][24 + lane] = w3; }
    __syncthreads();
    if (threadIdx.x < kNSums) {
        double acc = sh[0][threadIdx.x];
        for (int w = 1; w < kAccBlock / 64; ++w) acc += sh[w][threadIdx.x];
        // write-through (device-coherent) store: the partial sums reach the coherence point without a cache write-back
        __hip_atomic_store(&partials[(size_t)bx * kNSums + threadIdx.x], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // last active block -> solve.  Wave 0 (which stored the partials) drains its stores and counts the block; the block that
    // turns out to be the last one sums the partials of all blocks with ALL its waves (one round trip of loads), then its first
    // wave solves (wave-level synchronisation only, see icp_solve_tail).  The partials are write-through stores read back with
    // device-coherent loads, so no fence (= L2 write-back + L1 invalidate, ~3.5 us a pair) is needed on either side: draining
    // the store queue before the count is the release.
    __shared__ unsigned s_last;
    __shared__ double s_sums[kNSums];
    const unsigned nact = (unsigned)((ns + kAccPts - 1) / kAccPts);
    if (threadIdx.x < 64) {
        drain_stores();
        KT_MAX(18);
        if (threadIdx.x == 0) {
            const unsigned prev = atomicAdd(counter, 1u);
            const unsigned last = (prev == nact - 1u) ? 1u : 0u;
            if (last) *counter = 0u;                         // re-armed for the next launch
            s_last = last;
        }
    }
    __syncthreads();
    if (!s_last) return;
    KT_STAMP(19);
    __shared__ unsigned s_state[kTailWords];
    const bool fast_mail = mail.dst && mail.na + mail.nb <= 64;
    const TailPrev pv = tail_prefetch(st, false);
    const unsigned pre = (fast_mail && threadIdx.x < 64) ? mail_prefetch(mail) : 0u;
    tail_sums_block(partials, (int)nact, sh, s_sums);
    if (threadIdx.x >= 64) return;
    icp_solve_tail(st, s_sums, ns, mse_rel, false, pv, s_state, sg);
    KT_STAMP(25);
    if (fast_mail) icp_send_mail_fast(mail, pre, s_state, sg);
    else if (mail.dst) {
        drain_stores();                                      // the state went out through coherent stores (icp_solve_tail)
        wave_sync();
        icp_send_mail(mail, st);
    }
}

// (Round 5's single-workgroup form of these iterations for small problems - k_icp_small: the target's fine level in LDS, one lane per
// centroid, a batch's iterations back to back; bit-identical, 35 - 48 us per iteration against 15 - was removed in round 6:
// profiles/r05_icp_small.txt, the code is in the history at 195adaa.)
constexpr int kTailSegs = 16;                   // segments of tail_sums_block (below)

// value of lane `lane` (wave-uniform index) in every lane: v_readlane, no LDS round trip
__device__ __forceinline__ double lane_bcast(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 6x6 inverse by LU with partial pivoting on ONE wave, operands in registers.  Lane j < 6 holds column j of A, lane 6 + c
// column c of the identity: the elimination of [A | I] row by row IS the serial algorithm (devmath.h inv6) - the row swaps
// applied to the identity are its `piv`, the updates of the identity's columns its forward substitution (same operands, same
// order: y_i loses L_i0 y_0, then L_i1 y_1, ...), the back substitution runs on lanes 6..11 with U broadcast from lanes 0..5 -
// so the result is bit-identical; what is gone are the LDS round trips between the steps (v_readlane broadcasts instead:
// every index below is a compile-time constant after unrolling).  A is read from LDS, inv written to LDS.
__device__ __forceinline__ void inv6_wave(double (*A)[6], double (*inv)[6], bool* singular) {
    const int t = threadIdx.x;
    const int col = t < 6 ? t : 0;
    double a[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = (t < 6) ? A[i][col] : ((t - 6 == i) ? 1.0 : 0.0);
    bool sing = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        // pivot of column k (lane k), rows k..5: first maximum of |.|
        int p = k;
        double best = fabs(a[k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i)
            if (fabs(a[i]) > best) { best = fabs(a[i]); p = i; }
        p = __builtin_amdgcn_readlane(p, k);
        if (lane_bcast(best, k) == 0.0) sing = true;
#pragma unroll
        for (int i = k + 1; i < 6; ++i)
            if (p == i) { const double tmp = a[k]; a[k] = a[i]; a[i] = tmp; }          // wave-uniform branch
        // multipliers (lane k keeps them as L), then row i loses l_i * row k in the columns right of k and in the identity's
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double q = a[i] / a[k];
            const double l = lane_bcast(q, k);
            const double upd = a[i] - l * a[k];
            a[i] = (t == k) ? q : ((t > k) ? upd : a[i]);
        }
    }
    // back substitution on lanes 6..11 (x overwrites y from the bottom up); U(i, j) = a[i] of lane j
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double sacc = a[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) sacc = sacc - lane_bcast(a[i], j) * a[j];
        const double x = sacc / lane_bcast(a[i], i);
        if (t >= 6) a[i] = x;
    }
    if (t >= 6 && t < 12) {
#pragma unroll
        for (int i = 0; i < 6; ++i) inv[i][t - 6] = sing ? (double)NAN : a[i];
    }
    if (t == 0) *singular = sing;
    wave_sync();
}

// requested at the START of a tail (with the partial sums): two dependent round trips less after T
__device__ __forceinline__ TailPrev tail_prefetch(const IcpState* st, bool first) {
    TailPrev pv{0.f, 0, 1.7976931348623157e308};
    if (!first) {
        if (threadIdx.x < 16) pv.F = st->Tfinal[threadIdx.x];
        if (threadIdx.x == 0) { pv.iters = st->iters; pv.mse = st->prev_mse; }
    }
    return pv;
}

// the 28 sums over the block partials on ONE wave: two lanes per sum (even / odd blocks), then one add - a fixed order
__device__ __forceinline__ void tail_sums_wave(const double* partials, int nblocks, double* sums) {
    __shared__ double half[2][kNSums];
    const int t = threadIdx.x;
    if (t < 2 * kNSums) {
        const int k = t % kNSums, h = t / kNSums;
        double s = 0.0;
        // (written by other blocks of this launch through coherent stores.)  Many loads in flight, then the adds in block
        // order: the summation order stays fixed, the latency is paid once per pass
        int b = h;
        for (; b < nblocks; b += 64) {            // 32 guarded loads in flight (typical launches: one pass)
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u)
                v[u] = (b + 2 * u < nblocks) ? __hip_atomic_load(&partials[(size_t)(b + 2 * u) * kNSums + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (b + 2 * u < nblocks) s += v[u];
        }
        half[h][k] = s;
    }
    wave_sync();
    if (t < kNSums) sums[t] = half[0][t] + half[1][t];
    wave_sync();
}

// the same on a whole block of >= kTailSegs * 28 threads (the block that finished last, all of its waves still there):
// thread (k, seg) takes the blocks seg, seg + kTailSegs, ..., then the segments in order.  One round trip of loads for up to
// 4 * kTailSegs blocks instead of one per 64.  `segs` : kTailSegs x 32 doubles of LDS.  Ends with a block barrier.
__device__ __forceinline__ void tail_sums_block(const double* partials, int nblocks, double (*segs)[32], double* sums) {
    const int t = threadIdx.x;
    if (t < kTailSegs * kNSums) {
        const int k = t % kNSums, seg = t / kNSums;
        double s = 0.0;
        for (int b = seg; b < nblocks; b += 8 * kTailSegs) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = (b + u * kTailSegs < nblocks) ? __hip_atomic_load(&partials[(size_t)(b + u * kTailSegs) * kNSums + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b + u * kTailSegs < nblocks) s += v[u];
        }
        segs[seg][k] = s;
    }
    __syncthreads();
    if (t < kNSums) {
        double s = segs[0][t];
#pragma unroll
        for (int g = 1; g < kTailSegs; ++g) s += segs[g][t];
        sums[t] = s;
    }
    __syncthreads();
}

// 6x6 LU inverse of the summed system, x = inv*ATb, T from (alpha,beta,gamma,t), convergence tests of
// pcl::registration::DefaultConvergenceCriteria.  Runs on ONE wave (threadIdx.x < 64); `sums`: the 28 sums in LDS.
// `first`: iteration 0 of a call (the state is not read: final = identity, no previous MSE).
// (inlined on purpose: a call makes the kernel use scratch memory, and a dispatch that needs scratch behind one that does not -
// or the other way round - costs ~6 us of dispatch latency on MI355X: two such bubbles per outer iteration)
__device__ __forceinline__ void icp_solve_tail(IcpState* st, const double* sums, int ns, double mse_rel, bool first, TailPrev pv, unsigned* state_words,
                                               const StageGuard& sg) {
    __shared__ double A[6][6], inv[6][6], x[6], sc[6];
    __shared__ bool singular;
    __shared__ float T[16], F[16];
    const int t = threadIdx.x;
    const float F_prev = pv.F;
    const int iters_prev = pv.iters;
    const double mse_prev = pv.mse;
    KT_STAMP(first ? 6 : 22);
    if (t < 36) {           // symmetric fill from the 21 upper-triangle sums
        const int i = t / 6, j = t % 6, r = min(i, j), c = max(i, j);
        A[i][j] = sums[r * 6 - r * (r - 1) / 2 + (c - r)];
    }
    wave_sync();
    inv6_wave(A, inv, &singular);
    KT_STAMP(first ? 7 : 23);
    if (t < 6) {
        double s = 0.0;
        for (int c = 0; c < 6; ++c) s += inv[t][c] * sums[21 + c];
        x[t] = s;
    }
    wave_sync();
    if (t < 3) { double sn, cs; sincos(x[t], &sn, &cs); sc[t] = cs; sc[3 + t] = sn; }     // alpha, beta, gamma (sincos: one argument reduction, the values of sin() and cos())
    wave_sync();
    if (t == 0) {
        const double ca = sc[0], cb = sc[1], cg = sc[2], sa = sc[3], sb = sc[4], sg = sc[5];
        T[0] = (float)(cg * cb);
        T[1] = (float)(-sg * ca + cg * sb * sa);
        T[2] = (float)(sg * sa + cg * sb * ca);
        T[4] = (float)(sg * cb);
        T[5] = (float)(cg * ca + sg * sb * sa);
        T[6] = (float)(-cg * sa + sg * sb * ca);
        T[8] = (float)(-sb);
        T[9] = (float)(cb * sa);
        T[10] = (float)(cb * ca);
        T[3] = (float)x[3]; T[7] = (float)x[4]; T[11] = (float)x[5];
        T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
    }
    KT_STAMP(first ? 8 : 24);
    if (t < 16) F[t] = first ? ((t % 5 == 0) ? 1.f : 0.f) : F_prev;
    wave_sync();
    if (t < 16) {           // final = T * final (Eigen order), one element per lane
        const int i = t / 4, j = t % 4;
        float s = T[4 * i + 0] * F[0 + j];
        s = s + T[4 * i + 1] * F[4 + j];
        s = s + T[4 * i + 2] * F[8 + j];
        s = s + T[4 * i + 3] * F[12 + j];
        coh_store(&st->Tfinal[t], s);
        coh_store(&st->T[t], T[t]);
        state_words[t] = __float_as_uint(T[t]);           // the state as the mailbox message carries it (IcpState layout)
        state_words[16 + t] = __float_as_uint(s);
    }
    wave_sync();
    if (t == 0) {
    const int iters = iters_prev + 1;
    const double prev_mse = mse_prev;
    coh_store(&st->iters, iters);
    if (first) { coh_store(&st->reason, 0); coh_store(&st->pad, 0); }
    // pcl::registration::DefaultConvergenceCriteria<float>::hasConverged()
    int done = 0, reason = 0;
    const double cos_angle = 0.5 * (double)(T[0] + T[5] + T[10] - 1.0f);
    const double translation_sqr = (double)(T[3] * T[3] + T[7] * T[7] + T[11] * T[11]);
    const double mse = sums[27] / (double)ns;
    if (iters >= 100) { done = 1; reason = 1; }
    else if (cos_angle >= 1.0 - 1e-8 && translation_sqr <= 1e-8) { done = 1; reason = 2; }
    else if (fabs(mse - prev_mse) < 1e-12) { done = 1; reason = 3; }
    else if (fabs(mse - prev_mse) / prev_mse < mse_rel) { done = 1; reason = 4; }
    if (done) coh_store(&st->reason, reason);
    else coh_store(&st->prev_mse, mse);
    if (first && done) coh_store(&st->prev_mse, prev_mse);
    if (first || done) coh_store(&st->done, done);
    // (fields this tail leaves alone keep what an unconverged run holds: done = 0, reason = 0; prev_mse of the converged state
    // is the one before this iteration)
    state_words[32] = (unsigned)iters; state_words[33] = (unsigned)done; state_words[34] = (unsigned)(done ? reason : 0);
    state_words[35] = 0u;
    const double pm = done ? prev_mse : mse;
    state_words[36] = (unsigned)__double2loint(pm); state_words[37] = (unsigned)__double2hiint(pm);
    // the stage guard (stage_dev.h): the iteration's transformation is final now - does it end Stage 1 (R.cpp:881-894)?
    state_words[kStateWords] = 0u; state_words[kStateWords + 1] = 0u;
    if (sg.out && done) {
        float mn[3], mx[3], Tf[16];
        for (int d = 0; d < 3; ++d) { mn[d] = pw_ord2f(sg.bbox6[d]); mx[d] = pw_ord2f(sg.bbox6[3 + d]); }
        for (int e = 0; e < 16; ++e) Tf[e] = __uint_as_float(state_words[16 + e]);
        double bb[6];
        pw_octree_bbox(mn, mx, sg.resolution, bb);
        const float maxBB = pw_bb_corner_change(bb, Tf);
        const unsigned flag = (maxBB < sg.DTmin) ? 1u : 0u;
        coh_store(&sg.out[0], __float_as_uint(maxBB));
        coh_store(&sg.out[1], flag);
        state_words[kStateWords] = __float_as_uint(maxBB); state_words[kStateWords + 1] = flag;
    }
    }
    wave_sync();
}

// ---- classification + compaction + inner-ICP iteration 0 in ONE launch -------------------------------------------------------
// Steps (2)-(5a) of an outer iteration (R.cpp:750-877) used to be three launches (classify, compact, first k_icp_iter); the
// chain of dependent launches is what an outer iteration costs at 10^4 patches, so they are one launch now:
//   classify   one lane per source patch (classify_dev.h): stable flag, LoD min / max into the slot;
//   compact    order-preserving: wave scan, block scan, and the block's base = the stable counts of all blocks before it, read
//              from their published aggregates (one 64-bit word per block: epoch | stable points | stable patches; a block
//              only ever waits for blocks with smaller indices, which were dispatched before it);
//   ICP it. 0  the correspondences of PCL's first inner iteration ARE the centroid matches the front launch has just found
//              (same queries - the untransformed stable centroids - against the same target-centroid grid), so the LLS row of
//              a stable patch is formed by its own lane without another search; 28 sums per block in a fixed order (xor tree
//              in the wave, waves in order), blocks in order by the block that finishes last, which also writes the totals
//              into the slot, solves the 6x6 system (icp_solve_tail, first = true) and - when no further k_icp_iter launch
//              was enqueued behind it - sends the batch's mailbox message.
// `epoch` tags the aggregates of THIS launch (a per-pair launch counter, never 0): words left by earlier launches do not match.
constexpr int kClsBlock = 256;                                // patches per block ...
constexpr int kClsThreads = kClsBlock + 64;                   // ... on waves 1-4; wave 0 is the block's service wave (see below)
constexpr int kClsSegs = 9;                                   // 28 sums x 9 segments = 252 of the block's 256 threads
constexpr int kClsSegLen = (kClsBlock + kClsSegs - 1) / kClsSegs;


__device__ __forceinline__ unsigned long long agg_pack(unsigned epoch, int n, int pts) {
    return ((unsigned long long)(epoch & 0xffffu) << 48) | ((unsigned long long)(unsigned)pts << 16) | (unsigned long long)(unsigned)n;
}

__global__ void __launch_bounds__(kClsThreads) k_classify_icp0(ClassifyArgs a, int* __restrict__ stable, float4* __restrict__ stCT,
                                                             float4* __restrict__ stN, float4* __restrict__ wsrc,
                                                             float4* __restrict__ wsrcn, int* __restrict__ wmatch, unsigned* __restrict__ slot,
                                                             unsigned long long* __restrict__ agg, unsigned epoch, IcpState* st,
                                                             double* __restrict__ partials, unsigned* __restrict__ counter,
                                                             double mse_rel, IcpMail mail, StageGuard sg) {
    __shared__ int s_n[kClsBlock / 64], s_p[kClsBlock / 64];
    __shared__ float s_lod[kClsBlock / 64][2];
    __shared__ float s_row[kClsBlock][8];
    __shared__ double s_part[kClsSegs][kNSums];
    // Wave 0 classifies nothing: it is the block's SERVICE wave (aggregate, final partial sums, block count, and on the block
    // that finishes last the totals, the solve and the mailbox message), so that nothing the other four waves still have to do
    // - the look-back for the compaction base, the compacted outputs - is between the last partial sum and the solve.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool svc = wave == 0;
    const int ptid = tid - 64, pw = wave - 1;          // patch lane / patch wave (negative on the service wave)
    const int me = blockIdx.x, nb = gridDim.x;
    const int i = svc ? a.m2 : me * kClsBlock + ptid;
    if (blockIdx.x == 0 && threadIdx.x == 0) KT_STAMP(0);
    float lod = 0.0f;
    int f = 0, np = 0;
    pwdev::ClassifyRow cr;
    if (i < a.m2) {
        np = a.off2[i + 1] - a.off2[i];
        f = pwdev::classify_patch(a, i, &lod, &cr);
        stable[i] = f;
    }
    // block aggregate: stable patches / their points, LoD min / max
    int in = f;                                      // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(in, o);
        if (lane >= o) in += up;
    }
    int pts = f ? np : 0;
    float lo = (i < a.m2) ? lod : INFINITY, hi = (i < a.m2) ? lod : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        pts += __shfl_xor(pts, o);
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if (!svc && lane == 63) s_n[pw] = in;
    if (!svc && lane == 0) { s_p[pw] = pts; s_lod[pw][0] = lo; s_lod[pw][1] = hi; }
    __syncthreads();
    KT_MAX(1);
    int blk_n = 0, blk_p = 0, wave_off = 0;
#pragma unroll
    for (int w = 0; w < kClsBlock / 64; ++w) {
        if (w < pw) wave_off += s_n[w];
        blk_n += s_n[w]; blk_p += s_p[w];
    }
    unsigned lod_r0 = 0, lod_r1 = 0;                     // return values of the LoD atomics: waited for only before the count
    if (tid == 0) {
        float blo = INFINITY, bhi = 0.0f;
        for (int w = 0; w < kClsBlock / 64; ++w) { blo = fminf(blo, s_lod[w][0]); bhi = fmaxf(bhi, s_lod[w][1]); }
        if (bhi > 0.0f) {                                // positive floats order like their bit patterns
            lod_r0 = atomicMin(&slot[0], __float_as_uint(blo));
            lod_r1 = atomicMax(&slot[1], __float_as_uint(bhi));
        }
        coh_store(&agg[me], agg_pack(epoch, blk_n, blk_p));
    }
    // The LLS rows of inner iteration 0 and their 28 sums need no position: they go first, the wait for the preceding blocks'
    // aggregates (published about now by all of them) comes after.  Row = [a b c nx ny nz | d | d2], float-valued as PCL forms
    // it; all 0 for a lane without a stable patch.
    {
        float row[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (f) {
            const float4 c = cr.q, t = cr.t, tn = cr.tn;
            const float sx = c.x, sy = c.y, sz = c.z, dx = t.x, dy = t.y, dz = t.z, nx = tn.x, ny = tn.y, nz = tn.z;
            row[0] = nz * sy - ny * sz;
            row[1] = nx * sz - nz * sx;
            row[2] = ny * sx - nx * sy;
            row[3] = nx; row[4] = ny; row[5] = nz;
            row[6] = nx * dx + ny * dy + nz * dz - nx * sx - ny * sy - nz * sz;
            row[7] = cr.dct;
        }
        if (!svc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s_row[ptid][e] = row[e];
        }
    }
    KT_MAX(14);
    __syncthreads();
    KT_MAX(11);
    // thread (k, seg): sum k of the 28 (21 upper-triangle products row by row, 6 row * d, sum of d2 - the order of k_icp_iter)
    // over the patches of segment seg, in patch order; then the segments in order: a fixed summation order
    if (tid < kNSums * kClsSegs) {
        const int k = tid % kNSums, seg = tid / kNSums;
        int p_ = 0, q_ = 0;
        if (k < 21) {
            int kk = k;
            while (kk >= 6 - p_) { kk -= 6 - p_; ++p_; }
            q_ = p_ + kk;
        } else if (k < 27) { p_ = k - 21; q_ = 6; }
        else { p_ = 7; q_ = -1; }
        double acc = 0.0;
        const int lo = seg * kClsSegLen, hi = min(lo + kClsSegLen, kClsBlock);
        // all LDS reads of the segment first (in flight together), then the sum in row order: the loop with a read, a product and
        // an add per trip paid the LDS latency 29 times
        float u_[kClsSegLen], w_[kClsSegLen];
        const int qq = max(q_, 0);
#pragma unroll
        for (int j = 0; j < kClsSegLen; ++j) {
            const int r = min(lo + j, kClsBlock - 1);
            u_[j] = s_row[r][p_];
            w_[j] = s_row[r][qq];
        }
#pragma unroll
        for (int j = 0; j < kClsSegLen; ++j) {
            if (lo + j < hi) {
                const float u = u_[j], w = w_[j];
                double v;
                if (q_ < 0) v = (double)u;                                   // d2
                // PCL's ATA / ATb: products among the normal's components are float products (widened afterwards), everything
                // that involves a, b, c or d is a double product of the widened float values
                else v = (k < 21 && p_ >= 3) ? (double)(u * w) : (double)u * (double)w;
                acc += v;
            }
        }
        s_part[seg][k] = acc;
    }
    KT_MAX(12);
    __syncthreads();
    KT_MAX(13);
    if (tid < kNSums) {
        double acc = s_part[0][tid];
#pragma unroll
        for (int g = 1; g < kClsSegs; ++g) acc += s_part[g][tid];
        coh_store(&partials[(size_t)me * kNSums + tid], acc);
    }
    // The block's share of iteration 0 is done: count it, and let the block that finishes last go on with the totals and the solve
    // at once - the positions of the compacted outputs (below) are not on the way to T.
    if (svc) {
        KT_MAX(3);
        asm volatile("" ::"v"(lod_r0), "v"(lod_r1));    // the LoD atomics PERFORMED (their values are back) before this block counts itself
        drain_stores();
        unsigned last = 0;
        if (tid == 0) {
            const unsigned prev = atomicAdd(counter, 1u);
            last = (prev == (unsigned)nb - 1u) ? 1u : 0u;
            if (last) *counter = 0u;
        }
        last = (unsigned)__shfl((int)last, 0);
        if (last) {
            KT_STAMP(4);
            // totals (every aggregate is published by now), slot words, state, solve.  The aggregates and the partials are
            // requested together: one round trip
            __shared__ double s_sums[kNSums];
            __shared__ unsigned s_state[kTailWords];
            const bool fast_mail = mail.dst && mail.na + mail.nb <= 64 && mail.na >= 4;
            bool solved = false;
            unsigned pre = fast_mail ? mail_prefetch(mail) : 0u;
            const unsigned long long agg0 = (tid < nb) ? coh_load(&agg[tid]) : 0ull;
            tail_sums_wave(partials, nb, s_sums);
            int tn = (int)(agg0 & 0xffffu), tp = (int)((agg0 >> 16) & 0xffffffffu);
            for (int b = tid + 64; b < nb; b += 64) {
                const unsigned long long v = coh_load(&agg[b]);
                tn += (int)(v & 0xffffu); tp += (int)((v >> 16) & 0xffffffffu);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { tn += __shfl_xor(tn, o); tp += __shfl_xor(tp, o); }
            if (tid == 0) { coh_store(&slot[2], (unsigned)tn); coh_store(&slot[3], (unsigned)tp); }
            if (tn < 3) {              // min_number_correspondences_ = 3: no estimate (the host stops at < 4 stable patches anyway)
                if (tid < 16) { coh_store(&st->T[tid], (tid % 5 == 0) ? 1.f : 0.f); coh_store(&st->Tfinal[tid], (tid % 5 == 0) ? 1.f : 0.f); }
                if (tid == 0) {
                    coh_store(&st->iters, 0); coh_store(&st->reason, 0); coh_store(&st->pad, 0);
                    coh_store(&st->prev_mse, 1.7976931348623157e308);
                    coh_store(&st->done, 1);
                }
            } else {
                KT_STAMP(5);
                icp_solve_tail(st, s_sums, tn, mse_rel, true, TailPrev{0.f, 0, 1.7976931348623157e308}, s_state, sg);
                solved = true;
            }
            KT_STAMP(9);
            if (solved && fast_mail) {
                if (tid == 2) pre = (unsigned)tn;           // slot words 2 / 3 are this tail's own
                if (tid == 3) pre = (unsigned)tp;
                icp_send_mail_fast(mail, pre, s_state, sg);
            } else if (mail.dst) {
                drain_stores();
                wave_sync();
                icp_send_mail(mail, st);
            }
            KT_STAMP(10);
        }
        return;
    }
    // base of the compacted outputs: the stable patches of the blocks before this one (every wave for itself: no block barrier
    // that the wave in the solve would keep the others waiting at), then the outputs (generateCentroidCloudWithPatchNormals
    // semantics for the normal: (0,0,1) unless > 6 points and valid)
    int base = 0;
    for (int b = lane; b < me; b += 64) {
        unsigned long long v = coh_load(&agg[b]);
        while ((unsigned)(v >> 48) != (epoch & 0xffffu)) { __builtin_amdgcn_s_sleep(1); v = coh_load(&agg[b]); }
        base += (int)(v & 0xffffu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) base += __shfl_xor(base, o);
    KT_MAX(2);
    if (f) {
        const float4 c = a.ct2[i];
        // (a.nrm2 == nullptr: PWICP_SOURCE_NORMALS=0, loop.hip: source_normals() - the source normals have no consumer, PCL's
        // point-to-plane estimate reads the target's; the working arrays still get a unit vector to rotate)
        float4 n = a.nrm2 ? a.nrm2[i] : make_float4(0.f, 0.f, 1.f, 0.f);
        if (!(np > 6 && n.w != 0.0f)) n = make_float4(0.f, 0.f, 1.f, 0.f);
        n.w = 0.f;
        const int pos = base + wave_off + in - 1;
        stCT[pos] = c; stN[pos] = n;
        wsrc[pos] = c; wsrcn[pos] = n;
        wmatch[pos] = max(a.mCT[i], 0);              // the stable centroid's nearest target centroid (front launch): the VCM's match
    }
}

__global__ void k_icp_init(IcpState* st) {
    if (threadIdx.x == 0) {
        for (int k = 0; k < 16; ++k) {
            st->T[k] = (k % 5 == 0) ? 1.f : 0.f;
            st->Tfinal[k] = (k % 5 == 0) ? 1.f : 0.f;
        }
        st->iters = 0; st->done = 0; st->reason = 0; st->pad = 0;
        st->prev_mse = 1.7976931348623157e308;   // std::numeric_limits<double>::max()
    }
}

// ---- VCM (R.cpp:1273-1343) ------------------------------------------------------------------------------
constexpr int kVSums = 28;    // 21 ATA + 6 ATL + L^T L

__device__ __forceinline__ void vcm_row(float4 q, float4 p, float4 n, double* a, double* L) {
    const double Qx = q.x, Qy = q.y, Qz = q.z, Px = p.x, Py = p.y, Pz = p.z, Nx = n.x, Ny = n.y, Nz = n.z;
    a[0] = Nz * Qy - Ny * Qz;
    a[1] = Nx * Qz - Nz * Qx;
    a[2] = Ny * Qx - Nx * Qy;
    a[3] = Nx; a[4] = Ny; a[5] = Nz;
    *L = Nx * (Px - Qx) + Ny * (Py - Qy) + Nz * (Pz - Qz);
}

// calTransParaVCM (R.cpp:1273-1343) in ONE launch of 1024-thread blocks, `bid` of `nb_max`:
//   every block   NN of its 128 stable centroids among the target centroids (group-cooperative like k_icp_iter: 8 lanes share a
//                 point's search and each keeps 4 of the 27 sums), one partial per block;
//   last block    (device counter) the sums of the partials on all its waves, Qxx = (A^T A)^-1 and x = Qxx A^T L on its first wave,
//                 v^T v = L^T L - x^T A^T L from the sums (the residual pass over ALL points - one CU reading 0.5 MB: 9 us -
//                 only when those two terms cancel, see below),
//                 sigma0^2 Qxx (R.cpp:1330-1340) and, when asked, the run's closing mailbox message.
// (Two launches before: the second one's start-up cost as much as its work.)  ns: *ns_dev when ns_dev != nullptr.
constexpr int kVcmBlock = kAccBlock;

__device__ __forceinline__ void vcm_block(const GridDesc& g, const float4* __restrict__ tgt, const float4* __restrict__ tgt_n,
                                          const float4* __restrict__ src, int ns, int* __restrict__ match,
                                          double* __restrict__ partials, unsigned* __restrict__ counter,
                                          double* __restrict__ vcm, const VcmMail& mail, int bid, bool have_match) {
    __shared__ double sh[kVcmBlock / 64][32];
    __shared__ double sums[kNSums];
    __shared__ double A[6][6], Q[6][6], xs[6];
    __shared__ bool singular;
    __shared__ unsigned s_last;
    const int nact = (ns + kAccPts - 1) / kAccPts;
    if (bid >= nact) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = tid % kGroup;
    const int i = bid * kAccPts + tid / kGroup;
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
    if (i < ns) {
        const float4 q = src[i];
        // have_match: the stable centroids are the ones the fused classification has just compacted, and it left their nearest
        // target centroids (the front launch's matches: same points, same target, R.cpp:1292 would find them again) in `match`
        int bi;
        if (have_match) bi = match[i];
        else {
            const NNBest b = nn_query_group(g, q.x, q.y, q.z, sub);
            bi = b.idx();
            if (sub == 0) coh_store(&match[i], bi);
        }
        double a[6], L;
        vcm_row(q, tgt[bi], tgt_n[bi], a, &L);
        // sum k (0..27; 21 upper-triangle products row by row, then a[r]*L, then L*L) lives on lane sub = k % 8 as its (k / 8)-th value
        switch (sub) {
            case 0: w0 = a[0] * a[0]; w1 = a[1] * a[3]; w2 = a[3] * a[4]; w3 = a[3] * L; break;    // 0, 8, 16, 24
            case 1: w0 = a[0] * a[1]; w1 = a[1] * a[4]; w2 = a[3] * a[5]; w3 = a[4] * L; break;    // 1, 9, 17, 25
            case 2: w0 = a[0] * a[2]; w1 = a[1] * a[5]; w2 = a[4] * a[4]; w3 = a[5] * L; break;    // 2, 10, 18, 26
            case 3: w0 = a[0] * a[3]; w1 = a[2] * a[2]; w2 = a[4] * a[5]; w3 = L * L; break;       // 3, 11, 19, 27
            case 4: w0 = a[0] * a[4]; w1 = a[2] * a[3]; w2 = a[5] * a[5]; break;                   // 4, 12, 20
            case 5: w0 = a[0] * a[5]; w1 = a[2] * a[4]; w2 = a[0] * L; break;                      // 5, 13, 21
            case 6: w0 = a[1] * a[1]; w1 = a[2] * a[5]; w2 = a[1] * L; break;                      // 6, 14, 22
            default: w0 = a[1] * a[2]; w1 = a[3] * a[3]; w2 = a[2] * L; break;                     // 7, 15, 23
        }
    }
#pragma unroll
    for (int o = kGroup; o < 64; o <<= 1) {
        w0 += __shfl_xor(w0, o); w1 += __shfl_xor(w1, o); w2 += __shfl_xor(w2, o); w3 += __shfl_xor(w3, o);
    }
    if (lane < kGroup) { sh[wave][lane] = w0; sh[wave][8 + lane] = w1; sh[wave][16 + lane] = w2; sh[wave][24 + lane] = w3; }
    __syncthreads();
    if (tid < kVSums) {
        double acc = sh[0][tid];
        for (int w = 1; w < kVcmBlock / 64; ++w) acc += sh[w][tid];
        coh_store(&partials[(size_t)bid * kNSums + tid], acc);
    }
    drain_stores();                         // every wave: its matches (and wave 0's partials) performed ...
    __syncthreads();                        // ... before the block counts itself
    if (tid == 0) {
        const unsigned prev = atomicAdd(counter, 1u);
        s_last = (prev == (unsigned)nact - 1u) ? 1u : 0u;
        if (s_last) *counter = 0u;
    }
    __syncthreads();
    if (!s_last) return;
    KT_STAMP(26);
    // ---- the block that finished last: the sums of the partials on all its waves (one round trip), the solve on wave 0 ----
    __shared__ double s_vv;
    __shared__ int s_explicit;
    // (the diagnostic counters of the closing message: requested now, they come from HBM)
    unsigned long long ex = 0;
    if (mail.dst && tid < 64)
        for (int k = tid; k < 256; k += 64) ex += mail.examined[(size_t)k * 16];
    tail_sums_block(partials, nact, sh, sums);
    KT_STAMP(27);
    if (tid < 64) {
        const int t = tid;
        if (t < 36) {
            const int r0 = t / 6, c0 = t % 6, r = min(r0, c0), c = max(r0, c0);
            A[r0][c0] = sums[r * 6 - r * (r - 1) / 2 + (c - r)];
        }
        wave_sync();
        inv6_wave(A, Q, &singular);
        if (t < 6) {
            double s = 0;
            for (int c = 0; c < 6; ++c) s += Q[t][c] * sums[21 + c];
            xs[t] = s;
        }
        wave_sync();
        if (t == 0) {
            // v^T v with v = A x - L from the sums of the first pass - no second pass over the points:
            //     v^T v = L^T L - 2 x^T (A^T L) + x^T (A^T A) x.
            // The quadratic form, not its simplification L^T L - x^T A^T L (valid at the exact solution only): its gradient in x
            // vanishes at the least-squares solution, so an error of x - which grows with cond(A^T A) - enters in SECOND order and
            // what is left is the rounding of the sums, ~1e-16 L^T L.  The terms cancel when the fit explains nearly all of L;
            // below 1e-3 L^T L (relative error of v^T v <= ~1e-13 above it; never at the end of a registration, where x ~ 0) the
            // residuals are formed point by point below, as the reference does (R.cpp:1331-1333).
            double xtl = 0.0, xax = 0.0;
            for (int c = 0; c < 6; ++c) xtl += xs[c] * sums[21 + c];
            for (int r = 0; r < 6; ++r) {
                double row = 0.0;
                for (int c = 0; c < 6; ++c) row += A[r][c] * xs[c];
                xax += xs[r] * row;
            }
            const double vv_id = (sums[27] - 2.0 * xtl) + xax;
            s_vv = vv_id;
            s_explicit = (vv_id > 1e-3 * sums[27]) ? 0 : 1;
#ifdef PWICP_KTRACE
            pw_ktrace[20] = (unsigned long long)s_explicit; pw_ktrace[21] = (unsigned long long)__double_as_longlong(vv_id / sums[27]);
#endif
        }
        KT_STAMP(28);
    }
    __syncthreads();
    double vv = 0.0;
    if (s_explicit) {
    // ---- residuals of all points on the whole block: thread t takes points t, t + 1024, ... ; fixed summation order ----
    for (int p = tid; p < ns; p += kVcmBlock) {
        const int bi = coh_load(&match[p]);
        double a[6], L;
        vcm_row(src[p], tgt[bi], tgt_n[bi], a, &L);
        double r = 0;
        for (int c = 0; c < 6; ++c) r += a[c] * xs[c];
        r -= L;
        vv += r * r;
    }
    }
    KT_STAMP(29);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vv += __shfl_xor(vv, o);
    if (lane == 0) sh[wave][0] = vv;
    __syncthreads();
    if (tid >= 64) return;
    double vtpv = sh[0][0];
    for (int w = 1; w < kVcmBlock / 64; ++w) vtpv += sh[w][0];
    if (!s_explicit) vtpv = s_vv;
    const int t = tid;
    const double STD0 = sqrt(vtpv / (double)(ns - 6));
    double out = 0.0;
    if (t < 36) { out = STD0 * STD0 * Q[t / 6][t % 6]; vcm[t] = out; }
    KT_STAMP(30);
    if (!mail.dst) return;
    // message: 36 doubles | the folded diagnostic counter (256 partial counters, 128 bytes apart) | seq
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ex += __shfl_xor(ex, o);
    if (t < 36) {
        unsigned lo, hi;
        memcpy(&lo, &out, 4);
        memcpy(&hi, (const char*)&out + 4, 4);
        mail_store(&mail.dst[2 * t], lo);
        mail_store(&mail.dst[2 * t + 1], hi);
    }
    if (t == 0) {
        mail_store(&mail.dst[72], (unsigned)(ex & 0xffffffffull));
        mail_store(&mail.dst[73], (unsigned)(ex >> 32));
    }
    mail_drain();
    wave_sync();
    if (t == 0) mail_publish(mail.seq_ptr, mail.seq);
    KT_STAMP(31);
}

__global__ void __launch_bounds__(kVcmBlock) k_vcm(GridDesc g, const float4* __restrict__ tgt, const float4* __restrict__ tgt_n,
                                                   const float4* __restrict__ src, int ns, int* __restrict__ match,
                                                   double* __restrict__ partials, unsigned* __restrict__ counter,
                                                   double* __restrict__ vcm, VcmMail mail, int have_match) {
    vcm_block(g, tgt, tgt_n, src, ns, match, partials, counter, vcm, mail, (int)blockIdx.x, have_match != 0);
}

// The LAST update of a run (R.cpp:943-954) and calTransParaVCM (R.cpp:958-961) in one launch: the VCM works on the stable
// centroids as they were BEFORE the update (R.cpp:868) and on the static target, so it does not wait for the transform - the
// two used to be 14 + 16 + 8 us back to back.  Blocks [0, nb_vcm) are the VCM, the rest the transform (guarded by the ICP's done
// flag like k_transform_all).  `stage3_bits` != 0: the launch was enqueued BEFORE the host has seen this iteration's result, on
// the guess that it is the last one; the VCM then only runs if the iteration really reaches Stage 3, which in Stage 2 is
// exactly `currDT == LoDet_min` (R.cpp:896) - both are known here (slot word 0).
__global__ void __launch_bounds__(kVcmBlock) k_xf_vcm(GridDesc g, const float4* __restrict__ tgt, const float4* __restrict__ tgt_n,
                                                      const float4* __restrict__ stct, int* __restrict__ match,
                                                      double* __restrict__ partials, unsigned* __restrict__ counter,
                                                      double* __restrict__ vcm, VcmMail mail, int nb_vcm, unsigned stage3_bits,
                                                      const float4* cloud_in, const float4* ctbp_in, const float4* pat_in,
                                                      float4* cloud, int n, int nb_cloud, float4* ctbp, int n_ctbp, float4* pat,
                                                      int n_pat, const IcpState* __restrict__ st, const unsigned* __restrict__ slot_ro,
                                                      unsigned* __restrict__ bbox_part, unsigned* __restrict__ slot, int nb_rest) {
    __shared__ float shb[kVcmBlock / 64][6];
    const int done_in = st->done;               // (flag, count, stage word and T requested together)
    const unsigned ns_in = slot_ro[2];
    int bid = (int)blockIdx.x;
    const unsigned stage_in = (bid < nb_vcm && stage3_bits) ? coh_load(&slot_ro[0]) : 0u;
    Mat4 T;
#pragma unroll
    for (int i = 0; i < 16; ++i) T.m[i] = st->Tfinal[i];
    if (!done_in || ns_in < 4u) return;
    if (bid < nb_vcm) {
        if (stage3_bits && stage_in != stage3_bits) return;
        VT_BEGIN(0);
        vcm_block(g, tgt, tgt_n, stct, (int)ns_in, match, partials, counter, vcm, mail, bid, true);
        VT_END();
        return;
    }
    bid -= nb_vcm;
    if (bid < nb_rest) { VT_BEGIN(1); xf_rest_block<kVcmBlock>(T, ctbp_in, ctbp, n_ctbp, pat_in, pat, n_pat, bid, nb_rest); VT_END(); return; }
    VT_BEGIN(2);
    xf_cloud_block<kVcmBlock>(T, cloud_in, cloud, n, bid - nb_rest, nb_cloud, bbox_part, slot, shb);
    VT_END();
}

}  // namespace

// ==========================================================================================================
int IcpWork::reserve(pwicp_context* ctx, int ns_max) {
    int nb = std::max(div_up(std::max(ns_max, 1), kBlock), div_up(std::max(ns_max, 1), kAccPts));
    HIPCHK(ctx, src.reserve((size_t)std::max(ns_max, 1)));
    HIPCHK(ctx, srcn.reserve((size_t)std::max(ns_max, 1)));
    HIPCHK(ctx, match.reserve((size_t)std::max(ns_max, 1)));
    HIPCHK(ctx, partials.reserve((size_t)nb * kNSums));
    HIPCHK(ctx, state.reserve(1));
    HIPCHK(ctx, counter.reserve(1));
    HIPCHK(ctx, hipMemsetAsync(counter.p, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(ctx, qx.reserve(48));
    HIPCHK(ctx, vcm.reserve(36));
    HIPCHK(ctx, agg.reserve((size_t)div_up(std::max(ns_max, 1), kClsBlock) + 1));
    HIPCHK(ctx, hipMemsetAsync(agg.p, 0, sizeof(unsigned long long) * ((size_t)div_up(std::max(ns_max, 1), kClsBlock) + 1), ctx->stream));
    return PWICP_OK;
}

#ifdef PWICP_KTRACE
extern "C" __attribute__((visibility("default"))) int pwicp_debug_ktrace(unsigned long long* out32, int reset) {
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(pw_ktrace), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32];
        for (int i = 0; i < 32; ++i) z[i] = 0ull;
        z[0] = ~0ull;
        if (hipMemcpyToSymbol(HIP_SYMBOL(pw_ktrace), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// classification + compaction + inner-ICP iteration 0 (k_classify_icp0); `mail`: sent by this launch (no k_icp_iter follows
// in the batch)
int pw_classify_icp0_launch(pwicp_context* ctx, const ClassifyArgs& a, int* d_stable, float4* d_stCT, float4* d_stN, IcpWork* w,
                            unsigned* d_slot, double euclid_eps, const IcpMail* mail, const StageGuard* sg) {
    if (a.m2 <= 0) return PWICP_OK;
    w->epoch = (w->epoch % 0xffffu) + 1u;          // 1 .. 65535, never 0 (the buffer is zeroed once)
    IcpMail none{};
    hipLaunchKernelGGL(k_classify_icp0, dim3(div_up(a.m2, kClsBlock)), dim3(kClsThreads), 0, ctx->stream, a, d_stable, d_stCT, d_stN,
                       w->src.p, w->srcn.p, w->match.p, d_slot, w->agg.p, w->epoch, w->state.p, w->partials.p, w->counter.p, euclid_eps,
                       mail ? *mail : none, sg ? *sg : StageGuard{});
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

// Enqueues n_iter inner iterations (accumulate + solve each) on the stream; no host synchronisation.  The number
// of source points is ns_host, or *ns_dev when ns_dev != nullptr (then ns_max bounds the launch grid).
int pw_icp_enqueue(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w,
                   int ns_max, const unsigned* ns_dev, double euclid_eps, int n_iter, const IcpMail* mail, const FusedSelect* fs,
                   const StageGuard* sg) {
    if (ns_max <= 0) return PWICP_OK;
    const int nb = div_up(ns_max, kAccPts);
    IcpMail none{};
    FusedSelect nofs{};
    for (int k = 0; k < n_iter; ++k) {
        const bool sel = fs && fs->scratch && k < 2;           // passes 1 and 2 on the first two launches (the caller enqueues >= 2)
        hipLaunchKernelGGL(k_icp_iter, dim3(nb + (sel ? fs->nblk : 0)), dim3(kAccBlock), 0, ctx->stream, g, d_tgt, d_tgt_n, w->src.p,
                           w->srcn.p, ns_max, ns_dev, w->state.p, w->partials.p, w->counter.p, euclid_eps,
                           (mail && k == n_iter - 1) ? *mail : none, sel ? *fs : nofs, k + 1, sg ? *sg : StageGuard{});
    }
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

// d_src / d_srcn: working copies (modified in place). Returns final T and the iteration count.
int pw_icp_run(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w, int ns,
               double euclid_eps, float* T16, int* iters_out) {
    for (int k = 0; k < 16; ++k) T16[k] = (k % 5 == 0) ? 1.f : 0.f;
    if (iters_out) *iters_out = 0;
    if (ns < 3 || g.fine.n <= 0) return PWICP_OK;      // min_number_correspondences_ = 3: no update
    hipLaunchKernelGGL(k_icp_init, dim3(1), dim3(64), 0, ctx->stream, w->state.p);
    IcpState h;
    for (int done_iters = 0; done_iters < 100;) {
        PWCHK(pw_icp_enqueue(ctx, g, d_tgt, d_tgt_n, w, ns, nullptr, euclid_eps, 4, nullptr));
        HIPCHK(ctx, hipMemcpyAsync(&h, w->state.p, sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        done_iters = h.iters;
        if (h.done) break;
    }
    HIPCHK(ctx, hipGetLastError());
    memcpy(T16, h.Tfinal, sizeof(h.Tfinal));
    if (iters_out) *iters_out = h.iters;
    return PWICP_OK;
}

// enqueue only: the 6x6 result is left in w->vcm (device)
int pw_vcm_enqueue(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w,
                   const float4* d_src, int ns, const VcmMail* mail, bool have_match) {
    if (ns <= 0) return PWICP_OK;
    VcmMail none{};
    hipLaunchKernelGGL(k_vcm, dim3(div_up(ns, kAccPts)), dim3(kVcmBlock), 0, ctx->stream, g, d_tgt, d_tgt_n, d_src, ns, w->match.p,
                       w->partials.p, w->counter.p, w->vcm.p, mail ? *mail : none, have_match ? 1 : 0);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

// the run's last transform + the VCM in one launch (k_xf_vcm); ns_max bounds the VCM's grid, the count itself is slot word 2
int pw_xf_vcm_launch(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w,
                     const float4* d_stct, int ns_max, const VcmMail* mail, unsigned stage3_bits, const float4* d_cloud_in,
                     const float4* d_ctbp_in, const float4* d_pat_in, float4* d_cloud, int n, float4* d_ctbp, int n_ctbp, float4* d_pat,
                     int n_pat, unsigned* d_bbox_part, unsigned* d_slot) {
    const int nb_vcm = div_up(std::max(ns_max, 1), kAccPts);
    // (a cloud block ends with the bounding-box fold, ~3 us of dependent atomics: one block per CU, 4 k points each, not two)
    const int nb_cloud = std::min(div_up(n, kVcmBlock), ctx->n_cu);
    const int nb_rest = std::min(div_up(n_ctbp + n_pat, kVcmBlock), ctx->n_cu * 2);
    VcmMail none{};
    hipLaunchKernelGGL(k_xf_vcm, dim3(nb_vcm + nb_rest + nb_cloud), dim3(kVcmBlock), 0, ctx->stream, g, d_tgt, d_tgt_n, d_stct,
                       w->match.p, w->partials.p, w->counter.p, w->vcm.p, mail ? *mail : none, nb_vcm, stage3_bits, d_cloud_in,
                       d_ctbp_in, d_pat_in, d_cloud, n, nb_cloud, d_ctbp, n_ctbp, d_pat, n_pat, (const IcpState*)w->state.p,
                       (const unsigned*)d_slot, d_bbox_part, d_slot, nb_rest);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_vcm_run(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w,
               const float4* d_src, int ns, double* VCM36) {
    if (ns <= 0) { for (int i = 0; i < 36; ++i) VCM36[i] = NAN; return PWICP_OK; }
    PWCHK(pw_vcm_enqueue(ctx, g, d_tgt, d_tgt_n, w, d_src, ns, nullptr));
    HIPCHK(ctx, hipMemcpyAsync(VCM36, w->vcm.p, 36 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}
