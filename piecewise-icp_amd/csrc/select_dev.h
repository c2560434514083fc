// k-th smallest of the dense 1-NN distances (the percentile of calPercentileDistBetween2PC, C.cpp:145-179, 266-281) as a
// 3-pass radix select on the float bit patterns (11 + 11 + 10 bits) whose passes ride on kernels that run anyway:
//   pass 0  in the epilogue of the dense 1-NN kernel itself (every block histograms the d2 values it has just produced and
//           adds its bins to global replicas with fire-and-forget atomics),
//   pass 1  on the FIRST few blocks of the transform launch (k_transform_all) — first, so that they are dispatched at once
//           and run beside the transform instead of behind it; every block picks the bin of pass 0 for itself,
//   pass 2  on the first few blocks of the next iteration's front launch (k_front), which also sends the selected value to
//           the host mailbox.
// Passes 1 / 2 end with "the block that finishes last picks the bin" (device counter): no launch of its own for any pass
// (three launches of ~15 us each before).
#pragma once

#include <hip/hip_runtime.h>

#include "common.h"

constexpr int kFsBins = 2048;
constexpr int kFsRep = 8;                 // replicas of the pass-0 bins in global memory (same-address atomics serialise)
constexpr int kFsCtl = 16;                // control words: [0] prefix, [1] rank remaining, [2..4] finished blocks of pass 0..2,
                                          // [5] tag of the selection whose pass 1 has finished (passes 1 and 2 in ONE launch)
constexpr int kFsWords = kFsCtl + (kFsRep + 2) * kFsBins;
constexpr int kFsBlocks = 128;            // blocks devoted to an embedded pass

struct FusedSelect {
    unsigned* scratch = nullptr;          // kFsWords words, zeroed once at allocation (every selection leaves it zeroed); nullptr: off
    const float* vals = nullptr;          // the dense kernel's output (sentinel 0xffffffff in unused slots)
    int n = 0;                            // slots
    int k = 0;                            // rank (0-based) among the non-sentinel values ...
    const unsigned* n_valid_dev = nullptr;   // ... or, when set, int(*n_valid_dev * 0.75f) (C.cpp:177) evaluated on the device: the
                                          // number of values is then not known to the host when the launch is enqueued
    int nblk = 0;                         // blocks running an embedded pass
    float* out = nullptr;                 // selected value (device)
    SelectMail mail{};                    // sent by pass 2
};

// bin -> word of a pass-0 replica: neighbouring bins (the distances crowd a dozen of them) land on different 128-byte lines,
// because same-line atomics serialise
__device__ __forceinline__ int fs_word(int bin) { return (bin & 63) * 32 + (bin >> 6); }

// ---- pass 0, called by EVERY block of the dense kernel once its lanes have added their values to the LDS bins `h` ----------
// Fire-and-forget atomics into one of kFsRep replicas: no completion count, no pick here — a block of the dense kernel
// retires without waiting for anything; the kernel boundary orders the bins before pass 1, whose blocks pick for themselves.
// nbins: the bins `h` holds (squared distances are >= +0: their bit patterns end at bin 1023, so a kernel that only ever adds such
// values keeps half the array - kFsPosBins words)
constexpr int kFsPosBins = kFsBins / 2;
__device__ __forceinline__ void fs_pass0_epilogue(unsigned* h, const FusedSelect& fs, int nbins = kFsBins) {
    __syncthreads();
    unsigned* g0 = fs.scratch + kFsCtl + (int)(blockIdx.x & (kFsRep - 1)) * kFsBins;
    for (int t = threadIdx.x; t < nbins; t += blockDim.x) {
        const unsigned v = h[t];
        if (v) atomicAdd(&g0[fs_word(t)], v);
    }
}

// pick of pass 0 from the replicas (every block of pass 1 does it for itself: 64 loads per thread, blockDim.x == 256):
// returns the prefix (bin << 21) and the rank remaining inside that bin
__device__ __forceinline__ void fs_pick0(unsigned* h, const FusedSelect& fs, unsigned* prefix_out, unsigned* krem_out) {
    __shared__ unsigned s_w[4], s_res[2];
    const unsigned* g0 = fs.scratch + kFsCtl;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // replicas summed word by word (coalesced), scattered to bin order through the LDS bins `h` (zeroed again below)
#pragma unroll
    for (int j = 0; j < kFsBins / 256; ++j) {
        const int w = t + 256 * j;
        unsigned s = 0;
#pragma unroll
        for (int r = 0; r < kFsRep; ++r) s += g0[r * kFsBins + w];
        h[(w & 31) * 64 + (w >> 5)] = s;               // inverse of fs_word
    }
    __syncthreads();
    unsigned c[8], local = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        c[b] = h[t * 8 + b];
        local += c[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 8; ++b) h[t * 8 + b] = 0u;
    unsigned incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += s_w[w];
    const unsigned excl = base + incl - local;
    unsigned k = (unsigned)fs.k;
    if (fs.n_valid_dev) {
        const int nv = (int)*fs.n_valid_dev;
        int kk = (int)((float)nv * 0.75f);
        if (kk >= nv) kk = nv - 1;
        k = (unsigned)max(kk, 0);
    }
    if (k >= excl && k < excl + local) {
        unsigned run = excl;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (k >= run && k < run + c[b]) { s_res[0] = (unsigned)(t * 8 + b) << 21; s_res[1] = k - run; }
            run += c[b];
        }
    }
    __syncthreads();
    *prefix_out = s_res[0];
    *krem_out = s_res[1];
}

// What the values histogrammed SO FAR say about the percentile (k_nn_dense_far, whose launch starts when the dense kernel's own values
// are all in the bins): if at least k + 1 of them lie in bins <= b, the k-th smallest of ALL values - the ones still to come only add
// to the counts - is below the upper edge of bin b.  Returns that edge as float bits ((b + 1) << 21), ~0u when the bins do not decide
// it yet.  Read-only on the global bins; `h` (kFsBins words of LDS) is scratch and left zeroed; blockDim.x == 256.
// The edge must be the bit pattern of a finite positive float: bin b holds the values whose bits >> 21 == b, the last finite one
// is bin 1019 (+inf = 0x7F800000 opens bin 1020), and (b + 1) << 21 with b = 1023 would be 0x80000000 = -0.0f - "above" which every
// far query lies.  Bins >= 1020 are only reachable with inf / NaN distances (coordinates are checked finite at upload); they decide
// nothing.  NOTE for every consumer of the dense distances: a value written for a query that was cut short at the edge is an
// UPPER BOUND of its distance, good for the rank statistics below the edge (the percentile, R.cpp:905) and for nothing else.
constexpr unsigned kFsLastFiniteEdge = 1020u;
static_assert((kFsLastFiniteEdge << 21) == 0x7F800000u, "the last edge the selection may report is +inf");
__device__ __forceinline__ unsigned fs_partial_edge(unsigned* h, const FusedSelect& fs) {
    __shared__ unsigned s_w[4], s_edge;
    const unsigned* g0 = fs.scratch + kFsCtl;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_edge = ~0u;
#pragma unroll
    for (int j = 0; j < kFsBins / 256; ++j) {
        const int w = t + 256 * j;
        unsigned s = 0;
#pragma unroll
        for (int r = 0; r < kFsRep; ++r) s += __hip_atomic_load(&g0[r * kFsBins + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h[(w & 31) * 64 + (w >> 5)] = s;               // inverse of fs_word
    }
    __syncthreads();
    unsigned c[8], local = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        c[b] = h[t * 8 + b];
        local += c[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 8; ++b) h[t * 8 + b] = 0u;
    unsigned incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += s_w[w];
    const unsigned excl = base + incl - local;
    unsigned k = (unsigned)fs.k;
    if (fs.n_valid_dev) {
        const int nv = (int)*fs.n_valid_dev;
        int kk = (int)((float)nv * 0.75f);
        if (kk >= nv) kk = nv - 1;
        k = (unsigned)max(kk, 0);
    }
    if (k >= excl && k < excl + local) {
        unsigned run = excl;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (k >= run && k < run + c[b]) { const unsigned bin = (unsigned)(t * 8 + b); s_edge = bin + 1u <= kFsLastFiniteEdge ? (bin + 1u) << 21 : ~0u; }
            run += c[b];
        }
    }
    __syncthreads();
    return s_edge;
}

// ---- passes 1 and 2 on `fs.nblk` blocks of some other launch; `bidx` = index of this block among them --------------------
// chain != 0: passes 1 and 2 run in the SAME launch (k_xf_front: pass 1 on the first blocks of the grid, pass 2 on the next).
// The wave that picks pass 1's bin publishes prefix / rank through device-coherent stores and then the tag `chain`; the blocks
// of pass 2 wait for the tag (one lane polls, the block sleeps at a barrier).  They only ever wait for blocks with smaller
// indices, which were dispatched before them.
// nt: threads of the block that take part (256 for pass 1, whose pick is written for 256 threads; the caller has retired the
// others of a larger block before the call: a barrier only waits for waves that are still alive).  0: blockDim.x.
template <int PASS>
__device__ __forceinline__ void fs_pass_embedded(unsigned* h, const FusedSelect& fs, int bidx, unsigned chain = 0u, int nt_ = 0) {
    __shared__ unsigned s_last;
    const int nt = nt_ > 0 ? nt_ : (int)blockDim.x;
    unsigned prefix, k_in;
    if (PASS == 1) {
        fs_pick0(h, fs, &prefix, &k_in);               // from the bins the dense kernel left; leaves h zeroed
    } else {
        for (int t = threadIdx.x; t < kFsBins; t += nt) h[t] = 0u;
        if (chain) {
            if (threadIdx.x == 0)
                while (__hip_atomic_load(&fs.scratch[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != chain) __builtin_amdgcn_s_sleep(2);
            __syncthreads();
            prefix = __hip_atomic_load(&fs.scratch[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            k_in = __hip_atomic_load(&fs.scratch[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            prefix = fs.scratch[0]; k_in = fs.scratch[1];  // written by pass 1 (previous launch)
        }
    }
    __syncthreads();
    const int stride = fs.nblk * nt;
    // (the pass is a chain of memory round trips, not bandwidth: eight loads in flight per lane)
    for (int i0 = bidx * nt + (int)threadIdx.x; i0 < fs.n; i0 += 8 * stride) {
        unsigned u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = (i0 + j * stride < fs.n) ? __float_as_uint(fs.vals[i0 + j * stride]) : 0xffffffffu;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (u[j] == 0xffffffffu) continue;             // slot of a query that was not part of the launch
            if (PASS == 1) {
                if ((u[j] >> 21) == (prefix >> 21)) atomicAdd(&h[(u[j] >> 10) & 2047u], 1u);
            } else {
                if ((u[j] >> 10) == (prefix >> 10)) atomicAdd(&h[u[j] & 1023u], 1u);
            }
        }
    }
    __syncthreads();
    unsigned* gh = fs.scratch + kFsCtl + (kFsRep + PASS - 1) * kFsBins;
    unsigned seen = 0;
    for (int t = threadIdx.x; t < kFsBins; t += nt) {
        const unsigned v = h[t];
        if (v) seen |= atomicAdd(&gh[t], v);
    }
    asm volatile("" ::"v"(seen));                        // performed before the block is counted (see pass 0)
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&fs.scratch[2 + PASS], 1u) == (unsigned)fs.nblk - 1u) ? 1u : 0u;
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    // ---- pick on one wave: lane owns 32 consecutive bins ----
    const int lane = threadIdx.x;
    unsigned cnt[32], local = 0;
#pragma unroll
    for (int b = 0; b < 32; ++b) {
        cnt[b] = __hip_atomic_load(&gh[lane * 32 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        local += cnt[b];
    }
    unsigned incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    const unsigned excl = incl - local;
    const unsigned k = k_in;
    const bool mine = (k >= excl) && (k < incl);
    unsigned value = 0;
    if (mine) {
        unsigned run = excl;
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const unsigned c = cnt[b];
            if (k >= run && k < run + c) {
                const unsigned bin = (unsigned)(lane * 32 + b);
                const unsigned pf = (PASS == 1) ? (prefix | (bin << 10)) : (prefix | bin);
                value = pf;
                __hip_atomic_store(&fs.scratch[0], (PASS == 2) ? 0u : pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&fs.scratch[1], (PASS == 2) ? 0u : k - run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (PASS == 2 && fs.out) fs.out[0] = __uint_as_float(pf);
            }
            run += c;
        }
    }
#pragma unroll
    for (int b = 0; b < 32; ++b) gh[lane * 32 + b] = 0u;
    if (PASS == 1) {                                   // every block of this pass has read the pass-0 replicas: re-arm them
        unsigned* g0 = fs.scratch + kFsCtl;
        for (int w = lane; w < kFsRep * kFsBins; w += 64) g0[w] = 0u;
    }
    if (lane == 0) fs.scratch[2 + PASS] = 0u;
    if (PASS == 1 && chain) {                          // prefix / rank performed (write-through stores drained), then the tag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(&fs.scratch[5], chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PASS == 2 && fs.mail.dst) {
        const unsigned long long m = __ballot(mine);
        const int src = m ? (int)__ffsll((long long)m) - 1 : 0;
        value = (unsigned)__shfl((int)value, src);
        if (lane == 0) {
            mail_store(&fs.mail.dst[0], value);
            mail_drain();
            mail_publish(fs.mail.seq_ptr, fs.mail.seq);
        }
    }
}
