// Uniform-grid build + exact 1-NN search kernels (gfx950, wave64).
//
// Replaces the FLANN KD-tree behind pcl::registration::CorrespondenceEstimation::
// determineCorrespondences (reference call sites src/Registration.cpp:737-747, 1293-1297,
// 597-601, src/CommonFunc.cpp:269-273) and the per-inner-iteration search inside
// pcl::IterativeClosestPoint (Registration.cpp:1259-1267).
//
// Exactness: the result is the argmin over ALL target points of the float expression
// ((dx*dx) + dy*dy) + dz*dz (flann::L2_Simple<float>), ties to the lowest index.  The search
// scans the (2r+1)^3 block of cells around the query and stops only when the best float d2 is
// provably smaller than that of every point outside the block (conservative bound incl. the
// rounding slack of the cell assignment); otherwise it grows the block shell by shell.
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>

#include <hipcub/hipcub.hpp>

#include "common.h"
#include "nn_device.h"
#include "select_dev.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(unsigned u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}

// ---- bounding box ------------------------------------------------------------------------------
// out[0..2] = min (ordered-uint encoded), out[3..5] = max, out[6] = 1 if any coordinate is not finite
__global__ void __launch_bounds__(kBlock) k_bbox(const float4* __restrict__ p, int n, unsigned* __restrict__ out) {
    __shared__ float sh[kBlock / 64][6];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    bool bad = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 v = p[i];
        mn[0] = fminf(mn[0], v.x); mx[0] = fmaxf(mx[0], v.x);
        mn[1] = fminf(mn[1], v.y); mx[1] = fmaxf(mx[1], v.y);
        mn[2] = fminf(mn[2], v.z); mx[2] = fmaxf(mx[2], v.z);
        bad = bad || !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z));     // fminf/fmaxf silently drop NaN
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(&out[6], 1u);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], o));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o));
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int d = 0; d < 3; ++d) { sh[wave][d] = mn[d]; sh[wave][3 + d] = mx[d]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float a = sh[0][threadIdx.x], b = sh[0][3 + threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) { a = fminf(a, sh[w][threadIdx.x]); b = fmaxf(b, sh[w][3 + threadIdx.x]); }
        atomicMin(&out[threadIdx.x], f2ord(a));
        atomicMax(&out[3 + threadIdx.x], f2ord(b));
    }
}

// ---- counting sort by cell -----------------------------------------------------------------------
__global__ void k_cell_count(const float4* __restrict__ p, int n, GridLevel g, int* __restrict__ cnt,
                             int* __restrict__ cell_id, int* __restrict__ rank) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = p[i];
    int cx = min(max(cell_of(v.x, g.ox, g.inv_h), 0), g.nx - 1);
    int cy = min(max(cell_of(v.y, g.oy, g.inv_hy), 0), g.ny - 1);
    int cz = min(max(cell_of(v.z, g.oz, g.inv_hz), 0), g.nz - 1);
    int c = (cz * g.ny + cy) * g.nx + cx;
    cell_id[i] = c;
    rank[i] = atomicAdd(&cnt[c], 1);
}

__global__ void k_cell_scatter(const float4* __restrict__ p, int n, const int* __restrict__ cell_start,
                               const int* __restrict__ cell_id, const int* __restrict__ rank,
                               float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = p[i];
    v.w = __int_as_float(i);
    out[cell_start[cell_id[i]] + rank[i]] = v;
}

// ---- exclusive scan (reduce-then-scan, 4096 elements per block) ----------------------------------
constexpr int kScanTile = 4096;   // 256 threads x 16

__global__ void k_scan_tile_sums(const int* __restrict__ d, long long n, int* __restrict__ sums) {
    __shared__ int ws[4];
    long long base = (long long)blockIdx.x * kScanTile;
    int s = 0;
    for (int k = 0; k < 16; ++k) {
        long long i = base + (long long)threadIdx.x * 16 + k;
        if (i < n) s += d[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// single block: exclusive scan of m tile sums (m is small: n/4096)
__global__ void k_scan_sums(int* __restrict__ sums, int m) {
    __shared__ int sh[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < m; base += 1024) {
        int i = base + threadIdx.x;
        int v = (i < m) ? sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int t = (threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        int incl = sh[threadIdx.x];
        int c = carry;
        if (i < m) sums[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
}

__global__ void k_scan_apply(int* __restrict__ d, long long n, const int* __restrict__ sums) {
    __shared__ int sh[256];
    long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * 16;
    int v[16];
    int s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        long long i = base + k;
        v[k] = (i < n) ? d[i] : 0;
        s += v[k];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int t = (threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int run = sums[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        long long i = base + k;
        if (i < n) d[i] = run;
        run += v[k];
    }
}

// Stage 1 for every query, then the unresolved ones are compacted inside the block (ballot + LDS) so that the
// far path (stages 2/3) runs on densely packed waves.  Shared by k_nn_points and k_nn_dense_direct.
// `slot` = output index of this lane's query (< 0: lane inactive).
__device__ __forceinline__ void nn_block_search(const GridDesc& gd, bool active, float4 q, int slot,
                                                int* __restrict__ idx_out, float* __restrict__ d2_out,
                                                unsigned& cnt) {
    __shared__ float4 s_q[kBlock];
    __shared__ unsigned long long s_key[kBlock];
    __shared__ int s_slot[kBlock];
    __shared__ int s_wcnt[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool unresolved = false;
    NNBest b;
    b.key = kKeyInit;
    if (active) {
        if (gd.fine.n > 0 && !nn_stage1(gd, q.x, q.y, q.z, b, cnt)) unresolved = true;
        else {
            if (idx_out) idx_out[slot] = b.found() ? b.idx() : -1;
            d2_out[slot] = b.d2();
        }
    }
    const unsigned long long mask = __ballot(unresolved);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_wcnt[wave] = __popcll(mask);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        if (w < wave) base += s_wcnt[w];
        total += s_wcnt[w];
    }
    if (unresolved) {
        s_q[base + before] = q;
        s_key[base + before] = b.key;
        s_slot[base + before] = slot;
    }
    __syncthreads();
    if (total <= kBlock / 8) {
        // few far queries in this block (the tail case): 8 lanes share one query so that its long scan is not a
        // single lane's serial chain.  Only the data-independent ball-box path is shared; the rest stays per lane.
        const int qi = tid >> 3, sub = tid & 7;
        if (qi < total) {
            const float4 u = s_q[qi];
            NNBest c;
            c.key = s_key[qi];
            const GridLevel& cl = gd.coarse;
            bool coop = false;
            int x0 = 0, x1 = -1, y0 = 0, y1 = -1, z0 = 0, z1 = -1;
            if (c.found()) {
                const float rho = sqrtf(c.d2()) * 1.00001f + 2.0f * cl.slack;
                x0 = max(cell_of(u.x - rho, cl.ox, cl.inv_h), 0); x1 = min(cell_of(u.x + rho, cl.ox, cl.inv_h), cl.nx - 1);
                y0 = max(cell_of(u.y - rho, cl.oy, cl.inv_hy), 0); y1 = min(cell_of(u.y + rho, cl.oy, cl.inv_hy), cl.ny - 1);
                z0 = max(cell_of(u.z - rho, cl.oz, cl.inv_hz), 0); z1 = min(cell_of(u.z + rho, cl.oz, cl.inv_hz), cl.nz - 1);
                coop = (y1 - y0 + 1) * (z1 - z0 + 1) <= 64;
            }
            if (coop) {
                if (x0 <= x1 && y0 <= y1 && z0 <= z1)
                    cnt += scan_box_coop(cl, x0, x1, y0, y1, z0, z1, sub, 8, u.x, u.y, u.z, c);
                // min over the 8 lanes of the group (lanes 8k..8k+7 of one wave)
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) {
                    const unsigned long long other = __shfl_xor(c.key, o);
                    c.key = other < c.key ? other : c.key;
                }
            } else if (sub == 0) {
                nn_stage23(gd, u.x, u.y, u.z, c, cnt);
            }
            if (sub == 0) {
                const int o = s_slot[qi];
                if (idx_out) idx_out[o] = c.found() ? c.idx() : -1;
                d2_out[o] = c.d2();
            }
        }
    } else if (tid < total) {
        const float4 u = s_q[tid];
        NNBest c;
        c.key = s_key[tid];
        nn_stage23(gd, u.x, u.y, u.z, c, cnt);
        const int o = s_slot[tid];
        if (idx_out) idx_out[o] = c.found() ? c.idx() : -1;
        d2_out[o] = c.d2();
    }
}

__global__ void __launch_bounds__(kBlock) k_nn_points(GridDesc g, const float4* __restrict__ q, int nq,
                                                      int* __restrict__ idx, float* __restrict__ d2,
                                                      unsigned long long* __restrict__ examined) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned cnt = 0;
    const bool active = i < nq;
    const float4 v = active ? q[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    nn_block_search(g, active, v, i, idx, d2, cnt);
    add_examined(examined, cnt);
}

// few queries (centroid level): 8 lanes per query, see nn_query_group
__global__ void __launch_bounds__(kBlock) k_nn_points_group(GridDesc g, const float4* __restrict__ q, int nq,
                                                            int* __restrict__ idx, float* __restrict__ d2) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / kGroup, sub = t % kGroup;
    // a whole group is in or out of range together (kBlock is a multiple of kGroup)
    if (i >= nq) return;
    const float4 v = q[i];
    const NNBest b = nn_query_group(g, v.x, v.y, v.z, sub);
    if (sub == 0) {
        if (idx) idx[i] = b.found() ? b.idx() : -1;
        d2[i] = b.d2();
    }
}

constexpr int kXcds = 8;           // accelerator complex dies of an MI355X (one L2 each)
constexpr unsigned kSentinel = 0xffffffffu;   // d2 slot of a query that is not part of this launch

// Dense 1-NN straight from global memory (L1/L2) with the two-level search.  Stage 1 (27-cell stencil) runs for
// every query; the queries it leaves unresolved are then COMPACTED inside the block (ballot + LDS) so that the
// expensive far path runs on densely packed waves instead of a few lanes per wave (intra-wave divergence
// between resolved and unresolved lanes was ~3x the useful VALU work).  No global atomics.
__global__ void __launch_bounds__(kBlock) k_nn_dense_direct(GridDesc gd, const float4* __restrict__ pat,
                                                            const int* __restrict__ qorder,
                                                            const int* __restrict__ pt_patch,
                                                            const int* __restrict__ stable, int nq,
                                                            float* __restrict__ d2out,
                                                            unsigned long long* __restrict__ examined, int chunk) {
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Workgroup b therefore
    // takes tile (b % 8) * chunk + b / 8, so that every XCD walks ONE contiguous eighth of the Morton-ordered queries
    // and its L2 only ever holds that eighth of the target cloud (chunk = ceil(#tiles / 8); chunk = 0: identity).
    const int tile = chunk > 0 ? (int)(blockIdx.x % kXcds) * chunk + (int)(blockIdx.x / kXcds) : (int)blockIdx.x;
    const int i = tile * kBlock + threadIdx.x;
    unsigned cnt = 0;
    bool active = false;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < nq) {
        const int p = qorder ? qorder[i] : i;
        if (stable[pt_patch[p]]) { active = true; q = pat[p]; }
        else d2out[i] = __uint_as_float(kSentinel);
    }
    nn_block_search(gd, active, q, i, nullptr, d2out, cnt);
    add_examined(examined, cnt);
}

// far queries of a dense launch, handed over to k_nn_dense_far: (x, y, z, candidate d2) | slot in d2out | count | readers
struct DenseFarList {
    float4* q = nullptr;
    int* slot = nullptr;
    unsigned* count = nullptr;      // [0] entries, [1] blocks of k_nn_dense_far that have read it (the last one re-arms both)
    int edge = 1;                   // > 0: a far query whose distance is PROVED to lie above the percentile is not searched any further
                                    // (lists of at least `edge` queries; k_nn_dense_far)
};

// ---- dense 1-NN, distance only, disc-pruned (the default dense kernel) --------------------------------------------------
// One lane per query (queries in Morton order of their target cell, XCD-aware tile order as above).  `dl` is a level of
// SMALL cells (edge ~1.5 point spacings) over the target cloud: phase A takes a candidate from the query's own row
// segment (3 cells, ~7 points), phase B scans exactly the cells the candidate's ball touches (scan_disc).  A query whose
// ball is wider than kMaxRhoCells cells (far from the surface: first iterations, displaced areas) or that found no
// candidate is compacted inside the block and continues on the larger cells of `far` (fine, then coarse, then the
// general block / shell expansion).  Same float arithmetic per candidate as every other search: the result is the exact
// minimum d2 (bit-identical to k_nn_dense_direct and the oracle).
constexpr float kMaxRhoCells = 2.75f;
constexpr float kFirstBallCells = 1.1f;     // first ball of a far query without a usable candidate (k_nn_dense_far)

__device__ __forceinline__ float dense_far_path(const GridDesc& far, float4 q, float best, unsigned& cnt) {
    if (best < INFINITY) {
        const float sq = fast_sqrt_up(best);
        if (sq + 2.0f * far.fine.slack <= kMaxRhoCells * far.fine.h) {
            cnt += scan_disc_lean(far.fine, q.x, q.y, q.z, sq + 2.0f * far.fine.slack, INT_MIN, INT_MIN, 0, -1, 0, 0, best);
            return best;
        }
        if (sq + 2.0f * far.coarse.slack <= kMaxRhoCells * far.coarse.h) {
            cnt += scan_disc_lean(far.coarse, q.x, q.y, q.z, sq + 2.0f * far.coarse.slack, INT_MIN, INT_MIN, 0, -1, 0, 0, best);
            return best;
        }
    }
    // no candidate yet, or a very wide ball: the general search (stencil, ball box / block + shells on the coarse level)
    NNBest b;
    b.key = kKeyInit;
    unsigned ex = 0;
    b = nn_query(far, q.x, q.y, q.z, ex);
    cnt += ex;
    return fminf(best, b.d2());
}

// PERM = dl.perm: the level's axis roles (the query is put into the level's order for everything that concerns `dl`;
// the far path works on the levels of `far`, which are always in (x, y, z))
// FARG: the far queries of the block are searched by eight lanes each (more registers: 5 instead of 7 waves per SIMD, so only
// launches that expect many far queries use this instantiation - the first search of a pair whose probe found them)
#ifndef PW_DENSE_BLOCK
#define PW_DENSE_BLOCK 256
#endif
constexpr int kDenseBlock = PW_DENSE_BLOCK;     // threads per block of the dense search (a multiple of 64)
#ifdef PW_DENSE_BLOCKTRACE
// -DPW_DENSE_BLOCKTRACE (tools/dense_blocktrace.py): start / end (s_memrealtime, 10 ns) and XCD of every block of the last dense launch
__device__ unsigned long long pw_dense_bt[8 * 8192];       // per block: start, end, xcd, after the query, after phase A, after the ball, after the far queries, -
extern "C" __attribute__((visibility("default"))) int pwicp_debug_dense_blocktrace(unsigned long long* out, int n3) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pw_dense_bt), sizeof(unsigned long long) * (size_t)n3) == hipSuccess ? 0 : -1;   // (n3 <= 8 * 8192)
}
#endif
// The cell a query takes its FIRST candidate from (phase A: that cell and its two row neighbours): its own cell - or, for a query
// outside the level's grid (source points beyond the target's extent: the edge tiles of every pair), the grid cell nearest to it.
// Any target point is a valid first candidate - the disc scan that follows is exact around the TRUE query position - and it spares
// such a query the candidate-less general search (shell after shell on the coarse level, 15 - 30 us for the one wave that runs
// them): round 5's block trace (tools/dense_blocktrace.py) showed the launch waiting for exactly those blocks, the first and the
// last tiles of the strip order.  PW_DENSE_HOME_CLAMP=0: the query's own cell, inside the grid or not (rounds 2 - 4).
#ifndef PW_DENSE_HOME_CLAMP
#define PW_DENSE_HOME_CLAMP 1
#endif
__device__ __forceinline__ int dense_home_cell(int c, int n) {
#if PW_DENSE_HOME_CLAMP
    return min(max(c, 0), n - 1);
#else
    return c;
#endif
}

// PW_DENSE_FAST (round 6): on a level of columns the ball is ONE flat per-lane loop over a list of ranges in LDS (disc_ranges_columns /
// scan_ranges_flat, nn_device.h); 0 = every ball through scan_disc_lean's nested row loops (rounds 2 - 5)
#ifndef PW_DENSE_FAST
#define PW_DENSE_FAST 1
#endif
// PW_DENSE_A_HEAD: points of the query's own row segment requested at once before the four-per-pass loop (scan_d2_head; 0: the loop
// from the start)
#ifndef PW_DENSE_A_HEAD
#define PW_DENSE_A_HEAD 12
#endif
template <int PERM, bool FARG>
__global__ void __launch_bounds__(kDenseBlock) k_nn_dense_disc(GridLevel dl, GridDesc far, const float4* __restrict__ pat,
                                                          const int* __restrict__ qorder, const int* __restrict__ qpatch,
                                                          const int* __restrict__ stable, int nq,
                                                          float* __restrict__ d2out,
                                                          unsigned long long* __restrict__ examined, int chunk, FusedSelect fs,
                                                          DenseFarList fl, const float4* __restrict__ patq, int sub) {
    // LDS: the lanes' range lists (PW_DENSE_FAST: kDiscRangesMax x block int2, lane-major) while they search; once the whole block is
    // through (first barrier below) the same bytes hold the compacted far queries (.w carries the candidate d2 of an unresolved query).
    // The bins of the percentile selection's pass 0 (select_dev.h, fs.scratch != nullptr only) have LDS of their own: zeroed at the start
    // and filled as the lanes finish - sharing the list's bytes put a zeroing pass and a barrier into every block's epilogue (in-loop
    // launch 29.7 against 28.4 us)
    constexpr int kTailBytes = kDenseBlock * 16 + kDenseBlock * 4;
    constexpr int kListBytes = PW_DENSE_FAST ? kDiscRangesMax * kDenseBlock * 8 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kTailBytes > kListBytes ? kTailBytes : kListBytes];
    // (d2 >= +0: bins 0 .. 1023 only - 4 KB less LDS is a block more per CU, and a block per CU is worth 2 us of the launch INSIDE the
    // loop, where it finds the L2 cold (tools/inloop_dense.sh): 4 / 5 / 6 / 7 blocks per CU 35.1 / 31.3 / 29.0 / 27.2 us; an eighth -
    // the block's bookkeeping words squeezed into the bins of NaN patterns, 20 KB exactly - brings nothing more: 27.7)
    __shared__ unsigned s_hist[kFsPosBins];
    constexpr int kHistBins = kFsPosBins;
    __shared__ int s_wcnt[kDenseBlock / 64];
    float4* const s_q = (float4*)s_raw;
    int* const s_slot = (int*)(s_raw + kDenseBlock * 16);
#ifdef PW_DENSE_BLOCKTRACE
    if (threadIdx.x == 0 && blockIdx.x < 8192) { pw_dense_bt[8 * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); pw_dense_bt[8 * blockIdx.x + 2] = blockIdx.x % kXcds; }
#define PW_BT(k_) do { if (threadIdx.x == 0 && blockIdx.x < 8192) pw_dense_bt[8 * blockIdx.x + (k_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PW_BT(k_) do { } while (0)
#endif
    // block b runs on XCD b % 8: the XCDs take the ordered tiles in runs of `sub`.  One contiguous eighth of the list per XCD
    // (round 3) left the kernel waiting for the XCD whose eighth happened to be the expensive one (TA_BUSY max / mean 1.7 over
    // the CUs): 47.2 us -> 41.1 us with runs of four tiles; L2 locality does not show (runs of 1: 41.6, of 64: 44.7).
    // (Round 5, tools/dense_blocktrace.py: the launch drains for one block lifetime, ~11 us, after its last block has started.  Taking
    // the runs from both ends of the list alternately, or starting 1 / 4 ... 1 / 16 of the list before its end - so that the cloud's
    // edge tiles start first - was measured and is not in: 31.5 / 30.3 - 31.1 us against 30.3.)
    const int xr = (int)(blockIdx.x / kXcds);
    const int run = xr / sub;
    const int tile = chunk > 0 ? run * (kXcds * sub) + (int)(blockIdx.x % kXcds) * sub + xr % sub : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = tile * kDenseBlock + tid;
    unsigned cnt = 0;
    bool unresolved = false, have = false;       // have: `best` is this lane's final value
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    float best = INFINITY;
    if (fs.scratch) {
        for (int t = tid; t < kHistBins; t += kDenseBlock) s_hist[t] = 0u;
        __syncthreads();
    }
    if (i < nq) {
        // patq (the run's first search, on a source that has not moved yet): the queries lie in launch order, so the point comes
        // with the first round trip, and the stable flag of its patch shares the second one with the words of the query's own
        // row - one round trip less in a chain of about eight
        const int p = patq ? 0 : qorder[i], pa = qpatch[i];  // (p < 0: padding slot of a tile-ordered query list)
        int st = 0, loA = 0, hiA = 0, cx = 0, cy = 0, cz = 0;
        float ux = 0.f, uy = 0.f, uz = 0.f;
        if (patq) {
            q = patq[i];
            ux = PERM == 0 ? q.x : (PERM == 1 ? q.y : q.z); uy = PERM == 0 ? q.y : (PERM == 1 ? q.z : q.x); uz = PERM == 0 ? q.z : (PERM == 1 ? q.x : q.y);
            cx = dense_home_cell(cell_of(ux, dl.ox, dl.inv_h), dl.nx); cy = dense_home_cell(cell_of(uy, dl.oy, dl.inv_hy), dl.ny);
            cz = dense_home_cell(cell_of(uz, dl.oz, dl.inv_hz), dl.nz);
            row_range(dl, cy, cz, cx - 1, cx + 1, loA, hiA);
            st = stable[pa];
        } else {
            st = p >= 0 ? stable[pa] : 0;
            q = pat[max(p, 0)];                   // (both gathers in flight together)
        }
        if (st) {
            if (!patq) {
                ux = PERM == 0 ? q.x : (PERM == 1 ? q.y : q.z);       // the query in the level's axis order
                uy = PERM == 0 ? q.y : (PERM == 1 ? q.z : q.x);
                uz = PERM == 0 ? q.z : (PERM == 1 ? q.x : q.y);
                cx = dense_home_cell(cell_of(ux, dl.ox, dl.inv_h), dl.nx); cy = dense_home_cell(cell_of(uy, dl.oy, dl.inv_hy), dl.ny);
                cz = dense_home_cell(cell_of(uz, dl.oz, dl.inv_hz), dl.nz);
                row_range(dl, cy, cz, cx - 1, cx + 1, loA, hiA);
            }
            PW_BT(3);
#if PW_DENSE_A_HEAD > 0
            scan_d2_head<PERM, PW_DENSE_A_HEAD>(dl, loA, hiA, ux, uy, uz, best);
#else
            scan_d2_level<PERM>(dl, loA, hiA, ux, uy, uz, best);
#endif
            PW_BT(4);
            cnt += (unsigned)(hiA - loA);
            const float rho = fast_sqrt_up(best) + 2.0f * dl.slack;
            if (best < INFINITY && rho <= kMaxRhoCells * dl.h) {
                // (a row outside the grid / an empty clipped segment: nothing of the ball has been scanned yet)
                const bool in = cy >= 0 && cy < dl.ny && cz >= 0 && cz < dl.nz && max(cx - 1, 0) <= min(cx + 1, dl.nx - 1);
                have = true;
#if PW_DENSE_FAST
                const bool gyl = dl.inv_hy != 0.0f, gzl = dl.inv_hz != 0.0f;
                if (gyl != gzl) {                // a level of columns (wave-uniform): the flat loop
                    const int n = disc_ranges_columns(dl, ux, uy, uz, rho, in ? (gyl ? cy : cz) : INT_MIN, max(cx - 1, 0), min(cx + 1, dl.nx - 1),
                                                      loA, hiA, (int2*)s_raw, kDenseBlock, tid, cnt);
                    if (n >= 0) scan_ranges_flat<PERM>(dl, (const int2*)s_raw, kDenseBlock, tid, n, ux, uy, uz, best);
                    else { have = false; unresolved = true; }            // (cannot happen below kMaxRhoCells; a far query is always exact)
                } else
#endif
                cnt += scan_disc_lean<PERM, false, true>(dl, ux, uy, uz, rho, in ? cy : INT_MIN, in ? cz : INT_MIN, max(cx - 1, 0),
                                                         min(cx + 1, dl.nx - 1), loA, hiA, best);
                if (have) {
                    d2out[i] = best;
                    if (fs.scratch) atomicAdd(&s_hist[__float_as_uint(best) >> 21], 1u);
                }
            } else {
                unresolved = true;
            }
        } else {
            d2out[i] = __uint_as_float(kSentinel);
        }
    }
    PW_BT(5);
    // far queries: compacted inside the block so that their longer scans run on packed waves
    const unsigned long long mask = __ballot(unresolved);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_wcnt[wave] = __popcll(mask);
    __syncthreads();                                 // (every lane of the block is through with its range list: the LDS changes hands)
    PW_BT(6);
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kDenseBlock / 64; ++w) {
        if (w < wave) base += s_wcnt[w];
        total += s_wcnt[w];
    }
    if (unresolved) {
        s_q[base + before] = make_float4(q.x, q.y, q.z, best);
        s_slot[base + before] = i;
    }
    __syncthreads();
    if (FARG) {
        // ... and handed to the launch behind this one (k_nn_dense_far), which puts EIGHT lanes on each: a far query scans ~40
        // rows and ~200 candidates (the first iteration of a real pair: half the queries), and a launch of 10^5 queries is a
        // wave or two per SIMD - one lane per far query makes this launch as long as its slowest lane's chain of round trips
        // (217 us on the reference's Epoch_002 -> Epoch_001, 140 k points), eight lanes on 32 queries of the block at a time
        // cost five such chains one after the other.
        __shared__ unsigned s_base;
        if (tid == 0 && total > 0) s_base = atomicAdd(fl.count, (unsigned)total);
        __syncthreads();
        if (tid < total) {
            fl.q[s_base + tid] = s_q[tid];
            fl.slot[s_base + tid] = s_slot[tid];
        }
    } else if (tid < total) {
        const float4 u = s_q[tid];
        const float d = dense_far_path(far, u, u.w, cnt);
        d2out[s_slot[tid]] = d;
        if (fs.scratch) atomicAdd(&s_hist[__float_as_uint(d) >> 21], 1u);
    }
    PW_BT(7);
    add_examined(examined, cnt);
    if (fs.scratch) fs_pass0_epilogue(s_hist, fs, kHistBins);
#ifdef PW_DENSE_BLOCKTRACE
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 8192) pw_dense_bt[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

#ifdef PW_DENSE_SLOTS
// -DPW_DENSE_SLOTS (tools/dense_slots.py, round 6): what the wave of k_nn_dense_disc EXECUTES against what its lanes NEED.  A lane
// walks its ball z-slab by z-slab, four rows at a time, row by row in passes of four candidates; the wave runs every one of those
// loops to its slowest lane's count, i.e. sum over (slab, row slot, piece) of max over lanes of the passes - against the max over
// lanes of a lane's OWN sum of passes, which is what one flat per-lane loop over all its ranges would execute.  This kernel repeats
// the search's decisions (same phase A, same ball, same ranges) without scanning the ball and counts both, launch by launch.
__device__ unsigned long long pw_dense_slots[128];
extern "C" __attribute__((visibility("default"))) int pwicp_debug_dense_slots(unsigned long long* out, int n, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pw_dense_slots), sizeof(unsigned long long) * (size_t)n) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[128]; if (hipMemcpyToSymbol(HIP_SYMBOL(pw_dense_slots), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
__device__ __forceinline__ int slots_wmax(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ unsigned long long slots_wsum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int PERM>
__global__ void __launch_bounds__(kDenseBlock) k_dense_slots(GridLevel dl, const float4* __restrict__ pat, const int* __restrict__ qorder,
                                                             const int* __restrict__ qpatch, const int* __restrict__ stable, int nq, int chunk,
                                                             const float4* __restrict__ patq, int sub) {
    const int xr = (int)(blockIdx.x / kXcds);
    const int run = xr / sub;
    const int tile = chunk > 0 ? run * (kXcds * sub) + (int)(blockIdx.x % kXcds) * sub + xr % sub : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = tile * kDenseBlock + tid;
    int st = 0, loA = 0, hiA = 0, cx = 0, cy = 0, cz = 0;
    float ux = 0.f, uy = 0.f, uz = 0.f, best = INFINITY;
    if (i < nq) {
        const int p = patq ? 0 : qorder[i], pa = qpatch[i];
        st = (patq || p >= 0) ? stable[pa] : 0;
        const float4 q = patq ? patq[i] : pat[max(p, 0)];
        if (st) {
            ux = PERM == 0 ? q.x : (PERM == 1 ? q.y : q.z); uy = PERM == 0 ? q.y : (PERM == 1 ? q.z : q.x); uz = PERM == 0 ? q.z : (PERM == 1 ? q.x : q.y);
            cx = dense_home_cell(cell_of(ux, dl.ox, dl.inv_h), dl.nx); cy = dense_home_cell(cell_of(uy, dl.oy, dl.inv_hy), dl.ny);
            cz = dense_home_cell(cell_of(uz, dl.oz, dl.inv_hz), dl.nz);
            row_range(dl, cy, cz, cx - 1, cx + 1, loA, hiA);
            scan_d2_level<PERM>(dl, loA, hiA, ux, uy, uz, best);
        }
    }
    const int pA = st ? (hiA - loA + 3) >> 2 : 0;
    const float rho = fast_sqrt_up(best) + 2.0f * dl.slack;
    const bool res = st && best < INFINITY && rho <= kMaxRhoCells * dl.h;
    // the ball's rows exactly as scan_disc_lean enumerates them, wave-uniformly
    const GridLevel& g = dl;
    const bool gy = g.inv_hy != 0.0f, gz = g.inv_hz != 0.0f;
    const bool in = cy >= 0 && cy < dl.ny && cz >= 0 && cz < dl.nz && max(cx - 1, 0) <= min(cx + 1, dl.nx - 1);
    const int sy = in ? cy : INT_MIN, sz = in ? cz : INT_MIN, sx0 = max(cx - 1, 0), sx1 = min(cx + 1, dl.nx - 1);
    const int y0 = gy ? max(icell(uy - rho, g.oy, g.inv_hy), 0) : 0, y1 = gy ? min(icell(uy + rho, g.oy, g.inv_hy), g.ny - 1) : 0;
    const int z0 = gz ? max(icell(uz - rho, g.oz, g.inv_hz), 0) : 0, z1 = gz ? min(icell(uz + rho, g.oz, g.inv_hz), g.nz - 1) : 0;
    const float rho2 = rho * rho, slack2 = 2.0f * g.slack;
    const int nzl = res ? max(z1 - z0 + 1, 0) : 0, nyl = res ? max(y1 - y0 + 1, 0) : 0;
    const int mz = slots_wmax(nzl), myb = slots_wmax((nyl + 3) >> 2);
    int eB = 0, lB = 0, cB = 0, nranges = 0, setups = 0;
    bool fast = res && nzl <= 1 && y0 >= cy - 1 && y1 <= cy + 1 && in;
    for (int iz = 0; iz < mz; ++iz) {
        const int z = z0 + iz;
        bool zin = iz < nzl;
        float remz = rho2;
        if (zin && gz) {
            const float lo = g.oz + (float)z * g.h;
            const float ez = fmaxf(fmaxf(lo - uz, uz - (lo + g.h)) - slack2, 0.0f);
            remz = rho2 - ez * ez;
            if (!(remz > 0.0f)) zin = false;
        }
        for (int ib = 0; ib < myb; ++ib) {
            const bool bin = zin && y0 + 4 * ib <= y1;
            if (__ballot(bin)) ++setups;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int lo = 0, hi = 0, lo2 = 0, hi2 = 0;
                const int y = y0 + 4 * ib + k;
                if (bin && y <= y1) {
                    float rem = remz;
                    bool ok = true;
                    if (gy) {
                        const float l = g.oy + (float)y * g.h;
                        const float ey = fmaxf(fmaxf(l - uy, uy - (l + g.h)) - slack2, 0.0f);
                        rem = remz - ey * ey;
                        ok = rem > 0.0f;
                    }
                    if (ok) {
                        const float rx = fast_sqrt_up(rem) + slack2;
                        const int x0 = max(icell(ux - rx, g.ox, g.inv_h), 0), x1 = min(icell(ux + rx, g.ox, g.inv_h), g.nx - 1);
                        if (x0 <= x1) {
                            if (x0 < cx - 1 || x1 > cx + 1) fast = false;
                            const int row = (z * g.ny + y) * g.nx;
                            if (y == sy && z == sz) {
                                if (x0 < sx0) { lo = g.cell_start[row + x0]; hi = loA; }
                                if (x1 > sx1) { lo2 = hiA; hi2 = g.cell_start[row + x1 + 1]; }
                            } else {
                                lo = g.cell_start[row + x0];
                                hi = g.cell_start[row + x1 + 1];
                            }
                        }
                    }
                }
                const int p1 = (hi - lo + 3) >> 2, p2 = (hi2 - lo2 + 3) >> 2;
                eB += slots_wmax(p1) + slots_wmax(p2);
                lB += p1 + p2;
                cB += (hi - lo) + (hi2 - lo2);
                nranges += (p1 > 0) + (p2 > 0);
            }
        }
    }
    const int eA = slots_wmax(pA), mB = slots_wmax(lB), mAB = slots_wmax(pA + lB);
    const unsigned long long anyres = __ballot(res);
    const unsigned long long s_res = slots_wsum(res ? 1 : 0), s_pA = slots_wsum((unsigned long long)pA), s_lB = slots_wsum((unsigned long long)lB),
                             s_cA = slots_wsum((unsigned long long)(st ? hiA - loA : 0)), s_cB = slots_wsum((unsigned long long)cB),
                             s_unres = slots_wsum(st && !res ? 1 : 0), s_st = slots_wsum(st ? 1 : 0), s_fast = slots_wsum(fast ? 1 : 0);
    const bool allfast = __ballot(res && !fast) == 0ull;
    if (lane == 0 && __ballot(st != 0)) {
        unsigned long long* S = pw_dense_slots;
        atomicAdd(&S[0], anyres ? 1ull : 0ull); atomicAdd(&S[1], s_res); atomicAdd(&S[2], (unsigned long long)eA); atomicAdd(&S[3], s_pA);
        atomicAdd(&S[4], (unsigned long long)eB); atomicAdd(&S[5], s_lB); atomicAdd(&S[6], (unsigned long long)mB);
        atomicAdd(&S[7], (unsigned long long)mAB); atomicAdd(&S[8], (unsigned long long)setups); atomicAdd(&S[9], s_cA);
        atomicAdd(&S[10], s_cB); atomicAdd(&S[11], s_unres); atomicAdd(&S[12], s_st); atomicAdd(&S[13], s_fast);
        atomicAdd(&S[14], anyres && allfast ? 1ull : 0ull); atomicAdd(&S[15], 1ull);
        atomicAdd(&S[80 + min(eB >> 1, 15)], 1ull);
        atomicAdd(&S[96 + min(mB, 15)], 1ull);
    }
    if (res) {
        atomicAdd(&pw_dense_slots[16 + min((int)(rho * dl.inv_h * 4.0f), 15)], 1ull);
        atomicAdd(&pw_dense_slots[32 + min(nranges, 15)], 1ull);
        atomicAdd(&pw_dense_slots[48 + min(nzl * nyl, 15)], 1ull);
        atomicAdd(&pw_dense_slots[64 + min(lB, 15)], 1ull);
    }
}
#endif

// (Round 5's search on an LDS copy of the block's candidate window - k_nn_dense_win, exact, 37.5 - 51 us against 30 - was removed
// in round 6; its design and counters are profiles/r05_dense_variants.txt (1), the code is in the history at 195adaa.)

#ifndef PW_FAR_RUN
#define PW_FAR_RUN 4
#endif
// The far queries of the dense launch before it, eight lanes each (group-cooperative forms of dense_far_path's steps: the
// ball of the candidate on the larger cells of `far`, else the general search); their share of the selection's pass 0.
#ifdef PW_FAR_STATS
// -DPW_FAR_STATS (tools/far_stats.py): where the far queries end - [0] queries, [1] candidate's ball on the fine level, [2] small ball
// fine, [3] wide ball fine, [4] candidate's ball coarse, [5] small ball coarse, [6] wide ball coarse, [7] general search, [8+l] launches
__device__ unsigned long long pw_far_stats[16];
extern "C" __attribute__((visibility("default"))) int pwicp_debug_far_stats(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(pw_far_stats), sizeof(pw_far_stats)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(pw_far_stats), z, sizeof(z)); }
    return 0;
}
#define FAR_STAT(k) do { if (sub == 0) atomicAdd(&pw_far_stats[k], 1ull); } while (0)
#else
#define FAR_STAT(k) do {} while (0)
#endif
// The general search of ONE query by a whole block (distance only, level c in (x, y, z) order): the Chebyshev block of radius r
// around the query's cell, then shell after shell, until every point outside the scanned block is provably farther than the best
// one (the bound of nn_resolved) or the grid is exhausted - the stage 3 of nn_query_group, but with the row segments of a block /
// shell dealt to the threads, a prefix over their lengths in LDS, and the candidates taken thread by thread from that line.
// A query with nothing within 2.75 coarse cells (source points outside the target's coverage: a handful per launch on the
// reference's Epoch_012) walks ~10 shells; on eight lanes that is ~50 us and the launch waits for it - 107 -> 63 us, 55 -> 36 us
// for the two far launches of that pair.  Same float expression per candidate, same stopping rule: the same exact minimum.
// Called by ALL threads of the block with block-uniform arguments; s_lo / s_pre: kBlock (+ 1) ints, s_red: kBlock / 64 floats.
__device__ float nn_block_shells_d2(const GridLevel& c, float qx, float qy, float qz, float best_in, int* s_lo, int* s_pre, float* s_red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cx = cell_of(qx, c.ox, c.inv_h), cy = cell_of(qy, c.oy, c.inv_hy), cz = cell_of(qz, c.oz, c.inv_hz);
    const int ex = max(0, max(-cx, cx - (c.nx - 1))), ey = max(0, max(-cy, cy - (c.ny - 1))), ez = max(0, max(-cz, cz - (c.nz - 1)));
    int r = max(max(ex, ey), max(ez, 1));
    const int rcover = max(max(max(cx, c.nx - 1 - cx), max(cy, c.ny - 1 - cy)), max(cz, c.nz - 1 - cz));
    float best = best_in;
    for (bool first = true;; first = false) {
        // (only the rows that exist: a level of columns has ONE layer of them, whatever r)
        const int dy0 = max(-r, -cy), dy1 = min(r, c.ny - 1 - cy), dz0 = max(-r, -cz), dz1 = min(r, c.nz - 1 - cz);
        const int wy = max(dy1 - dy0 + 1, 0), wz = max(dz1 - dz0 + 1, 0);
        const long long nseg = (first ? 1ll : 2ll) * wy * wz;
        float local = INFINITY;
        for (long long sb = 0; sb < nseg; sb += kBlock) {
            const long long sg = sb + tid;
            int lo = 0, hi = 0;
            if (sg < nseg) {
                const int t = (int)(first ? sg : sg >> 1), part = first ? 0 : (int)(sg & 1);
                const int dz = dz0 + t / wy, dy = dy0 + t % wy;
                if (first || dz == -r || dz == r || dy == -r || dy == r) {
                    if (part == 0) row_range(c, cy + dy, cz + dz, cx - r, cx + r, lo, hi);
                } else {
                    row_range(c, cy + dy, cz + dz, part ? cx + r : cx - r, part ? cx + r : cx - r, lo, hi);
                }
            }
            const int len = hi - lo;
            int incl = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(incl, o);
                if (lane >= o) incl += u;
            }
            if (lane == 63) s_pre[kBlock - 1 - wave] = incl;         // (parked at the far end until the sums are known)
            __syncthreads();
            int basew = 0, T = 0;
#pragma unroll
            for (int k = 0; k < kBlock / 64; ++k) {
                const int v = s_pre[kBlock - 1 - k];
                if (k < wave) basew += v;
                T += v;
            }
            __syncthreads();
            s_lo[tid] = lo;
            s_pre[tid] = basew + incl - len;
            __syncthreads();
            for (int cnd = tid; cnd < T; cnd += kBlock) {
                int a = 0, b = kBlock - 1;                           // last j with s_pre[j] <= cnd
                while (a < b) {
                    const int m = (a + b + 1) >> 1;
                    if (s_pre[m] <= cnd) a = m; else b = m - 1;
                }
                nn_consider_d2<0>(c.pts[s_lo[a] + (cnd - s_pre[a])], qx, qy, qz, local);
            }
            __syncthreads();
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) local = fminf(local, __shfl_xor(local, o));
        if (lane == 0) s_red[wave] = local;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kBlock / 64; ++k) best = fminf(best, s_red[k]);
        __syncthreads();
#ifdef PW_FAR_STATS
        if (tid == 0) { atomicAdd(&pw_far_stats[10], 1ull); atomicMax(&pw_far_stats[11], (unsigned long long)r); }
#endif
        const float bound = (float)r * c.h - 2.0f * c.slack;
        if (best < INFINITY && bound > 0.0f && best < bound * bound * 0.99999f) break;
        if (r >= rcover) break;
        ++r;
    }
    return best;
}

__global__ void __launch_bounds__(kBlock) k_nn_dense_far(GridDesc far, DenseFarList fl, float* __restrict__ d2out, FusedSelect fs,
                                                         unsigned long long* __restrict__ examined) {
    __shared__ unsigned s_hist[kFsBins];
    __shared__ unsigned s_n, s_last;
    constexpr int kSlowCap = 32;            // queries of this block that need the general search: the whole block takes them at the end
    __shared__ int s_slow[kSlowCap];
    __shared__ float s_slow_d[kSlowCap];
    __shared__ unsigned s_nslow;
    __shared__ int s_blo[kBlock], s_bpre[kBlock];
    __shared__ float s_bred[kBlock / 64];
    const int tid = threadIdx.x;
    // The search only feeds the percentile of the distances (C.cpp:177), and the launch starts with the dense kernel's own values in
    // the selection's bins: when k + 1 of THOSE already lie below an edge, the percentile does - whatever the far queries add - and a
    // far query is finished as soon as everything within some radius g of it has been examined without a result below the edge
    // (min(d, g^2) >= edge: its true distance lies above the percentile, and so does the candidate d - or infinity - written for it:
    // an upper bound of the true value, so that a block that reads the bins later still counts true values only below ITS edge).
    // The rank statistics below the edge are untouched: the selected value is bit for bit the one of the exact search.
    // Source points outside the overlap of a pair are what this is for: the widest balls and the general search.  fl.edge == 0: off.
    if (tid == 0) { s_n = __hip_atomic_load(&fl.count[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_nslow = 0u; }
    __syncthreads();
    const int n = (int)s_n;
    float edge = INFINITY;
    if (fs.scratch && fl.edge > 0 && n >= fl.edge) {
        const unsigned eb = fs_partial_edge(s_hist, fs);
        if (eb != ~0u) edge = __uint_as_float(eb);
    } else if (fs.scratch) {
        for (int t = tid; t < kFsBins; t += kBlock) s_hist[t] = 0u;
        __syncthreads();
    }
    const int sub = tid % kGroup;
    unsigned n_bounded = 0;         // queries of this group that were NOT searched to the end (proved above the percentile's edge)
    // everything within g of the query examined, best result d: is the query's distance known to lie above the percentile?
    auto above_edge = [&](float d, float g) { return g > 0.0f && fminf(d, g * g * 0.99999f) >= edge; };
    // (entry qi to block qi % #blocks: neighbours on the list are neighbours in space, and the few queries that need the general
    // search come in clusters - dealt out like this they end up on different blocks instead of queueing on one)
    // PW_FAR_RUN consecutive entries stay together (their balls share cache lines), the runs of a block are #blocks runs apart
    constexpr int kRun = PW_FAR_RUN;
    for (int qi = (((tid / kGroup) / kRun) * (int)gridDim.x + (int)blockIdx.x) * kRun + (tid / kGroup) % kRun; qi < n;
         qi += (int)gridDim.x * (kBlock / kGroup)) {
        const float4 u = fl.q[qi];
        float d = u.w;                              // the candidate of the query's own row segment, if any
        // Level by level (the larger cells of `far`: fine, then coarse): with a candidate whose ball fits kMaxRhoCells cells,
        // that ball; otherwise the widest ball that fits - whatever it finds within its guaranteed radius (the ball minus the
        // rounding slack of the cell boundaries) is the exact minimum, and most queries of a misaligned first iteration have
        // no candidate at all in their own three cells.  Only a query with nothing within 2.75 coarse cells takes the general
        // search.
        bool done = false;
#pragma unroll
        for (int lv = 0; lv < 2 && !done; ++lv) {
            const GridLevel& L = lv == 0 ? far.fine : far.coarse;
            const float lim = kMaxRhoCells * L.h, sl2 = 2.0f * L.slack;
            const float sq = d < INFINITY ? fast_sqrt_up(d) : INFINITY;
            if (sq + sl2 <= lim) {
                scan_disc_group<kGroup>(L, u.x, u.y, u.z, sq + sl2, sub, d);
                done = true;
                FAR_STAT(1 + 3 * lv);
            } else {
                // (a small ball first: three quarters of a real pair's far queries are within a cell of the surface)
                const float r1 = kFirstBallCells * L.h;
                scan_disc_group<kGroup>(L, u.x, u.y, u.z, r1, sub, d);
                done = d < INFINITY && fast_sqrt_up(d) + sl2 <= r1;
                if (!done && above_edge(d, r1 - sl2)) { done = true; ++n_bounded; }
                if (done) FAR_STAT(2 + 3 * lv);
                if (!done) {
                    const float sq2 = d < INFINITY ? fast_sqrt_up(d) + sl2 : INFINITY;
                    const float r2 = fminf(sq2, lim);                  // the candidate's ball if it fits, else the widest
                    scan_disc_group<kGroup>(L, u.x, u.y, u.z, r2, sub, d);
                    done = sq2 <= lim || (d < INFINITY && fast_sqrt_up(d) + sl2 <= lim);
                    if (!done && above_edge(d, r2 - sl2)) { done = true; ++n_bounded; }
                    if (done) FAR_STAT(3 + 3 * lv);
                }
            }
        }
        FAR_STAT(0);
        if (!done) {                                // nothing within 2.75 coarse cells: the general search, by the whole block below
            FAR_STAT(7);
            unsigned at = 0u;
            if (sub == 0) at = atomicAdd(&s_nslow, 1u);
            at = (unsigned)__shfl((int)at, 0, kGroup);
            if (at < (unsigned)kSlowCap) {
                if (sub == 0) { s_slow[at] = qi; s_slow_d[at] = d; }
                continue;
            }
            const NNBest b = nn_query_group<kGroup>(far, u.x, u.y, u.z, sub);      // (more than the block keeps: on the group's lanes)
            d = fminf(d, b.d2());
        }
        if (sub == 0) {
            d2out[fl.slot[qi]] = d;
            if (fs.scratch) atomicAdd(&s_hist[__float_as_uint(d) >> 21], 1u);
        }
    }
    __syncthreads();
    for (int k = 0, nk = (int)min(s_nslow, (unsigned)kSlowCap); k < nk; ++k) {
        const int qi = s_slow[k];
        const float4 u = fl.q[qi];
        const float d = nn_block_shells_d2(far.coarse, u.x, u.y, u.z, s_slow_d[k], s_blo, s_bpre, s_bred);
        if (tid == 0) {
            d2out[fl.slot[qi]] = d;
            if (fs.scratch) atomicAdd(&s_hist[__float_as_uint(d) >> 21], 1u);
        }
    }
    if (fs.scratch) fs_pass0_epilogue(s_hist, fs);
    // honest accounting: a query that was cut short is NOT a completed 1-NN query - counted in bits 40.. of the block's
    // diagnostic counter (bits 0..39: candidates examined), pwicp_result.n_dense_bounded
    if (examined) {
        unsigned long long nb = sub == 0 ? (unsigned long long)n_bounded : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nb += __shfl_xor(nb, o);
        if ((tid & 63) == 0 && nb) atomicAdd(&examined[(blockIdx.x & 255) * 16], nb << 40);
    }
    // the block that reads the count last re-arms the list for the next launch
    __syncthreads();
    if (tid == 0) {
        const unsigned prev = atomicAdd(&fl.count[1], 1u);
        s_last = (prev == gridDim.x - 1u) ? 1u : 0u;
        if (s_last) { fl.count[0] = 0u; fl.count[1] = 0u; }
    }
}

// passes 1 / 2 of the fused selection as launches of their own (only when no transform / front launch follows the dense
// search: the iteration that reaches Stage 3 while still in Stage 1)
template <int PASS>
__global__ void __launch_bounds__(kBlock) k_fs_pass(FusedSelect fs) {
    __shared__ unsigned h[kFsBins];
    fs_pass_embedded<PASS>(h, fs, (int)blockIdx.x);
}

// patch id of the i-th query of the dense search (static: one coalesced load instead of a dependent gather per launch)
__global__ void k_gather_int(const int* __restrict__ src, const int* __restrict__ order, int n, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[order[i]];
}

// the queries of the dense search themselves in launch order (static while the source has not moved)
__global__ void k_gather_f4(const float4* __restrict__ src, const int* __restrict__ order, int n, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[max(order[i], 0)];
}

// Morton code (10 bits per axis) of the fine cell of each point, for the one-off query ordering
__device__ __forceinline__ unsigned part1by2(unsigned x) {
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
__global__ void k_morton_keys(GridLevel g, const float4* __restrict__ p, int n, int shift, unsigned* __restrict__ keys,
                              int* __restrict__ vals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = p[i];
    const unsigned cx = (unsigned)min(max(cell_of(v.x, g.ox, g.inv_h), 0), g.nx - 1) >> shift;
    const unsigned cy = (unsigned)min(max(cell_of(v.y, g.oy, g.inv_hy), 0), g.ny - 1) >> shift;
    const unsigned cz = (unsigned)min(max(cell_of(v.z, g.oz, g.inv_hz), 0), g.nz - 1) >> shift;
    keys[i] = part1by2(cx) | (part1by2(cy) << 1) | (part1by2(cz) << 2);
    vals[i] = i;
}

// keys of the STRIP order: the cells of level g (axis roles g.perm) in blocks of xb cells along the row direction by rb rows;
// blocks row-major, rows inside a block one after the other, cells of a row segment left to right.  64 consecutive queries
// then lie along one row: their candidate ranges tile one contiguous span of the row's points.
__global__ void k_strip_keys(GridLevel g, const float4* __restrict__ p, int n, int xb, int rb, int serp, unsigned* __restrict__ keys,
                             int* __restrict__ vals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = p[i];
    const float ux = g.perm == 0 ? v.x : (g.perm == 1 ? v.y : v.z), uy = g.perm == 0 ? v.y : (g.perm == 1 ? v.z : v.x),
                uz = g.perm == 0 ? v.z : (g.perm == 1 ? v.x : v.y);
    const unsigned cx = (unsigned)min(max(cell_of(ux, g.ox, g.inv_h), 0), g.nx - 1);
    const unsigned cy = (unsigned)min(max(cell_of(uy, g.oy, g.inv_hy), 0), g.ny - 1);
    const unsigned cz = (unsigned)min(max(cell_of(uz, g.oz, g.inv_hz), 0), g.nz - 1);
    const unsigned row = cz * (unsigned)g.ny + cy, nxb = ((unsigned)g.nx + xb - 1) / xb;
    // (serp: the rows of a block boustrophedon - a wave that crosses a row end stays in one place, k_nn_dense_win)
    const unsigned xin = cx % xb, rin = row % rb;
    keys[i] = ((row / rb) * nxb + cx / xb) * (unsigned)(rb * xb) + rin * xb + ((serp && (rin & 1u)) ? (unsigned)xb - 1u - xin : xin);
    vals[i] = i;
}

// ---- calibration of the PMC byte counters for THIS search's access pattern (tools/gather_calibration.py) ----------------------
// FETCH_SIZE is calibrated for wide coalesced streams (MI355X_MICROARCH.md: x2 for 16 B / lane); the dense search gathers 12-byte
// candidates at per-lane addresses.  This kernel does exactly that with a byte count known by construction: lane t reads the 12
// bytes at offset t * stride of a buffer far larger than the Infinity Cache (so every line comes from HBM, once).
__global__ void __launch_bounds__(256) k_gather_calibration(const float* __restrict__ buf, long long n, int stride_floats, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const PwXyz3 p = *(const PwXyz3*)(buf + t * (long long)stride_floats);
    const float s = p.x + p.y + p.z;
    if (s == 12345.678f) out[0] = s;               // (never: keeps the load)
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int pwicp_debug_gather_calibration(pwicp_context* ctx, long long n_gathers, int stride_bytes, int launches) {
    if (!ctx || n_gathers <= 0 || stride_bytes < 12 || stride_bytes % 4) return PWICP_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf<float> buf, out;
    HIPCHK(ctx, buf.reserve((size_t)n_gathers * (size_t)(stride_bytes / 4) + 4));
    HIPCHK(ctx, out.reserve(4));
    HIPCHK(ctx, hipMemsetAsync(buf.p, 0, ((size_t)n_gathers * (size_t)(stride_bytes / 4) + 4) * sizeof(float), ctx->stream));
    for (int l = 0; l < launches; ++l)
        hipLaunchKernelGGL(k_gather_calibration, dim3((unsigned)((n_gathers + 255) / 256)), dim3(256), 0, ctx->stream, buf.p, n_gathers,
                           stride_bytes / 4, out.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PWICP_OK;
}
namespace {

// ---- exact k-NN of every point of a cloud within the cloud itself (segmentation front end) ------------------------
// Replaces the per-point cl::KDTree::FindKNearestNeighbors loop of PatchGenerationAndRefinement
// (src/Segmentation.cpp:30-41; codelibrary/util/tree/kd_tree.h:266-280): the k nearest points (the query itself
// included, distance 0) in ascending order of the DOUBLE squared distance ((dx*dx)+dy*dy)+dz*dz of the
// float->double converted coordinates (util/metric/squared_euclidean.h), ties by index.
// One lane per point, in cell order (neighbouring lanes scan the same cells).  The lane's sorted candidate list
// lives in global memory in column layout (entry e of lane t at [e*n + t]: coalesced across the wave).
// The (2r+1)^3 block is grown until the k-th distance is provably smaller than anything outside the block.
// Real = double: the front end's metric (codelibrary, squared distance in double).  Real = float: PCL's searches
// (flann::L2_Simple<float>).  out_nb (optional): row `self` = the k neighbour indices.  out_mean (optional): the mean
// of sqrt(d2) over neighbours 1..k-1 (the query itself is neighbour 0), accumulated in double and rounded to float —
// the per-point statistic of pcl::StatisticalOutlierRemoval (filters/impl/statistical_outlier_removal.hpp).
template <typename Real>
__global__ void __launch_bounds__(kBlock) k_knn(GridLevel g, int k, Real* __restrict__ nd, int* __restrict__ ni,
                                                int* __restrict__ out_nb, float* __restrict__ out_mean) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = g.n;
    if (t >= n) return;
    const float4 q = g.pts[t];
    const int self = __float_as_int(q.w);
    const Real qx = (Real)q.x, qy = (Real)q.y, qz = (Real)q.z;
    const int cx = cell_of(q.x, g.ox, g.inv_h), cy = cell_of(q.y, g.oy, g.inv_hy), cz = cell_of(q.z, g.oz, g.inv_hz);
    const int rcover = max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz));
    int cnt = 0;
    for (int r = 2;; ++r) {
        cnt = 0;
        for (int dz = -r; dz <= r; ++dz)
            for (int dy = -r; dy <= r; ++dy) {
                int lo, hi;
                row_range(g, cy + dy, cz + dz, cx - r, cx + r, lo, hi);
                for (int j = lo; j < hi; ++j) {
                    const float4 p = g.pts[j];
                    const Real dx = qx - (Real)p.x, dy2 = qy - (Real)p.y, dz2 = qz - (Real)p.z;
                    Real d2 = dx * dx;
                    d2 = d2 + dy2 * dy2;
                    d2 = d2 + dz2 * dz2;
                    const int id = __float_as_int(p.w);
                    // sorted insertion by (d2, id)
                    if (cnt == k) {
                        const Real wd = nd[(size_t)(k - 1) * n + t];
                        const int wi = ni[(size_t)(k - 1) * n + t];
                        if (!(d2 < wd || (d2 == wd && id < wi))) continue;
                    }
                    int pos = cnt < k ? cnt : k - 1;
                    while (pos > 0) {
                        const Real pd = nd[(size_t)(pos - 1) * n + t];
                        const int pi = ni[(size_t)(pos - 1) * n + t];
                        if (!(d2 < pd || (d2 == pd && id < pi))) break;
                        nd[(size_t)pos * n + t] = pd;
                        ni[(size_t)pos * n + t] = pi;
                        --pos;
                    }
                    nd[(size_t)pos * n + t] = d2;
                    ni[(size_t)pos * n + t] = id;
                    if (cnt < k) ++cnt;
                }
            }
        if (r >= rcover) break;
        if (cnt == k) {
            const double bound = (double)r * (double)g.h - 2.0 * (double)g.slack;
            if (bound > 0.0 && (double)nd[(size_t)(k - 1) * n + t] < bound * bound * 0.99999) break;
        }
    }
    if (out_nb)
        for (int e = 0; e < k; ++e) out_nb[(size_t)self * k + e] = (e < cnt) ? ni[(size_t)e * n + t] : -1;
    if (out_mean) {
        double s = 0.0;
        for (int e = 1; e < cnt; ++e) s += (double)sqrtf((float)nd[(size_t)e * n + t]);
        out_mean[self] = (float)(s / (double)(k - 1));
    }
}

// ---- k-th smallest of non-negative floats: 3-pass radix select on the bit pattern ---------------------
// scratch layout: [0] prefix, [1] k remaining, [8 + pass*2048 ...] histograms
constexpr int kSelBins = 2048;

__global__ void k_select_init(unsigned* scratch, int k) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 8 + 3 * kSelBins) scratch[t] = (t == 1) ? (unsigned)k : 0u;
}

// Histogram pass + pick in ONE launch: the block that flushes its bins last (device counter in scratch[2]) scans the
// 2048 bins on its first wave, narrows the prefix / rank for the next pass and re-arms the bins and the counter, so a
// selection is three launches and the scratch buffer is ready for the next selection when the last one ends.
// The final pass can publish the selected value to a host mailbox (see k_mail in loop.hip) from the same launch.
template <int PASS>
__global__ void k_select_pass(const float* __restrict__ v, int n, int k0, unsigned* __restrict__ scratch,
                              float* __restrict__ out, SelectMail mail) {
    // 8 replicas of the histogram, chosen by lane: squared distances cluster in a few exponent bins, and same-bin
    // LDS atomics from the 64 lanes of a wave would serialise
    __shared__ unsigned h[8 * kSelBins];
    for (int t = threadIdx.x; t < 8 * kSelBins; t += blockDim.x) h[t] = 0;
    __syncthreads();
    unsigned* hr = h + (threadIdx.x & 7) * kSelBins;
    const unsigned prefix = scratch[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned u = __float_as_uint(v[i]);
        if (u == 0xffffffffu) continue;          // slot of a query that was not part of the launch
        if (PASS == 0) {
            atomicAdd(&hr[u >> 21], 1u);
        } else if (PASS == 1) {
            if ((u >> 21) == (prefix >> 21)) atomicAdd(&hr[(u >> 10) & 2047u], 1u);
        } else {
            if ((u >> 10) == (prefix >> 10)) atomicAdd(&hr[u & 1023u], 1u);
        }
    }
    __syncthreads();
    unsigned* gh = scratch + 8 + PASS * kSelBins;
    unsigned seen = 0;
    for (int t = threadIdx.x; t < kSelBins; t += blockDim.x) {
        unsigned s = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += h[r * kSelBins + t];
        if (s) seen |= atomicAdd(&gh[t], s);
    }
    // The bin updates must have been PERFORMED before this block is counted.  They are device-coherent read-modify-
    // writes, so waiting for their return values is enough (an acknowledged store is not: it may still be on its way
    // to the coherence point — observed as a wrong percentile once in ~100 runs); no cache write-back is needed.
    asm volatile("" ::"v"(seen));
    __syncthreads();
    if (threadIdx.x >= 64) return;
    unsigned last = 0;
    if (threadIdx.x == 0) last = (atomicAdd(&scratch[2], 1u) == gridDim.x - 1u) ? 1u : 0u;
    last = (unsigned)__shfl((int)last, 0);
    if (!last) return;
    // ---- pick (one wave: each lane owns 32 consecutive bins) ----
    const int lane = threadIdx.x;
    unsigned cnt[32];
    unsigned local = 0;
#pragma unroll
    for (int b = 0; b < 32; ++b) {
        cnt[b] = __hip_atomic_load(&gh[lane * 32 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        local += cnt[b];
    }
    unsigned incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const unsigned excl = incl - local;
    const unsigned k = (PASS == 0) ? (unsigned)k0 : scratch[1];
    const bool mine = (k >= excl) && (k < incl);
    unsigned value = 0;
    if (mine) {
        unsigned run = excl;
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const unsigned c = cnt[b];
            if (k >= run && k < run + c) {
                const unsigned bin = (unsigned)(lane * 32 + b);
                unsigned pf = prefix;
                if (PASS == 0) pf = bin << 21;
                else if (PASS == 1) pf |= bin << 10;
                else pf |= bin;
                value = pf;
                scratch[0] = (PASS == 2) ? 0u : pf;          // prefix for the next pass; cleared after the last one
                scratch[1] = k - run;
                if (PASS == 2) out[0] = __uint_as_float(pf);
            }
            run += c;
        }
    }
    // re-arm this pass's bins and the counter
#pragma unroll
    for (int b = 0; b < 32; ++b) gh[lane * 32 + b] = 0u;
    if (lane == 0) scratch[2] = 0u;
    if (PASS == 2 && mail.dst) {
        // exactly one lane found the value
        const unsigned long long m = __ballot(mine);
        const int src = m ? (int)__ffsll((long long)m) - 1 : 0;
        value = (unsigned)__shfl((int)value, src);
        if (lane == 0) {
            mail_store(&mail.dst[0], value);
            mail_drain();
            mail_publish(mail.seq_ptr, mail.seq);
        }
    }
}

__global__ void k_count_below(const float* __restrict__ d2, int n, float thr, unsigned* __restrict__ count) {
    unsigned c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        c += (sqrtf(d2[i]) < thr) ? 1u : 0u;     // R.cpp:607-608: float sqrt, strict <
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// mean number of points in the 27-cell stencil, weighted by the points of the centre cell
__global__ void k_kbar27(GridLevel g, unsigned long long* __restrict__ acc) {
    long long ncell = (long long)g.nx * g.ny * g.nz;
    unsigned long long s = 0;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < ncell;
         c += (long long)gridDim.x * blockDim.x) {
        int own = g.cell_start[c + 1] - g.cell_start[c];
        if (!own) continue;
        int cx = (int)(c % g.nx), cy = (int)((c / g.nx) % g.ny), cz = (int)(c / ((long long)g.nx * g.ny));
        unsigned tot = 0;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy) {
                int y = cy + dy, z = cz + dz;
                if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
                int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
                int row = (z * g.ny + y) * g.nx;
                tot += g.cell_start[row + x1 + 1] - g.cell_start[row + x0];
            }
        s += (unsigned long long)own * tot;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(acc, s);
}

}  // namespace

// =====================================================================================================
int pw_exclusive_scan(pwicp_context* ctx, int* d_data, long long n, DevBuf<int>* tmp) {
    if (n <= 0) return PWICP_OK;
    int tiles = (int)((n + kScanTile - 1) / kScanTile);
    HIPCHK(ctx, tmp->reserve((size_t)tiles + 1));
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(256), 0, ctx->stream, d_data, n, tmp->p);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, tmp->p, tiles);
    hipLaunchKernelGGL(k_scan_apply, dim3(tiles), dim3(256), 0, ctx->stream, d_data, n, tmp->p);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

namespace {

// one level: counting sort of the points by cell of edge h over the bounding box [mn, mx]
int build_level(pwicp_context* ctx, const float4* d_pts, int n, float h, const float* mn, const float* mx,
                GridLevel* d, DevBuf<int>* cell_start, DevBuf<float4>* pts, int flat_axis) {
    // cap the dense cell array at 2^28 cells (1 GiB of int32): coarser cells stay exact, only slower
    for (;;) {
        double cells = 1.0;
        for (int k = 0; k < 3; ++k)
            if (k != flat_axis || flat_axis == 0) cells *= std::floor((double)(mx[k] - mn[k]) / h) + 2.0;
        if (cells <= 268435456.0) break;
        h *= 1.26f;
    }
    d->n = n;
    d->perm = 0;
    d->h = h;
    d->inv_h = 1.0f / h;
    d->ox = mn[0]; d->oy = mn[1]; d->oz = mn[2];
    d->nx = (int)std::floor((mx[0] - mn[0]) * d->inv_h) + 1;
    d->ny = (int)std::floor((mx[1] - mn[1]) * d->inv_h) + 1;
    d->nz = (int)std::floor((mx[2] - mn[2]) * d->inv_h) + 1;
    d->inv_hy = d->inv_hz = d->inv_h;
    if (flat_axis == 1) { d->ny = 1; d->inv_hy = 0.0f; }       // columns along y
    if (flat_axis == 2) { d->nz = 1; d->inv_hz = 0.0f; }       // columns along z
    float maxabs = 0.f;
    for (int k = 0; k < 3; ++k) maxabs = std::max(maxabs, std::max(std::fabs(mn[k]), std::fabs(mx[k])));
    const int maxdim = std::max(d->nx, std::max(d->ny, d->nz));
    d->slack = h * 1.0e-6f * (float)(maxdim + 1) + 4.0f * FLT_EPSILON * maxabs;

    const long long ncell = (long long)d->nx * d->ny * d->nz;
    HIPCHK(ctx, cell_start->reserve((size_t)ncell + 1));
    HIPCHK(ctx, pts->reserve((size_t)n));
    HIPCHK(ctx, hipMemsetAsync(cell_start->p, 0, (size_t)(ncell + 1) * sizeof(int), ctx->stream));
    DevBuf<int> cell_id, rank, tmp;
    HIPCHK(ctx, cell_id.reserve((size_t)n));
    HIPCHK(ctx, rank.reserve((size_t)n));
    d->cell_start = cell_start->p;
    d->pts = pts->p;
    d->pts3 = nullptr;
    hipLaunchKernelGGL(k_cell_count, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_pts, n, *d,
                       cell_start->p, cell_id.p, rank.p);
    PWCHK(pw_exclusive_scan(ctx, cell_start->p, ncell + 1, &tmp));
    hipLaunchKernelGGL(k_cell_scatter, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_pts, n,
                       cell_start->p, cell_id.p, rank.p, pts->p);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // temporaries die here
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

}  // namespace

// axis-aligned bounding box of n > 0 device points (synchronises the stream)
int pw_bbox(pwicp_context* ctx, const float4* d_pts, int n, float mn[3], float mx[3]) {
    DevBuf<unsigned> bb;
    HIPCHK(ctx, bb.reserve(8));
    static const unsigned init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    HIPCHK(ctx, hipMemcpyAsync(bb.p, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    int nb = std::min(div_up(n, kBlock), ctx->n_cu * 4);
    hipLaunchKernelGGL(k_bbox, dim3(nb), dim3(kBlock), 0, ctx->stream, d_pts, n, bb.p);
    unsigned hb[8];
    HIPCHK(ctx, hipMemcpyAsync(hb, bb.p, sizeof(hb), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 3; ++k) { mn[k] = ord2f(hb[k]); mx[k] = ord2f(hb[3 + k]); }
    if (hb[6]) {
        ctx->set_err("non-finite coordinates in the cloud");
        return PWICP_E_INVALID;
    }
    for (int k = 0; k < 3; ++k)
        if (!(std::isfinite(mn[k]) && std::isfinite(mx[k]))) {
            ctx->set_err("non-finite coordinates in the cloud");
            return PWICP_E_INVALID;
        }
    return PWICP_OK;
}

// PWICP_E_INVALID if any coordinate of the n device points is NaN or infinite (queries as well as targets must be finite:
// a NaN query has no nearest neighbour)
int pw_check_finite(pwicp_context* ctx, const float4* d_pts, int n) {
    if (n <= 0) return PWICP_OK;
    float mn[3], mx[3];
    return pw_bbox(ctx, d_pts, n, mn, mx);
}

int pw_grid_build(pwicp_context* ctx, const float4* d_pts, int n, float cell_edge, Grid* g) {
    GridDesc& d = g->d;
    memset(&d, 0, sizeof(d));
    if (n <= 0) {
        HIPCHK(ctx, g->cell_start.reserve(2));
        HIPCHK(ctx, hipMemsetAsync(g->cell_start.p, 0, 2 * sizeof(int), ctx->stream));
        HIPCHK(ctx, g->pts.reserve(1));
        GridLevel e{};
        e.nx = e.ny = e.nz = 1; e.h = 1.f; e.inv_h = 1.f; e.inv_hy = 1.f; e.inv_hz = 1.f; e.n = 0;
        e.cell_start = g->cell_start.p; e.pts = g->pts.p;
        d.fine = e; d.coarse = e;
        return PWICP_OK;
    }
    float mn[3], mx[3];
    PWCHK(pw_bbox(ctx, d_pts, n, mn, mx));
    const float h = cell_edge > 0.f ? cell_edge : 1.f;
    // Layout of the levels: cells (3-D) or columns (cells in two axes, unbounded along y or z).  A search visits the
    // 3 x 3 rows around the query's cell; with columns that is 3 rows instead of 9 — for surface-like clouds whose
    // sheet is roughly normal to the collapsed axis the candidate set is the same, and a row costs about as much as
    // kRowCost candidates (measured on the dense 1-NN kernel: 9 rows + 79 candidates 107 us, 3 rows + 81 candidates
    // 73 us per 754 k queries).  The exactness bound r*h - 2*slack only uses the two gridded axes, so every layout
    // gives the same (d2, index) minimum; the choice is purely a cost estimate from the measured stencil occupancy.
    // PWICP_GRID_LAYOUT = 0 / 1 / 2 forces cells / y-columns / z-columns.
    constexpr double kRowCost = 8.0;
    static int forced = -2;
    if (forced == -2) { const char* e = getenv("PWICP_GRID_LAYOUT"); forced = e ? atoi(e) : -1; }
    auto kbar_of = [&](const GridLevel& lv, double* out) -> int {
        DevBuf<unsigned long long> acc;
        HIPCHK(ctx, acc.reserve(1));
        HIPCHK(ctx, hipMemsetAsync(acc.p, 0, sizeof(unsigned long long), ctx->stream));
        const long long ncell = (long long)lv.nx * lv.ny * lv.nz;
        int nbk = (int)std::min<long long>((ncell + kBlock - 1) / kBlock, (long long)ctx->n_cu * 16);
        hipLaunchKernelGGL(k_kbar27, dim3(nbk), dim3(kBlock), 0, ctx->stream, lv, acc.p);
        unsigned long long hacc = 0;
        HIPCHK(ctx, hipMemcpyAsync(&hacc, acc.p, sizeof(hacc), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *out = (double)hacc / (double)n;
        return PWICP_OK;
    };
    int best_axis = 0;
    double best_cost = 0.0;
    {
        // candidates: cells, and columns along the thinner of the y / z extents (x is the row direction)
        int cand[2] = {0, (mx[1] - mn[1]) < (mx[2] - mn[2]) ? 1 : 2};
        int ncand = 2;
        if (forced >= 0 && forced <= 2) { cand[0] = forced; ncand = 1; }
        for (int c = 0; c < ncand; ++c) {
            GridLevel lv{};
            DevBuf<int> cs;
            DevBuf<float4> ps;
            PWCHK(build_level(ctx, d_pts, n, h, mn, mx, &lv, &cs, &ps, cand[c]));
            double kb = 0.0;
            PWCHK(kbar_of(lv, &kb));
            const double cost = kRowCost * (cand[c] ? 3.0 : 9.0) + kb;
            if (c == 0 || cost < best_cost) {
                best_cost = cost;
                best_axis = cand[c];
                d.fine = lv;
                g->cell_start.swap(cs);
                g->pts.swap(ps);
                g->kbar27 = kb;
            }
        }
    }
    static float coarse_factor = 0.f;          // PWICP_COARSE_FACTOR: coarse edge / fine edge (default 2: measured best of 1.5 / 2 / 2.5 / 3 / 4)
    if (coarse_factor == 0.f) { const char* e = getenv("PWICP_COARSE_FACTOR"); coarse_factor = e ? (float)atof(e) : 2.0f; if (!(coarse_factor > 1.f)) coarse_factor = 2.0f; }
    PWCHK(build_level(ctx, d_pts, n, coarse_factor * d.fine.h, mn, mx, &d.coarse, &g->ccell_start, &g->cpts, best_axis));
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_nn_launch(pwicp_context* ctx, const GridDesc& g, const float4* d_q, int nq, int* d_idx, float* d_d2,
                 unsigned long long* d_examined) {
    if (nq <= 0) return PWICP_OK;
    // small launches are latency bound: spread each query over 8 lanes; large ones are throughput bound: one lane each
    if (!d_examined && nq <= 262144) {
        hipLaunchKernelGGL(k_nn_points_group, dim3(div_up((long long)nq * kGroup, kBlock)), dim3(kBlock), 0, ctx->stream, g,
                           d_q, nq, d_idx, d_d2);
        HIPCHK(ctx, hipGetLastError());
        return PWICP_OK;
    }
    hipLaunchKernelGGL(k_nn_points, dim3(div_up(nq, kBlock)), dim3(kBlock), 0, ctx->stream, g, d_q, nq, d_idx,
                       d_d2, d_examined);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_select_kth_launch(pwicp_context* ctx, const float* d_vals, int n, int k, unsigned* d_scratch, float* d_out,
                         bool armed, const SelectMail* mail) {
    if (n <= 0) return PWICP_E_INVALID;
    // few blocks: every block ends with one global atomic per non-empty bin, and same-address atomics serialise
    const int nbl = ctx->n_cu;
    int nb = std::min(div_up(n, kBlock), nbl);
    // `armed`: the scratch buffer was zeroed when it was allocated and only ever used by this function (every
    // selection leaves it zeroed again)
    if (!armed)
        hipLaunchKernelGGL(k_select_init, dim3(div_up(8 + 3 * kSelBins, kBlock)), dim3(kBlock), 0, ctx->stream, d_scratch, k);
    SelectMail none{};
    hipLaunchKernelGGL(k_select_pass<0>, dim3(nb), dim3(kBlock), 0, ctx->stream, d_vals, n, k, d_scratch, d_out, none);
    hipLaunchKernelGGL(k_select_pass<1>, dim3(nb), dim3(kBlock), 0, ctx->stream, d_vals, n, k, d_scratch, d_out, none);
    hipLaunchKernelGGL(k_select_pass<2>, dim3(nb), dim3(kBlock), 0, ctx->stream, d_vals, n, k, d_scratch, d_out,
                       mail ? *mail : none);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_count_below_launch(pwicp_context* ctx, const float* d_d2, int n, float thr, unsigned* d_count) {
    HIPCHK(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned), ctx->stream));
    if (n > 0) {
        int nb = std::min(div_up(n, kBlock), ctx->n_cu * 4);
        hipLaunchKernelGGL(k_count_below, dim3(nb), dim3(kBlock), 0, ctx->stream, d_d2, n, thr, d_count);
    }
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

// third level of small cells over the same points, same layout (cells / columns) as g->d.fine
namespace {
// (x, y, z) -> the axis order of a permuted level; w (the original index) is kept
// the points of a small-cell level once more, packed x, y, z (GridLevel::pts3)
__global__ void k_pack_xyz3(const float4* __restrict__ in, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + kPts3Pad + (n & 1)) return;
    // (behind the last point: far-away points - a scan may read past the end of its range, nn_device.h scan_d2_level)
    const float4 p = i < n ? in[i] : make_float4(1.0e18f, 1.0e18f, 1.0e18f, 0.f);
    out[3 * (size_t)i] = p.x; out[3 * (size_t)i + 1] = p.y; out[3 * (size_t)i + 2] = p.z;
}
int pack_level(pwicp_context* ctx, GridLevel* lv, DevBuf<float>* buf) {
    HIPCHK(ctx, buf->reserve(((size_t)lv->n + kPts3Pad + 2) * 3));
    hipLaunchKernelGGL(k_pack_xyz3, dim3(div_up(lv->n + kPts3Pad + 1, kBlock)), dim3(kBlock), 0, ctx->stream, lv->pts, lv->n, buf->p);
    lv->pts3 = buf->p;
    return PWICP_OK;
}
__global__ void k_permute_axes(const float4* __restrict__ in, int n, int perm, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    out[i] = perm == 1 ? make_float4(p.y, p.z, p.x, p.w) : make_float4(p.z, p.x, p.y, p.w);
}
}  // namespace

int pw_grid_add_dense(pwicp_context* ctx, const float4* d_pts, int n, float cell_edge, Grid* g) {
    g->has_dense = g->has_dense_alt = false;
    for (auto& x : g->extra) x.has = false;
    if (n <= 0 || !(cell_edge > 0.f)) return PWICP_OK;
    float mn[3], mx[3];
    PWCHK(pw_bbox(ctx, d_pts, n, mn, mx));
    const int axis = g->d.fine.ny == 1 && g->d.fine.inv_hy == 0.0f ? 1 : (g->d.fine.nz == 1 && g->d.fine.inv_hz == 0.0f ? 2 : 0);
    PWCHK(build_level(ctx, d_pts, n, cell_edge, mn, mx, &g->dense, &g->dcell_start, &g->dpts, axis));
    g->has_dense = true;
    PWCHK(pack_level(ctx, &g->dense, &g->dpts3));
    static int want_alt = -1;              // PWICP_DENSE_ALT=0: no level of columns beside a level of cells (A/B measurements)
    if (want_alt < 0) { const char* e = getenv("PWICP_DENSE_ALT"); want_alt = e ? atoi(e) : 1; }
    if (axis == 0 && want_alt) {
        // The target as a whole prefers cells (steep or volumetric parts: pw_grid_build).  The QUERIES of a pair may still
        // all lie where the cloud is a gentle sheet (patches on steep faces rarely survive the planarity gate), and there
        // a level of columns costs the disc search a fifth of the rows: keep one, the pair decides (pw_dense_level_for).
        const int alt = (mx[1] - mn[1]) < (mx[2] - mn[2]) ? 1 : 2;
        PWCHK(build_level(ctx, d_pts, n, cell_edge, mn, mx, &g->dense_alt, &g->acell_start, &g->apts, alt));
        g->has_dense_alt = true;
        PWCHK(pack_level(ctx, &g->dense_alt, &g->apts3));
        // ... and levels with other axis roles: columns along x (no layout on (x, y, z) has them: a face in the y-z plane), cells
        // with their rows along y / along z (a sheet tilted about that axis is contiguous along it).  Built on a copy of the
        // points in the permuted order; which one a pair uses is decided by a cost probe on its queries (pw_dense_level_for).
        static int want_perm = -1;         // PWICP_DENSE_PERM=0: none (A/B measurements)
        if (want_perm < 0) { const char* e = getenv("PWICP_DENSE_PERM"); want_perm = e ? atoi(e) : 1; }
        if (want_perm) {
            DevBuf<float4> tmp;
            HIPCHK(ctx, tmp.reserve((size_t)n));
            const int perm_of[3] = {1, 1, 2}, flat_of[3] = {2, 0, 0};       // columns along x | cells, rows along y | cells, rows along z
            for (int e = 0; e < 3; ++e) {
                const int perm = perm_of[e];
                hipLaunchKernelGGL(k_permute_axes, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_pts, n, perm, tmp.p);
                const float pmn[3] = {perm == 1 ? mn[1] : mn[2], perm == 1 ? mn[2] : mn[0], perm == 1 ? mn[0] : mn[1]};
                const float pmx[3] = {perm == 1 ? mx[1] : mx[2], perm == 1 ? mx[2] : mx[0], perm == 1 ? mx[0] : mx[1]};
                PWCHK(build_level(ctx, tmp.p, n, cell_edge, pmn, pmx, &g->extra[e].lv, &g->extra[e].cell_start, &g->extra[e].pts, flat_of[e]));
                g->extra[e].lv.perm = perm;
                g->extra[e].has = true;
                PWCHK(pack_level(ctx, &g->extra[e].lv, &g->extra[e].pts3));
            }
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // (tmp goes out of scope)
        }
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PWICP_OK;
}

namespace {
// Cost of the disc search on a level for the queries of a pair, on every `stride`-th query: candidates examined and rows
// whose begin / end words are read (the two things the search pays for), exactly as k_nn_dense_disc would do it.
template <int PERM>
__global__ void k_dense_probe(GridLevel dl, const float4* __restrict__ q4, int nq, int stride, unsigned long long* __restrict__ acc) {
    unsigned long long cand = 0, rows = 0, far = 0;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * stride; i < nq; i += gridDim.x * blockDim.x * stride) {
        const float4 q = q4[i];
        const float ux = PERM == 0 ? q.x : (PERM == 1 ? q.y : q.z);
        const float uy = PERM == 0 ? q.y : (PERM == 1 ? q.z : q.x);
        const float uz = PERM == 0 ? q.z : (PERM == 1 ? q.x : q.y);
        const int cx = cell_of(ux, dl.ox, dl.inv_h), cy = cell_of(uy, dl.oy, dl.inv_hy), cz = cell_of(uz, dl.oz, dl.inv_hz);
        int loA, hiA;
        row_range(dl, cy, cz, cx - 1, cx + 1, loA, hiA);
        float best = INFINITY;
        scan_d2_level<PERM>(dl, loA, hiA, ux, uy, uz, best);
        cand += (unsigned long long)(hiA - loA);
        rows += 1;
        const float rho = fast_sqrt_up(best) + 2.0f * dl.slack;
        if (best < INFINITY && rho <= kMaxRhoCells * dl.h) {
            const bool in = cy >= 0 && cy < dl.ny && cz >= 0 && cz < dl.nz && max(cx - 1, 0) <= min(cx + 1, dl.nx - 1);
            const unsigned c = scan_disc_lean<PERM, true, true>(dl, ux, uy, uz, rho, in ? cy : INT_MIN, in ? cz : INT_MIN, max(cx - 1, 0),
                                                          min(cx + 1, dl.nx - 1), loA, hiA, best);
            cand += c & 0xfffffu;
            rows += c >> 20;
        } else {
            far += 1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cand += __shfl_xor(cand, o); rows += __shfl_xor(rows, o); far += __shfl_xor(far, o); }
    if ((threadIdx.x & 63) == 0) {
        if (cand) atomicAdd(acc, cand);
        if (rows) atomicAdd(acc + 1, rows);
        if (far) atomicAdd(acc + 2, far);
    }
}
}  // namespace

namespace {
// points in the 3 x 3 (x 3) cells around every query, summed (the stencil occupancy the queries actually see)
__global__ void k_query_occupancy(GridLevel g, const float4* __restrict__ q, int nq, unsigned long long* __restrict__ acc) {
    unsigned long long s = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
        const float4 v = q[i];
        const int cx = cell_of(v.x, g.ox, g.inv_h), cy = cell_of(v.y, g.oy, g.inv_hy), cz = cell_of(v.z, g.oz, g.inv_hz);
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy) {
                if ((g.inv_hz == 0.0f && dz != 0) || (g.inv_hy == 0.0f && dy != 0)) continue;
                int lo, hi;
                row_range(g, cy + dy, cz + dz, cx - 1, cx + 1, lo, hi);
                s += (unsigned long long)(hi - lo);
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(acc, s);
}
}  // namespace

// The small-cell level the dense search of a pair should use: the target's layout, or — when that is cells and a level of
// columns exists — columns if the queries' stencils hold at most kColsTolerance times as many points there (rows are what
// the disc search pays for: 36 cell rows against 6 column rows per ball).  Synchronises the stream (one-off, at pair creation).
int pw_dense_level_for(pwicp_context* ctx, const Grid& g, const float4* d_q, int nq, const GridLevel** out, double* far_frac) {
    *out = g.has_dense ? &g.dense : nullptr;
    if (far_frac) *far_frac = 0.0;
    if (!g.has_dense || nq <= 0) return PWICP_OK;
    // candidates: the target's own layout (cells), the columns beside it, the levels with other axis roles.  Cost of a level =
    // candidates + kRowCost * rows + kFarCost * queries that leave the disc search, probed on ~16 k of the pair's queries.
    constexpr double kRowCost = 8.0, kFarCost = 40.0;   // (measured on the steep scene of tools/scene_shapes.py: a far query ~ 40-50 candidates)
    const GridLevel* cand[5] = {&g.dense, &g.dense_alt, nullptr, nullptr, nullptr};
    int nc = g.has_dense_alt ? 2 : 1;           // (a single level: the probe still tells how many queries start far)
    if (g.has_dense_alt)
        for (const auto& x : g.extra)
            if (x.has) cand[nc++] = &x.lv;
    DevBuf<unsigned long long> acc;
    HIPCHK(ctx, acc.reserve(15));
    HIPCHK(ctx, hipMemsetAsync(acc.p, 0, 15 * sizeof(unsigned long long), ctx->stream));
    const int stride = std::max(1, nq / 16384);
    const int nb = std::max(1, std::min(div_up(div_up(nq, stride), kBlock), ctx->n_cu * 4));
    for (int c = 0; c < nc; ++c) {
        const GridLevel& lv = *cand[c];
        if (lv.perm == 0) hipLaunchKernelGGL(k_dense_probe<0>, dim3(nb), dim3(kBlock), 0, ctx->stream, lv, d_q, nq, stride, acc.p + 3 * c);
        else if (lv.perm == 1) hipLaunchKernelGGL(k_dense_probe<1>, dim3(nb), dim3(kBlock), 0, ctx->stream, lv, d_q, nq, stride, acc.p + 3 * c);
        else hipLaunchKernelGGL(k_dense_probe<2>, dim3(nb), dim3(kBlock), 0, ctx->stream, lv, d_q, nq, stride, acc.p + 3 * c);
    }
    unsigned long long h[15];
    HIPCHK(ctx, hipMemcpyAsync(h, acc.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double best_cost = 0.0;
    const bool trace = getenv("PWICP_TRACE") != nullptr;
    for (int c = 0; c < nc; ++c) {
        const double cost = (double)h[3 * c] + kRowCost * (double)h[3 * c + 1] + kFarCost * (double)h[3 * c + 2];
        if (trace)
            fprintf(stderr, "[pwicp dense level] candidate %d (perm %d, %s): %llu candidates, %llu rows, %llu far -> cost %.3g\n", c, cand[c]->perm,
                    (cand[c]->inv_hy == 0.0f || cand[c]->inv_hz == 0.0f) ? "columns" : "cells", h[3 * c], h[3 * c + 1], h[3 * c + 2], cost);
        if (c == 0 || cost < best_cost) {
            best_cost = cost; *out = cand[c];
            if (far_frac) *far_frac = (double)h[3 * c + 2] / (double)std::max(div_up(nq, stride), 1);
        }
    }
    return PWICP_OK;
}

int pw_fs_pass_launch(pwicp_context* ctx, int pass, const FusedSelect& fs) {
    if (pass == 1) hipLaunchKernelGGL(k_fs_pass<1>, dim3(fs.nblk), dim3(kBlock), 0, ctx->stream, fs);
    else hipLaunchKernelGGL(k_fs_pass<2>, dim3(fs.nblk), dim3(kBlock), 0, ctx->stream, fs);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_gather_f4_launch(pwicp_context* ctx, const float4* d_src, const int* d_order, int n, float4* d_out) {
    if (n > 0) hipLaunchKernelGGL(k_gather_f4, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_src, d_order, n, d_out);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_gather_int_launch(pwicp_context* ctx, const int* d_src, const int* d_order, int n, int* d_out) {
    if (n > 0) hipLaunchKernelGGL(k_gather_int, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_src, d_order, n, d_out);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_nn_dense_launch(pwicp_context* ctx, const GridDesc& g, const float4* d_pat, const int* d_qorder,
                           const int* d_pt_patch, const int* d_stable, int nq, float* d_d2,
                           unsigned long long* d_examined, const GridLevel* dense, const int* d_qpatch, const FusedSelect* fs,
                           DenseFarBuffers* far_bufs, const float4* d_patq) {
    if (nq <= 0) return PWICP_OK;
    const bool far_group = far_bufs != nullptr;
    DenseFarList fl{};
    if (far_group) {
        HIPCHK(ctx, far_bufs->q.reserve((size_t)nq));
        HIPCHK(ctx, far_bufs->slot.reserve((size_t)nq));
        if (!far_bufs->count.p) {
            HIPCHK(ctx, far_bufs->count.reserve(2));
            HIPCHK(ctx, hipMemsetAsync(far_bufs->count.p, 0, 2 * sizeof(unsigned), ctx->stream));
        }
        fl.q = far_bufs->q.p; fl.slot = far_bufs->slot.p; fl.count = far_bufs->count.p;
        // (PWICP_DENSE_FAR_EDGE: 0 - every far query searched to the end; n - only lists of at least n far queries look at the bins)
        static const int far_edge = getenv("PWICP_DENSE_FAR_EDGE") ? std::max(atoi(getenv("PWICP_DENSE_FAR_EDGE")), 0) : 1;
        fl.edge = far_edge;
    }
    if (dense && d_qpatch && d_qorder) {
        const int tiles = div_up(nq, kDenseBlock);
        static int sub_env = -1;            // PWICP_DENSE_XCD_SUB: tiles per run dealt to an XCD (0: one contiguous eighth each)
        if (sub_env < 0) { const char* e = getenv("PWICP_DENSE_XCD_SUB"); sub_env = e ? std::max(atoi(e), 0) : 4; }
        int chunk = div_up(tiles, kXcds);
        const int sub = sub_env > 0 ? std::min(sub_env, chunk) : chunk;
        chunk = div_up(chunk, sub) * sub;
        FusedSelect none{};
#define PW_DENSE(PERM_, FARG_)                                                                                              \
    hipLaunchKernelGGL((k_nn_dense_disc<PERM_, FARG_>), dim3(chunk * kXcds), dim3(kDenseBlock), 0, ctx->stream, *dense, g, d_pat, d_qorder, \
                       d_qpatch, d_stable, nq, d_d2, d_examined, chunk, fs ? *fs : none, fl, d_patq, sub)
        if (far_group) { if (dense->perm == 0) PW_DENSE(0, true); else if (dense->perm == 1) PW_DENSE(1, true); else PW_DENSE(2, true); }
        else { if (dense->perm == 0) PW_DENSE(0, false); else if (dense->perm == 1) PW_DENSE(1, false); else PW_DENSE(2, false); }
#undef PW_DENSE
#ifdef PW_DENSE_SLOTS
        {
            if (dense->perm == 0) hipLaunchKernelGGL((k_dense_slots<0>), dim3(chunk * kXcds), dim3(kDenseBlock), 0, ctx->stream, *dense, d_pat, d_qorder, d_qpatch, d_stable, nq, chunk, d_patq, sub);
            else if (dense->perm == 1) hipLaunchKernelGGL((k_dense_slots<1>), dim3(chunk * kXcds), dim3(kDenseBlock), 0, ctx->stream, *dense, d_pat, d_qorder, d_qpatch, d_stable, nq, chunk, d_patq, sub);
            else hipLaunchKernelGGL((k_dense_slots<2>), dim3(chunk * kXcds), dim3(kDenseBlock), 0, ctx->stream, *dense, d_pat, d_qorder, d_qpatch, d_stable, nq, chunk, d_patq, sub);
        }
#endif
        if (far_group)
        {
            // blocks per CU: four, each walking its share of the list (measured on the reference's scans, 120 k far queries: loop
            // 0.465 / 0.437 / 0.434 / 0.438 / 0.443 / 0.462 / 0.533 ms with 1 / 2 / 3 / 4 / 6 / 8 / 16 - a block's epilogue, the
            // flush of its selection bins, costs more than a second and third pass over the list)
            constexpr int per_cu = 4;
            hipLaunchKernelGGL(k_nn_dense_far, dim3((unsigned)std::min(div_up((long long)nq * kGroup, kBlock), ctx->n_cu * per_cu)), dim3(kBlock), 0,
                               ctx->stream, g, fl, d_d2, fs ? *fs : none, d_examined);
        }
        HIPCHK(ctx, hipGetLastError());
        return PWICP_OK;
    }
    // no small-cell level (PWICP_DISC_CELL_FACTOR=0): the 27-cell stencil kernel with (d2, index) keys, kept for A/B runs
    {
        const int tiles = div_up(nq, kBlock);
        const int chunk = div_up(tiles, kXcds);
        hipLaunchKernelGGL(k_nn_dense_direct, dim3(chunk ? chunk * kXcds : tiles), dim3(kBlock), 0, ctx->stream, g, d_pat, d_qorder,
                           d_pt_patch, d_stable, nq, d_d2, d_examined, chunk);
    }
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

// permutation of 0..n-1 that lists the points in Morton order of their fine cell in grid g
int pw_morton_order(pwicp_context* ctx, const GridDesc& g, const float4* d_pts, int n, DevBuf<int>* order, const GridLevel* strip_lv) {
    HIPCHK(ctx, order->reserve((size_t)std::max(n, 1)));
    if (n <= 0) return PWICP_OK;
    // The queries of the dense search in STRIP order on the level the pair searches (k_strip_keys): blocks of 32 cells x 8 rows.
    // In Morton order (round 3) the 64 queries of a wave sit in ~4 x 4 cells of four rows and every gather of the wave touches
    // ~20 cache lines (TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD); along a row their windows tile one span of the row's
    // points.  41.3 -> 39.1 us.  PWICP_QUERY_ORDER="xb,rb" (0: Morton order of the fine cells).
    static int xb = -1, rb = 8;
    const int serp = 1;                     // rows of a block boustrophedon (a wave that crosses a row end stays in one place)
    if (xb < 0) {
        xb = 32;
        if (const char* e = getenv("PWICP_QUERY_ORDER")) { xb = std::max(atoi(e), 0); const char* c = strchr(e, ','); if (!c) c = strchr(e, ':'); if (c) rb = std::max(atoi(c + 1), 1); }
    }
    if (xb > 0 && strip_lv && (double)strip_lv->nx * strip_lv->ny * strip_lv->nz * 1.1 + (double)xb * rb * 2 < 4.0e9) {
        DevBuf<unsigned> keys, keys_out;
        DevBuf<int> vals;
        HIPCHK(ctx, keys.reserve((size_t)n));
        HIPCHK(ctx, keys_out.reserve((size_t)n));
        HIPCHK(ctx, vals.reserve((size_t)n));
        hipLaunchKernelGGL(k_strip_keys, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, *strip_lv, d_pts, n, xb, rb, serp, keys.p, vals.p);
        size_t tbytes = 0;
        HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tbytes, keys.p, keys_out.p, vals.p, order->p, n, 0, 32, ctx->stream));
        DevBuf<unsigned char> tmp;
        HIPCHK(ctx, tmp.reserve(tbytes));
        HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tbytes, keys.p, keys_out.p, vals.p, order->p, n, 0, 32, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipGetLastError());
        return PWICP_OK;
    }
    int shift = 0;
    while (((std::max(g.fine.nx, std::max(g.fine.ny, g.fine.nz)) - 1) >> shift) > 1023) ++shift;
    DevBuf<unsigned> keys, keys_out;
    DevBuf<int> vals;
    HIPCHK(ctx, keys.reserve((size_t)n));
    HIPCHK(ctx, keys_out.reserve((size_t)n));
    HIPCHK(ctx, vals.reserve((size_t)n));
    hipLaunchKernelGGL(k_morton_keys, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, g.fine, d_pts, n, shift,
                       keys.p, vals.p);
    size_t tbytes = 0;
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tbytes, keys.p, keys_out.p, vals.p, order->p, n, 0, 30,
                                                   ctx->stream));
    DevBuf<unsigned char> tmp;
    HIPCHK(ctx, tmp.reserve(tbytes));
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tbytes, keys.p, keys_out.p, vals.p, order->p, n, 0, 30,
                                                   ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

// k nearest neighbours of every point of the cloud within the cloud (row i of d_nb = neighbours of point i)
// The same search with the lane's sorted candidate list in LDS (entry e of lane l at [e * 64 + l]: conflict free), for
// k <= KMAX.  The list is where the time goes (a candidate shifts ~10 entries on average, ~100 candidates per point): in
// global memory that is ~24 KB of traffic per point, in LDS it is a handful of ds_read / ds_write per shift.  One
// wavefront per block; 36.9 KB (double) / 24.6 KB (float) of LDS per block.
template <typename Real, int KMAX>
__global__ void __launch_bounds__(64) k_knn_lds(GridLevel g, int k, int* __restrict__ out_nb, float* __restrict__ out_mean,
                                                int* __restrict__ rev_count = nullptr) {
    __shared__ Real s_d[KMAX * 64];
    __shared__ int s_i[KMAX * 64];
    const int lane = threadIdx.x;
    const int t = blockIdx.x * 64 + lane;
    const int n = g.n;
    if (t >= n) return;
    Real* const nd = s_d + lane;
    int* const ni = s_i + lane;
    const float4 q = g.pts[t];
    const int self = __float_as_int(q.w);
    const Real qx = (Real)q.x, qy = (Real)q.y, qz = (Real)q.z;
    const int cx = cell_of(q.x, g.ox, g.inv_h), cy = cell_of(q.y, g.oy, g.inv_hy), cz = cell_of(q.z, g.oz, g.inv_hz);
    const int rcover = max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz));
    const int ry_max = g.inv_hy != 0.0f ? INT_MAX : 0, rz_max = g.inv_hz != 0.0f ? INT_MAX : 0;     // collapsed axes have one row
    int cnt = 0;
    Real worst = (Real)0;
    int worst_i = 0;
    // one candidate into the sorted list (by (d2, id); the current k-th entry is kept in registers)
    auto consider = [&](const float4 p) {
        const Real dx = qx - (Real)p.x, dy2 = qy - (Real)p.y, dz2 = qz - (Real)p.z;
        Real d2 = dx * dx;
        d2 = d2 + dy2 * dy2;
        d2 = d2 + dz2 * dz2;
        const int id = __float_as_int(p.w);
        if (cnt == k && !(d2 < worst || (d2 == worst && id < worst_i))) return;
        int pos = cnt < k ? cnt : k - 1;
#ifndef PW_KNN_SHIFT1
        // four entries per step: their eight LDS reads are in flight together, then the (up to) four moves - one entry per step
        // is a read -> compare -> write chain that pays the LDS latency ~10 times per candidate at one wavefront per SIMD.
        // (The list ascends, so "the candidate comes before entry e" holds for a prefix of the four.)
        while (pos >= 4) {
            const Real p1 = nd[(pos - 1) * 64], p2 = nd[(pos - 2) * 64], p3 = nd[(pos - 3) * 64], p4 = nd[(pos - 4) * 64];
            const int i1 = ni[(pos - 1) * 64], i2 = ni[(pos - 2) * 64], i3 = ni[(pos - 3) * 64], i4 = ni[(pos - 4) * 64];
            const bool c1 = d2 < p1 || (d2 == p1 && id < i1);
            if (!c1) break;
            const bool c2 = d2 < p2 || (d2 == p2 && id < i2);
            const bool c3 = c2 && (d2 < p3 || (d2 == p3 && id < i3));
            const bool c4 = c3 && (d2 < p4 || (d2 == p4 && id < i4));
            nd[pos * 64] = p1; ni[pos * 64] = i1;
            if (c2) { nd[(pos - 1) * 64] = p2; ni[(pos - 1) * 64] = i2; }
            if (c3) { nd[(pos - 2) * 64] = p3; ni[(pos - 2) * 64] = i3; }
            if (c4) { nd[(pos - 3) * 64] = p4; ni[(pos - 3) * 64] = i4; }
            const int moved = 1 + (c2 ? 1 : 0) + (c3 ? 1 : 0) + (c4 ? 1 : 0);
            pos -= moved;
            if (moved < 4) goto placed;
        }
#endif
        while (pos > 0) {
            const Real pd = nd[(pos - 1) * 64];
            const int pi = ni[(pos - 1) * 64];
            if (!(d2 < pd || (d2 == pd && id < pi))) break;
            nd[pos * 64] = pd;
            ni[pos * 64] = pi;
            --pos;
        }
#ifndef PW_KNN_SHIFT1
    placed:
#endif
        nd[pos * 64] = d2;
        ni[pos * 64] = id;
        if (cnt < k) ++cnt;
        if (cnt == k) { worst = nd[(k - 1) * 64]; worst_i = ni[(k - 1) * 64]; }
    };
    // the points of a row segment, four loads in flight (one wavefront per SIMD: a load per candidate was a round trip per candidate)
    auto scan = [&](int lo, int hi) {
        int j = lo;
#ifndef PW_KNN_SHIFT1
        for (; j + 4 <= hi; j += 4) {
            const float4 a = g.pts[j], b = g.pts[j + 1], c = g.pts[j + 2], d = g.pts[j + 3];
            consider(a); consider(b); consider(c); consider(d);
        }
#endif
        for (; j < hi; ++j) consider(g.pts[j]);
    };
    // The block of cells grows shell by shell and the list is kept: every cell is scanned once (the search with its list in
    // global memory restarts at every radius: the slowest lane of a wavefront - a corner of the cloud, a hole: radius 6 to 12
    // cells - made its whole wavefront rescan (2r+1)^2 or ^3 cells for every r on the way, a floor of ~4 ms per launch).
    for (int r = 0;; ++r) {
        const int rz = min(r, rz_max), ry = min(r, ry_max);
        for (int dz = -rz; dz <= rz; ++dz)
            for (int dy = -ry; dy <= ry; ++dy) {
                int lo, hi;
                if (max(abs(dy), abs(dz)) == r) {             // a row that no smaller block had: its whole x-range
                    row_range(g, cy + dy, cz + dz, cx - r, cx + r, lo, hi);
                    scan(lo, hi);
                } else {                                        // a row of the previous block: its two new end cells
                    row_range(g, cy + dy, cz + dz, cx - r, cx - r, lo, hi);
                    scan(lo, hi);
                    row_range(g, cy + dy, cz + dz, cx + r, cx + r, lo, hi);
                    scan(lo, hi);
                }
            }
        if (r >= rcover) break;
        if (cnt == k && r >= 1) {
            const double bound = (double)r * (double)g.h - 2.0 * (double)g.slack;
            if (bound > 0.0 && (double)worst < bound * bound * 0.99999) break;
        }
    }
    if (out_nb)
        for (int e = 0; e < k; ++e) out_nb[(size_t)self * k + e] = (e < cnt) ? ni[e * 64] : -1;
    // (how often every point occurs in the rows: the sizes of the graph's reverse index, which the front end's fusion builds next -
    // 45 M atomics that disappear behind this kernel's LDS round trips instead of a 0.8 ms pass of their own)
    if (rev_count)
        for (int e = 0; e < cnt; ++e) atomicAdd(&rev_count[ni[e * 64]], 1);
    if (out_mean) {
        double s = 0.0;
        for (int e = 1; e < cnt; ++e) s += (double)sqrtf((float)nd[e * 64]);
        out_mean[self] = (float)(s / (double)(k - 1));
    }
}
constexpr int kKnnLdsMax = 48;

__global__ void k_knn_count_rows(const int* __restrict__ nb, long long m, int* __restrict__ rev_count) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m && nb[t] >= 0) atomicAdd(&rev_count[nb[t]], 1);
}
// (d_rev_count, optional: [n] zeroed by the caller; on return - in stream order - entry x is the number of rows that hold point x)
int pw_knn_launch(pwicp_context* ctx, const GridDesc& g, int k, int* d_nb, int* d_rev_count) {
    const int n = g.fine.n;
    if (n <= 0) return PWICP_OK;
    static const bool lds_off = getenv("PWICP_KNN_LDS") && atoi(getenv("PWICP_KNN_LDS")) == 0;      // A/B knob
    if (k <= kKnnLdsMax && !lds_off) {
        hipLaunchKernelGGL((k_knn_lds<double, kKnnLdsMax>), dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, g.fine, k, d_nb, (float*)nullptr,
                           d_rev_count);
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipGetLastError());
        return PWICP_OK;
    }
    DevBuf<double> nd;
    DevBuf<int> ni;
    HIPCHK(ctx, nd.reserve((size_t)n * k));
    HIPCHK(ctx, ni.reserve((size_t)n * k));
    hipLaunchKernelGGL(k_knn<double>, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, g.fine, k, nd.p, ni.p, d_nb,
                       (float*)nullptr);
    if (d_rev_count)
        hipLaunchKernelGGL(k_knn_count_rows, dim3((unsigned)div_up((long long)n * k, 256)), dim3(256), 0, ctx->stream, (const int*)d_nb,
                           (long long)n * k, d_rev_count);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

// mean distance of every point to its mean_k nearest OTHER points (float metric), d_mean[i] for point i
int pw_knn_mean_dist_launch(pwicp_context* ctx, const GridDesc& g, int mean_k, float* d_mean) {
    const int n = g.fine.n;
    if (n <= 0) return PWICP_OK;
    const int k = mean_k + 1;
    static const bool lds_off = getenv("PWICP_KNN_LDS") && atoi(getenv("PWICP_KNN_LDS")) == 0;
    if (k <= 16 && !lds_off) {
        // (the outlier removal's 14 + 1 neighbours: a 16-entry list is 8 KB of LDS per wavefront instead of 24.6 - five times the
        // wavefronts per CU for a search that is a chain of LDS round trips)
        hipLaunchKernelGGL((k_knn_lds<float, 16>), dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, g.fine, k, (int*)nullptr, d_mean);
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipGetLastError());
        return PWICP_OK;
    }
    if (k <= kKnnLdsMax && !lds_off) {
        hipLaunchKernelGGL((k_knn_lds<float, kKnnLdsMax>), dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, g.fine, k, (int*)nullptr, d_mean);
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipGetLastError());
        return PWICP_OK;
    }
    DevBuf<float> nd;
    DevBuf<int> ni;
    HIPCHK(ctx, nd.reserve((size_t)n * k));
    HIPCHK(ctx, ni.reserve((size_t)n * k));
    hipLaunchKernelGGL(k_knn<float>, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, g.fine, k, nd.p, ni.p,
                       (int*)nullptr, d_mean);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}
