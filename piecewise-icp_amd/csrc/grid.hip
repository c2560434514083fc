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
#include <cmath>

#include "common.h"
#include "nn_device.h"

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
// out[0..2] = min (ordered-uint encoded), out[3..5] = max, out[6] = max |coord|
__global__ void k_bbox(const float4* __restrict__ p, int n, unsigned* __restrict__ out) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 v = p[i];
        mn[0] = fminf(mn[0], v.x); mx[0] = fmaxf(mx[0], v.x);
        mn[1] = fminf(mn[1], v.y); mx[1] = fmaxf(mx[1], v.y);
        mn[2] = fminf(mn[2], v.z); mx[2] = fmaxf(mx[2], v.z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], o));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            atomicMin(&out[d], f2ord(mn[d]));
            atomicMax(&out[3 + d], f2ord(mx[d]));
        }
    }
}

// ---- counting sort by cell -----------------------------------------------------------------------
__global__ void k_cell_count(const float4* __restrict__ p, int n, GridDesc g, int* __restrict__ cnt,
                             int* __restrict__ cell_id, int* __restrict__ rank) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = p[i];
    int cx = min(max(cell_of(v.x, g.ox, g.inv_h), 0), g.nx - 1);
    int cy = min(max(cell_of(v.y, g.oy, g.inv_h), 0), g.ny - 1);
    int cz = min(max(cell_of(v.z, g.oz, g.inv_h), 0), g.nz - 1);
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

__global__ void __launch_bounds__(kBlock) k_nn_points(GridDesc g, const float4* __restrict__ q, int nq,
                                                      int* __restrict__ idx, float* __restrict__ d2,
                                                      unsigned long long* __restrict__ examined) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned cnt = 0;
    if (i < nq) {
        float4 v = q[i];
        NNBest b = nn_query(g, v.x, v.y, v.z, cnt);
        if (idx) idx[i] = b.idx;
        d2[i] = b.d2;
    }
    add_examined(examined, cnt);
}

__global__ void __launch_bounds__(kBlock) k_nn_patches(GridDesc g, const float4* __restrict__ pat,
                                                       const int* __restrict__ off, const int* __restrict__ list,
                                                       const int* __restrict__ soff, int n_list, int n_pts,
                                                       float* __restrict__ d2,
                                                       unsigned long long* __restrict__ examined) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned cnt = 0;
    if (i < n_pts) {
        // largest j with soff[j] <= i
        int lo = 0, hi = n_list;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (soff[mid] <= i) lo = mid; else hi = mid;
        }
        float4 v = pat[off[list[lo]] + (i - soff[lo])];
        NNBest b = nn_query(g, v.x, v.y, v.z, cnt);
        d2[i] = b.d2;
    }
    add_examined(examined, cnt);
}

// ---- k-th smallest of non-negative floats: 3-pass radix select on the bit pattern ---------------------
// scratch layout: [0] prefix, [1] k remaining, [8 + pass*2048 ...] histograms
constexpr int kSelBins = 2048;

__global__ void k_select_init(unsigned* scratch, int k) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 8 + 3 * kSelBins) scratch[t] = (t == 1) ? (unsigned)k : 0u;
}

template <int PASS>
__global__ void k_select_hist(const float* __restrict__ v, int n, unsigned* __restrict__ scratch) {
    __shared__ unsigned h[kSelBins];
    for (int t = threadIdx.x; t < kSelBins; t += blockDim.x) h[t] = 0;
    __syncthreads();
    const unsigned prefix = scratch[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned u = __float_as_uint(v[i]);
        if (PASS == 0) {
            atomicAdd(&h[u >> 21], 1u);
        } else if (PASS == 1) {
            if ((u >> 21) == (prefix >> 21)) atomicAdd(&h[(u >> 10) & 2047u], 1u);
        } else {
            if ((u >> 10) == (prefix >> 10)) atomicAdd(&h[u & 1023u], 1u);
        }
    }
    __syncthreads();
    unsigned* gh = scratch + 8 + PASS * kSelBins;
    for (int t = threadIdx.x; t < kSelBins; t += blockDim.x)
        if (h[t]) atomicAdd(&gh[t], h[t]);
}

template <int PASS>
__global__ void k_select_pick(unsigned* __restrict__ scratch, float* __restrict__ out) {
    // one wave: each lane owns 32 consecutive bins
    const unsigned* gh = scratch + 8 + PASS * kSelBins;
    int lane = threadIdx.x;
    unsigned local = 0;
    for (int b = 0; b < 32; ++b) local += gh[lane * 32 + b];
    unsigned incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    unsigned excl = incl - local;
    unsigned k = scratch[1];
    bool mine = (k >= excl) && (k < incl);
    if (mine) {
        unsigned run = excl;
        for (int b = 0; b < 32; ++b) {
            unsigned c = gh[lane * 32 + b];
            if (k < run + c) {
                unsigned bin = (unsigned)(lane * 32 + b);
                unsigned prefix = scratch[0];
                if (PASS == 0) prefix = bin << 21;
                else if (PASS == 1) prefix |= bin << 10;
                else prefix |= bin;
                scratch[0] = prefix;
                scratch[1] = k - run;
                if (PASS == 2) out[0] = __uint_as_float(prefix);
                break;
            }
            run += c;
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
__global__ void k_kbar27(GridDesc g, unsigned long long* __restrict__ acc) {
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

int pw_grid_build(pwicp_context* ctx, const float4* d_pts, int n, float cell_edge, Grid* g) {
    GridDesc& d = g->d;
    memset(&d, 0, sizeof(d));
    d.n = n;
    if (n <= 0) {
        d.nx = d.ny = d.nz = 1; d.h = 1.f; d.inv_h = 1.f;
        HIPCHK(ctx, g->cell_start.reserve(2));
        HIPCHK(ctx, hipMemsetAsync(g->cell_start.p, 0, 2 * sizeof(int), ctx->stream));
        HIPCHK(ctx, g->pts.reserve(1));
        d.cell_start = g->cell_start.p; d.pts = g->pts.p;
        return PWICP_OK;
    }
    DevBuf<unsigned> bb;
    HIPCHK(ctx, bb.reserve(8));
    unsigned init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    HIPCHK(ctx, hipMemcpyAsync(bb.p, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    int nb = std::min(div_up(n, kBlock), ctx->n_cu * 8);
    hipLaunchKernelGGL(k_bbox, dim3(nb), dim3(kBlock), 0, ctx->stream, d_pts, n, bb.p);
    unsigned hb[8];
    HIPCHK(ctx, hipMemcpyAsync(hb, bb.p, sizeof(hb), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) { mn[k] = ord2f(hb[k]); mx[k] = ord2f(hb[3 + k]); }
    for (int k = 0; k < 3; ++k)
        if (!(std::isfinite(mn[k]) && std::isfinite(mx[k]))) {
            ctx->set_err("pw_grid_build: non-finite coordinates in target cloud");
            return PWICP_E_INVALID;
        }
    float h = cell_edge > 0.f ? cell_edge : 1.f;
    // cap the dense cell array at 2^28 cells (1 GiB of int32): coarser cells stay exact, only slower
    for (;;) {
        double cells = 1.0;
        for (int k = 0; k < 3; ++k) cells *= std::floor((double)(mx[k] - mn[k]) / h) + 2.0;
        if (cells <= 268435456.0) break;
        h *= 1.26f;
    }
    d.h = h;
    d.inv_h = 1.0f / h;
    d.ox = mn[0]; d.oy = mn[1]; d.oz = mn[2];
    d.nx = (int)std::floor((mx[0] - mn[0]) * d.inv_h) + 1;
    d.ny = (int)std::floor((mx[1] - mn[1]) * d.inv_h) + 1;
    d.nz = (int)std::floor((mx[2] - mn[2]) * d.inv_h) + 1;
    float maxabs = 0.f;
    for (int k = 0; k < 3; ++k) maxabs = std::max(maxabs, std::max(std::fabs(mn[k]), std::fabs(mx[k])));
    int maxdim = std::max(d.nx, std::max(d.ny, d.nz));
    d.slack = h * 1.0e-6f * (float)(maxdim + 1) + 4.0f * FLT_EPSILON * maxabs;

    long long ncell = (long long)d.nx * d.ny * d.nz;
    HIPCHK(ctx, g->cell_start.reserve((size_t)ncell + 1));
    HIPCHK(ctx, g->pts.reserve((size_t)n));
    HIPCHK(ctx, hipMemsetAsync(g->cell_start.p, 0, (size_t)(ncell + 1) * sizeof(int), ctx->stream));
    DevBuf<int> cell_id, rank, tmp;
    HIPCHK(ctx, cell_id.reserve((size_t)n));
    HIPCHK(ctx, rank.reserve((size_t)n));
    d.cell_start = g->cell_start.p;
    d.pts = g->pts.p;
    hipLaunchKernelGGL(k_cell_count, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_pts, n, d,
                       g->cell_start.p, cell_id.p, rank.p);
    PWCHK(pw_exclusive_scan(ctx, g->cell_start.p, ncell + 1, &tmp));
    hipLaunchKernelGGL(k_cell_scatter, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_pts, n,
                       g->cell_start.p, cell_id.p, rank.p, g->pts.p);
    // Kbar of the 27-cell stencil (reported with every run)
    DevBuf<unsigned long long> acc;
    HIPCHK(ctx, acc.reserve(1));
    HIPCHK(ctx, hipMemsetAsync(acc.p, 0, sizeof(unsigned long long), ctx->stream));
    int nbk = (int)std::min<long long>((ncell + kBlock - 1) / kBlock, (long long)ctx->n_cu * 16);
    hipLaunchKernelGGL(k_kbar27, dim3(nbk), dim3(kBlock), 0, ctx->stream, d, acc.p);
    unsigned long long hacc = 0;
    HIPCHK(ctx, hipMemcpyAsync(&hacc, acc.p, sizeof(hacc), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    g->kbar27 = (double)hacc / (double)n;
    return PWICP_OK;
}

int pw_nn_launch(pwicp_context* ctx, const GridDesc& g, const float4* d_q, int nq, int* d_idx, float* d_d2,
                 unsigned long long* d_examined) {
    if (nq <= 0) return PWICP_OK;
    hipLaunchKernelGGL(k_nn_points, dim3(div_up(nq, kBlock)), dim3(kBlock), 0, ctx->stream, g, d_q, nq, d_idx,
                       d_d2, d_examined);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_nn_patches_launch(pwicp_context* ctx, const GridDesc& g, const float4* d_pat, const int* d_off,
                         const int* d_list, const int* d_soff, int n_list, int n_pts, float* d_d2,
                         unsigned long long* d_examined) {
    if (n_pts <= 0) return PWICP_OK;
    hipLaunchKernelGGL(k_nn_patches, dim3(div_up(n_pts, kBlock)), dim3(kBlock), 0, ctx->stream, g, d_pat, d_off,
                       d_list, d_soff, n_list, n_pts, d_d2, d_examined);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_select_kth_launch(pwicp_context* ctx, const float* d_vals, int n, int k, unsigned* d_scratch, float* d_out) {
    if (n <= 0) return PWICP_E_INVALID;
    int nb = std::min(div_up(n, kBlock), ctx->n_cu * 4);
    hipLaunchKernelGGL(k_select_init, dim3(div_up(8 + 3 * kSelBins, kBlock)), dim3(kBlock), 0, ctx->stream,
                       d_scratch, k);
    hipLaunchKernelGGL(k_select_hist<0>, dim3(nb), dim3(kBlock), 0, ctx->stream, d_vals, n, d_scratch);
    hipLaunchKernelGGL(k_select_pick<0>, dim3(1), dim3(64), 0, ctx->stream, d_scratch, d_out);
    hipLaunchKernelGGL(k_select_hist<1>, dim3(nb), dim3(kBlock), 0, ctx->stream, d_vals, n, d_scratch);
    hipLaunchKernelGGL(k_select_pick<1>, dim3(1), dim3(64), 0, ctx->stream, d_scratch, d_out);
    hipLaunchKernelGGL(k_select_hist<2>, dim3(nb), dim3(kBlock), 0, ctx->stream, d_vals, n, d_scratch);
    hipLaunchKernelGGL(k_select_pick<2>, dim3(1), dim3(64), 0, ctx->stream, d_scratch, d_out);
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
