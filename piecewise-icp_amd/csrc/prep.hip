// Preprocessing on the GPU: voxel-grid down-sampling + statistical outlier removal.
//
// Reference: PCpreprocessing src/CommonFunc.cpp:423-439 -> pcl::VoxelGrid (filters/impl/voxel_grid.hpp applyFilter:
// ijk = floor(p * inverse_leaf) - min_b, linear index i + j*dx + k*dx*dy, float centroid per occupied voxel, output
// in ascending index order) and SORfilter CommonFunc.cpp:442-452 -> pcl::StatisticalOutlierRemoval (mean distance to
// the k nearest other points, global mean + sample stddev, keep d <= mean + mult*stddev).
// Points of a voxel are summed in the order pcl::VoxelGrid's std::sort leaves them in for the reference's released build
// (host/msvc_sort.h: a sequential procedure, run on the host on the 4-byte keys while the device sorts and scans), or in
// input order with PWICP_VOXEL_ORDER=input - exactly like the host version (host/preprocess.cpp).
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cmath>
#include <vector>

#include "common.h"
#include "../host/io.h"
#include "../host/preprocess.h"

namespace {

constexpr int kBlock = 256;

struct VgParams {
    float inv;
    int minb0, minb1, minb2, mul1, mul2;
};

__global__ void k_vg_keys(const float4* __restrict__ p, int n, VgParams v, unsigned* __restrict__ keys, int* __restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = p[i];
    const int i0 = (int)(floorf(q.x * v.inv) - (float)v.minb0);
    const int i1 = (int)(floorf(q.y * v.inv) - (float)v.minb1);
    const int i2 = (int)(floorf(q.z * v.inv) - (float)v.minb2);
    keys[i] = (unsigned)(i0 + i1 * v.mul1 + i2 * v.mul2);
    vals[i] = i;
}

__global__ void k_vg_heads(const unsigned* __restrict__ keys, int n, int* __restrict__ head) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
    if (i == n) head[n] = 0;
}

// head scanned exclusively -> vid[i] = voxel of sorted entry i (for heads) ; start[voxel] = i
__global__ void k_vg_starts(const unsigned* __restrict__ keys, const int* __restrict__ vid, int n, int* __restrict__ start) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0 || keys[i] != keys[i - 1]) start[vid[i]] = i;
}

// one lane per voxel: float sums in input order, divided by the count (pcl::CentroidPoint / AccumulatorXYZ)
__global__ void k_vg_centroids(const float4* __restrict__ p, const int* __restrict__ order, const int* __restrict__ start, int m,
                               int n, float4* __restrict__ out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m) return;
    const int lo = start[v], hi = (v + 1 < m) ? start[v + 1] : n;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int j = lo; j < hi; ++j) {
        const float4 q = p[order[j]];
        c0 += q.x; c1 += q.y; c2 += q.z;
    }
    const float cnt = (float)(hi - lo);
    out[v] = make_float4(c0 / cnt, c1 / cnt, c2 / cnt, 1.0f);
}

__global__ void k_sor_flags(const float* __restrict__ d, int n, double thr, int* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keep[i] = ((double)d[i] > thr) ? 0 : 1;
    if (i == n) keep[n] = 0;
}

__global__ void k_sor_emit(const float4* __restrict__ p, const float* __restrict__ d, int n, double thr,
                           const int* __restrict__ pos, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !((double)d[i] > thr)) out[pos[i]] = p[i];
}

}  // namespace

// pcl::VoxelGrid's index-overflow test (filters/impl/voxel_grid.hpp, PCL 1.8.1): per axis (int64)((max - min) * inverse_leaf)
// + 1 in float arithmetic, product > INT_MAX -> warning, and the filter hands the INPUT cloud through unchanged.
static bool voxel_index_overflows(const float* mn, const float* mx, float inv) {
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1,
                    dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    return (double)dx * (double)dy * (double)dz > 2147483647.0;
}

// d_in: n points on the device.  d_out receives the filtered cloud (m points), *m_out its size.
// downsample = false: PCpreprocessing(..., isDownSamp = false, ...) = SORfilter alone (C.cpp:436-452); `leaf` then only
// sizes the search grid (<= 0: estimated from the cloud).
int pw_preprocess_dev(pwicp_context* ctx, const float4* d_in, int n, bool downsample, float leaf, int sor_k, double sor_mult,
                      DevBuf<float4>* d_out, int* m_out) {
    *m_out = 0;
    if (n <= 0) return PWICP_OK;
    pwhost::StageTimer tm;                  // PWICP_TRACE=1
    float mn[3], mx[3];
    PWCHK(pw_bbox(ctx, d_in, n, mn, mx));
    DevBuf<float4> vox;
    DevBuf<int> tmp;
    const float4* sor_in = d_in;
    int m = n;
    if (downsample && voxel_index_overflows(mn, mx, 1.0f / leaf)) {
        fprintf(stderr, "[pwicp] Leaf size is too small for the input dataset. Integer indices would overflow: the cloud is passed on "
                        "unfiltered (pcl::VoxelGrid semantics).\n");
        downsample = false;
        leaf = 0.f;                        // says nothing about the point spacing: the search grid edge is estimated below
    }
    if (!downsample && !(leaf > 0.f)) {
        // grid edge for the SOR search only (any edge is exact): ~2 mean spacings of a surface-like cloud in its bounding box
        const double ex = (double)mx[0] - mn[0], ey = (double)mx[1] - mn[1], ez = (double)mx[2] - mn[2];
        const double a = std::max(ex * ey, std::max(ex * ez, ey * ez));
        leaf = (float)std::max(std::sqrt(std::max(a, 1e-12) / (double)n), 1e-6);
    }
    if (downsample) {
    // ---- voxel grid -----------------------------------------------------------------------------------------------
    const float inv = 1.0f / leaf;
    VgParams v;
    v.inv = inv;
    int divb[3];
    const int minb[3] = {(int)std::floor(mn[0] * inv), (int)std::floor(mn[1] * inv), (int)std::floor(mn[2] * inv)};
    for (int k = 0; k < 3; ++k) divb[k] = (int)std::floor(mx[k] * inv) - minb[k] + 1;
    if ((double)divb[0] * divb[1] * divb[2] > 2147483647.0) {      // (unreachable after the test above save for rounding at the edge)
        ctx->set_err("pwicp_preprocess: leaf size too small for the cloud extent (voxel index overflows int32)");
        return PWICP_E_INVALID;
    }
    v.minb0 = minb[0]; v.minb1 = minb[1]; v.minb2 = minb[2];
    v.mul1 = divb[0]; v.mul2 = divb[0] * divb[1];
    DevBuf<unsigned> keys, keys_s;
    DevBuf<int> vals, order, head, start;
    HIPCHK(ctx, keys.reserve((size_t)n));
    HIPCHK(ctx, keys_s.reserve((size_t)n));
    HIPCHK(ctx, vals.reserve((size_t)n));
    HIPCHK(ctx, order.reserve((size_t)n));
    HIPCHK(ctx, head.reserve((size_t)n + 1));
    hipLaunchKernelGGL(k_vg_keys, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_in, n, v, keys.p, vals.p);
    int end_bit = 1;
    while (end_bit < 32 && (1ll << end_bit) < (long long)divb[0] * divb[1] * divb[2]) ++end_bit;
    size_t tb = 0;
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys.p, keys_s.p, vals.p, order.p, n, 0, end_bit, ctx->stream));
    DevBuf<unsigned char> tsort;
    HIPCHK(ctx, tsort.reserve(tb));
    // the reference build's order inside a voxel: keys to the host (4 B per point) before the device sort is queued, the
    // sequential sort there while the device sorts / scans, the permutation back (4 B per point)
    const bool msvc = pwhost::voxel_order_is_msvc();
    std::vector<pwhost::VoxelEntry> he;
    if (msvc) {
        std::vector<unsigned> hk((size_t)n);
        HIPCHK(ctx, hipMemcpyAsync(hk.data(), keys.p, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        he.resize((size_t)n);
        for (int i = 0; i < n; ++i) he[(size_t)i] = pwhost::VoxelEntry{hk[(size_t)i], i};
        tm.lap("  prep: bbox, voxel keys down");
    }
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(tsort.p, tb, keys.p, keys_s.p, vals.p, order.p, n, 0, end_bit, ctx->stream));
    hipLaunchKernelGGL(k_vg_heads, dim3(div_up(n + 1, kBlock)), dim3(kBlock), 0, ctx->stream, keys_s.p, n, head.p);
    PWCHK(pw_exclusive_scan(ctx, head.p, (long long)n + 1, &tmp));
    m = 0;
    HIPCHK(ctx, hipMemcpyAsync(&m, head.p + n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int> ho;                       // lives until the synchronisation below
    if (msvc) {
        const bool sorted = pwhost::voxel_sort_msvc(he.data(), he.size());
        tm.lap("  prep: std::sort order (host)");
        if (sorted) {
            ho.resize((size_t)n);
            for (int i = 0; i < n; ++i) ho[(size_t)i] = he[(size_t)i].pt;
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(order.p, ho.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        } else {
            fprintf(stderr, "[pwicp] voxel grid: std::sort's depth budget ran out on this input; the points of a voxel are summed in "
                            "input order instead.\n");
        }
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, start.reserve((size_t)m + 1));
    HIPCHK(ctx, vox.reserve((size_t)m));
    hipLaunchKernelGGL(k_vg_starts, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, keys_s.p, head.p, n, start.p);
    hipLaunchKernelGGL(k_vg_centroids, dim3(div_up(m, kBlock)), dim3(kBlock), 0, ctx->stream, d_in, order.p, start.p, m, n, vox.p);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));          // the sort temporaries die with this scope
    sor_in = vox.p;
    tm.lap("  prep: voxel centroids");
    }
    // ---- statistical outlier removal ---------------------------------------------------------------------------------
    Grid g;
    PWCHK(pw_grid_build(ctx, sor_in, m, 2.0f * leaf, &g));
    DevBuf<float> dist;
    HIPCHK(ctx, dist.reserve((size_t)m));
    PWCHK(pw_knn_mean_dist_launch(ctx, g.d, sor_k, dist.p));
    // global mean / sample stddev in the reference's (sequential, double) order: 4 bytes per point over PCIe
    std::vector<float> hd((size_t)m);
    HIPCHK(ctx, hipMemcpyAsync(hd.data(), dist.p, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    tm.lap("  prep: SOR grid + 14-NN mean distances down");
    double sum = 0, sq = 0;
    for (int i = 0; i < m; ++i) { sum += hd[(size_t)i]; sq += (double)(hd[(size_t)i] * hd[(size_t)i]); }
    const double mean = sum / (double)m;
    const double var = (sq - sum * sum / (double)m) / ((double)m - 1);
    const double thr = mean + sor_mult * std::sqrt(var);
    DevBuf<int> keep;
    HIPCHK(ctx, keep.reserve((size_t)m + 1));
    hipLaunchKernelGGL(k_sor_flags, dim3(div_up(m + 1, kBlock)), dim3(kBlock), 0, ctx->stream, dist.p, m, thr, keep.p);
    PWCHK(pw_exclusive_scan(ctx, keep.p, (long long)m + 1, &tmp));
    int kept = 0;
    HIPCHK(ctx, hipMemcpyAsync(&kept, keep.p + m, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, d_out->reserve((size_t)std::max(kept, 1)));
    hipLaunchKernelGGL(k_sor_emit, dim3(div_up(m, kBlock)), dim3(kBlock), 0, ctx->stream, sor_in, dist.p, m, thr, keep.p, d_out->p);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    *m_out = kept;
    tm.lap("  prep: SOR statistics (host) + compaction");
    return PWICP_OK;
}

extern "C" int pwicp_preprocess_dev(pwicp_context* ctx, const float* cloud_xyz4, int n, float voxel_size, int sor_k,
                                    double sor_mult, float* out_xyz4, int* n_out) {
    if (!ctx) return PWICP_E_INVALID;
    if (!cloud_xyz4 || !out_xyz4 || !n_out || n < 0 || !(voxel_size > 0.f) || sor_k <= 0) {
        ctx->set_err("pwicp_preprocess_dev: invalid argument");
        return PWICP_E_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf<float4> in, out;
    HIPCHK(ctx, in.reserve((size_t)std::max(n, 1)));
    if (n > 0) HIPCHK(ctx, hipMemcpyAsync(in.p, cloud_xyz4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    int m = 0;
    PWCHK(pw_preprocess_dev(ctx, in.p, n, true, voxel_size, sor_k, sor_mult, &out, &m));
    if (m > 0) HIPCHK(ctx, hipMemcpy(out_xyz4, out.p, (size_t)m * sizeof(float4), hipMemcpyDeviceToHost));
    *n_out = m;
    return PWICP_OK;
}

// SORfilter (C.cpp:441-452; decl C.h) = PCpreprocessing with isDownSamp = false: pcl::StatisticalOutlierRemoval alone.
// spacing_hint (> 0) only sizes the search grid.
extern "C" int pwicp_sor_filter_dev(pwicp_context* ctx, const float* cloud_xyz4, int n, int sor_k, double sor_mult,
                                    float spacing_hint, float* out_xyz4, int* n_out) {
    if (!ctx) return PWICP_E_INVALID;
    if (!cloud_xyz4 || !out_xyz4 || !n_out || n < 0 || sor_k <= 0) {
        ctx->set_err("pwicp_sor_filter_dev: invalid argument");
        return PWICP_E_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf<float4> in, out;
    HIPCHK(ctx, in.reserve((size_t)std::max(n, 1)));
    if (n > 0) HIPCHK(ctx, hipMemcpyAsync(in.p, cloud_xyz4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    int m = 0;
    PWCHK(pw_preprocess_dev(ctx, in.p, n, false, spacing_hint, sor_k, sor_mult, &out, &m));
    if (m > 0) HIPCHK(ctx, hipMemcpy(out_xyz4, out.p, (size_t)m * sizeof(float4), hipMemcpyDeviceToHost));
    *n_out = m;
    return PWICP_OK;
}
