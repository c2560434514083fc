// Preprocessing (setup stage, host): voxel-grid down-sampling + statistical outlier removal, mean point spacing.
//
// Reference: PCpreprocessing src/CommonFunc.cpp:423-439 (pcl::VoxelGrid, leaf = Res), SORfilter
// CommonFunc.cpp:442-452 (pcl::StatisticalOutlierRemoval, k = 14, sigma multiplier 2.7 pair / 5.0 4D),
// calPCresolution CommonFunc.cpp:239-263.  PCL 1.8.1 semantics: filters/impl/voxel_grid.hpp applyFilter
// (ijk = floor(p * inverse_leaf) - min_b, output = float centroid per occupied voxel in ascending linear index
// i + j*dx + k*dx*dy), filters/impl/statistical_outlier_removal.hpp applyFilterIndices (mean distance to the k
// nearest OTHER points, global mean + sample stddev, keep d <= mean + mult*stddev).
// SURVEY.md §8 row f2 ("next"): host today.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>

#include "kdtree.h"
#include "msvc_sort.h"
#include "parallel.h"
#include "preprocess.h"
#include "pwicp.h"

namespace pwhost {

// PWICP_VOXEL_ORDER = msvc (default) | input: the order the points of one voxel are summed in (msvc_sort.h)
bool voxel_order_is_msvc() {
    const char* e = std::getenv("PWICP_VOXEL_ORDER");
    return !(e && std::strcmp(e, "input") == 0);
}

// (voxel index, point) entries -> the order pcl::VoxelGrid's std::sort leaves them in.  false: the sort's depth budget ran
// out (adversarial input); `e` then holds a permutation in an unspecified state and the caller sorts it stably instead.
bool voxel_sort_msvc(VoxelEntry* e, size_t n) {
    // $PWICP_MSVC_SORT_BUDGET (tests only): initial depth budget of the introsort instead of n, so that the heap-sort fall-back runs
    std::ptrdiff_t budget = -1;
    if (const char* b = std::getenv("PWICP_MSVC_SORT_BUDGET")) budget = (std::ptrdiff_t)std::atoll(b);
    return msvc_order::sort(e, e + n, [](const VoxelEntry& a, const VoxelEntry& b) { return a.idx < b.idx; }, host_threads(), budget);
}

// returns the number of output points; out must hold n points
int voxel_grid(const float* in4, int n, float leaf, float* out4) {
    if (n <= 0) return 0;
    const float inv = 1.0f / leaf;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            const float v = in4[4 * (size_t)i + d];
            mn[d] = std::min(mn[d], v);
            mx[d] = std::max(mx[d], v);
        }
    {   // pcl::VoxelGrid (PCL 1.8.1 voxel_grid.hpp): index overflow -> warning, output = input
        const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1,
                        dz = (long long)((mx[2] - mn[2]) * inv) + 1;
        if ((double)dx * (double)dy * (double)dz > 2147483647.0) {
            std::fprintf(stderr, "[pwicp] Leaf size is too small for the input dataset. Integer indices would overflow: the cloud is "
                                 "passed on unfiltered (pcl::VoxelGrid semantics).\n");
            std::memcpy(out4, in4, (size_t)n * 16);
            return n;
        }
    }
    int minb[3], divb[3];
    for (int d = 0; d < 3; ++d) {
        minb[d] = (int)std::floor(mn[d] * inv);
        divb[d] = (int)std::floor(mx[d] * inv) - minb[d] + 1;
    }
    const int mul1 = divb[0], mul2 = divb[0] * divb[1];
    using Entry = VoxelEntry;
    std::vector<Entry> e((size_t)n);
    auto fill = [&] {
        for (int i = 0; i < n; ++i) {
            const float* p = in4 + 4 * (size_t)i;
            const int i0 = (int)(std::floor(p[0] * inv) - (float)minb[0]);
            const int i1 = (int)(std::floor(p[1] * inv) - (float)minb[1]);
            const int i2 = (int)(std::floor(p[2] * inv) - (float)minb[2]);
            e[(size_t)i] = Entry{(unsigned)(i0 + i1 * mul1 + i2 * mul2), i};
        }
    };
    fill();
    // PCL sorts with std::sort (unstable): the points of a voxel in the order the reference's build leaves them in, or
    // (PWICP_VOXEL_ORDER=input) in input order
    bool sorted = false;
    if (voxel_order_is_msvc()) {
        sorted = voxel_sort_msvc(e.data(), e.size());
        if (!sorted) {
            std::fprintf(stderr, "[pwicp] voxel grid: std::sort's depth budget ran out on this input; the points of a voxel are "
                                 "summed in input order instead.\n");
            fill();
        }
    }
    if (!sorted) std::stable_sort(e.begin(), e.end(), [](const Entry& a, const Entry& b) { return a.idx < b.idx; });
    int m = 0;
    for (size_t i = 0; i < e.size();) {
        size_t j = i;
        float c0 = 0, c1 = 0, c2 = 0;
        while (j < e.size() && e[j].idx == e[i].idx) {
            const float* p = in4 + 4 * (size_t)e[j].pt;
            c0 += p[0]; c1 += p[1]; c2 += p[2];
            ++j;
        }
        const float cnt = (float)(j - i);
        float* o = out4 + 4 * (size_t)m;
        o[0] = c0 / cnt; o[1] = c1 / cnt; o[2] = c2 / cnt; o[3] = 1.0f;
        ++m;
        i = j;
    }
    return m;
}

int sor_filter(const float* in4, int n, int mean_k, double std_mul, float* out4) {
    if (n <= 0) return 0;
    KdTree<float> tree;
    tree.build(in4, n, 4);
    std::vector<float> dist((size_t)n);
    std::vector<KdTree<float>::Hit> hits((size_t)mean_k + 1);
    for (int i = 0; i < n; ++i) {
        const int c = tree.knn(in4 + 4 * (size_t)i, mean_k + 1, hits.data());
        double s = 0.0;
        for (int k = 1; k < c; ++k) s += (double)std::sqrt(hits[(size_t)k].d2);   // k = 0 is the query point itself
        dist[(size_t)i] = (float)(s / mean_k);
    }
    double sum = 0, sq = 0;
    for (int i = 0; i < n; ++i) { sum += dist[(size_t)i]; sq += (double)(dist[(size_t)i] * dist[(size_t)i]); }
    const double mean = sum / (double)n;
    const double var = (sq - sum * sum / (double)n) / ((double)n - 1);
    const double thr = mean + std_mul * std::sqrt(var);
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (!((double)dist[(size_t)i] > thr)) { std::memcpy(out4 + 4 * (size_t)m, in4 + 4 * (size_t)i, 16); ++m; }
    return m;
}

float pc_resolution(const float* c4, int n) {
    KdTree<float> tree;
    tree.build(c4, n, 4);
    float res = 0.0f;
    int cnt = 0;
    KdTree<float>::Hit h[2];
    for (int i = 0; i < n; ++i) {
        if (tree.knn(c4 + 4 * (size_t)i, 2, h) != 2) return 0.0f;
        res += std::sqrt(h[1].d2);
        ++cnt;
    }
    if (cnt) res /= (float)cnt;
    return res;
}

}  // namespace pwhost

extern "C" {

PWICP_API int pwicp_preprocess(const float* cloud_xyz4, int n, float voxel_size, int sor_k, double sor_mult, float* out_xyz4,
                     int* n_out) {
    if (!cloud_xyz4 || !out_xyz4 || !n_out || n < 0 || !(voxel_size > 0.f) || sor_k <= 0) return PWICP_E_INVALID;
    std::vector<float> tmp((size_t)std::max(n, 1) * 4);
    const int m = pwhost::voxel_grid(cloud_xyz4, n, voxel_size, tmp.data());
    *n_out = pwhost::sor_filter(tmp.data(), m, sor_k, sor_mult, out_xyz4);
    return PWICP_OK;
}

// SORfilter (C.cpp:441-452) on the host
PWICP_API int pwicp_sor_filter(const float* cloud_xyz4, int n, int sor_k, double sor_mult, float* out_xyz4, int* n_out) {
    if (!cloud_xyz4 || !out_xyz4 || !n_out || n < 0 || sor_k <= 0) return PWICP_E_INVALID;
    *n_out = pwhost::sor_filter(cloud_xyz4, n, sor_k, sor_mult, out_xyz4);
    return PWICP_OK;
}

PWICP_API float pwicp_pc_resolution(const float* cloud_xyz4, int n) {
    if (!cloud_xyz4 || n < 2) return 0.0f;
    return pwhost::pc_resolution(cloud_xyz4, n);
}

}  // extern "C"
