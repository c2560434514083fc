// Segmentation front end (setup stage, host): per-point k-NN (k = 45, the query point included), PCA plane normals
// and the boundary-preserving supervoxel segmentation of Lin et al. (ISPRS J. 2018), producing the supervoxel
// label of every point — the input of pwicp_pair_create / pwicp_select_patches.
//
// Reference: PatchGenerationAndRefinement src/Segmentation.cpp:18-68, which drives the vendored library
// codelibrary/util/tree/kd_tree.h (k-NN, ascending squared distance), geometry/point_cloud/
// pca_estimate_normals.h:42-108 (closed-form smallest eigenvector in double), geometry/point_cloud/
// supervoxel_segmentation.h:65-265 (fusion by increasing lambda, boundary refinement, relabelling), with the
// metric of include/Segmentation.h:362-375 and the supervoxel count = number of occupied grid cells of edge
// `resolution` (geometry/point_cloud/grid_sample.h:30-75).
//
// This is SURVEY.md §8 row f1 ("next"): it runs once per cloud on the host today; the registration loop itself
// never touches it.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <unordered_set>
#include <vector>

#include "kdtree.h"
#include "pwicp.h"

namespace {

struct V3 {
    double x, y, z;
};

// pca_estimate_normals.h:42-108 with unit weights, points in the given order
V3 pca_normal(const double* pts, const int* nb, int k) {
    double cx = 0, cy = 0, cz = 0, sum = 0;
    for (int i = 0; i < k; ++i) {
        const double* p = pts + 3 * (size_t)nb[i];
        const double w = 1.0;
        cx += w * p[0]; cy += w * p[1]; cz += w * p[2];
        sum += w;
    }
    const double inv = 1.0 / sum;
    cx *= inv; cy *= inv; cz *= inv;
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0, s = 0;
    for (int i = 0; i < k; ++i) {
        const double* p = pts + 3 * (size_t)nb[i];
        const double x = p[0] - cx, y = p[1] - cy, z = p[2] - cz, w = 1.0;
        a00 += w * x * x; a01 += w * x * y; a02 += w * x * z;
        a11 += w * y * y; a12 += w * y * z; a22 += w * z * z;
        s += w;
    }
    const double t = 1.0 / s;
    a00 = a00 * t; a01 = a01 * t; a02 = a02 * t; a11 = a11 * t; a12 = a12 * t; a22 = a22 * t;
    // least eigenvalue of the covariance matrix (trigonometric form)
    const double q = (a00 + a11 + a22) / 3.0;
    double pq = (a00 - q) * (a00 - q) + (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) +
                2.0 * (a01 * a01 + a02 * a02 + a12 * a12);
    pq = std::sqrt(pq / 6.0);
    const double mpq = std::pow(1.0 / pq, 3.0);
    const double det_b = mpq * ((a00 - q) * ((a11 - q) * (a22 - q) - a12 * a12) - a01 * (a01 * (a22 - q) - a12 * a02) +
                                a02 * (a01 * a12 - (a11 - q) * a02));
    const double r = 0.5 * det_b;
    double phi;
    if (r <= -1.0) phi = M_PI / 3.0;
    else if (r >= 1.0) phi = 0.0;
    else phi = std::acos(r) / 3.0;
    const double eig = q + 2.0 * pq * std::cos(phi + M_PI * (2.0 / 3.0));
    V3 n;
    n.x = a01 * a12 - a02 * (a11 - eig);
    n.y = a01 * a02 - a12 * (a00 - eig);
    n.z = (a00 - eig) * (a11 - eig) - a01 * a01;
    const double norm = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
    if (norm == 0.0) return V3{0.0, 0.0, 1.0};
    const double f = 1.0 / norm;
    n.x *= f; n.y *= f; n.z *= f;
    return n;
}

struct Metric {      // Segmentation.h:362-375
    const double* pts;
    const V3* nrm;
    double resolution;
    double operator()(int a, int b) const {
        const V3 &n1 = nrm[a], &n2 = nrm[b];
        const double dot = n1.x * n2.x + n1.y * n2.y + n1.z * n2.z;
        const double t1 = pts[3 * (size_t)a] - pts[3 * (size_t)b], t2 = pts[3 * (size_t)a + 1] - pts[3 * (size_t)b + 1],
                     t3 = pts[3 * (size_t)a + 2] - pts[3 * (size_t)b + 2];
        const double dist = std::sqrt(t1 * t1 + t2 * t2 + t3 * t3);
        return 1.0 - std::fabs(dot) + dist / resolution * 0.4;
    }
};

struct DisjointSet {     // codelibrary/util/set/disjoint_set.h (path halving, Link(i -> j))
    mutable std::vector<int> parent;
    explicit DisjointSet(int n) : parent((size_t)n) { for (int i = 0; i < n; ++i) parent[(size_t)i] = i; }
    int find(int i) const {
        while (i != parent[(size_t)i]) {
            parent[(size_t)i] = parent[(size_t)parent[(size_t)i]];
            i = parent[(size_t)i];
        }
        return i;
    }
    void link(int i, int j) { parent[(size_t)i] = j; }
};

// grid_sample.h:30-75: only the NUMBER of occupied cells is consumed by the segmentation
int count_occupied_cells(const double* pts, int n, double resolution) {
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            mn[d] = std::min(mn[d], pts[3 * (size_t)i + d]);
            mx[d] = std::max(mx[d], pts[3 * (size_t)i + d]);
        }
    const int size1 = (int)((mx[0] - mn[0]) / resolution + 1), size2 = (int)((mx[1] - mn[1]) / resolution + 1),
              size3 = (int)((mx[2] - mn[2]) / resolution + 1);
    std::unordered_set<uint64_t> cells;
    cells.reserve((size_t)n / 8 + 16);
    for (int i = 0; i < n; ++i) {
        int x = (int)((pts[3 * (size_t)i] - mn[0]) / resolution);
        int y = (int)((pts[3 * (size_t)i + 1] - mn[1]) / resolution);
        int z = (int)((pts[3 * (size_t)i + 2] - mn[2]) / resolution);
        x = std::min(std::max(x, 0), size1 - 1);
        y = std::min(std::max(y, 0), size2 - 1);
        z = std::min(std::max(z, 0), size3 - 1);
        cells.insert(((uint64_t)(uint32_t)x << 42) ^ ((uint64_t)(uint32_t)y << 21) ^ (uint64_t)(uint32_t)z);
    }
    return (int)cells.size();
}

// supervoxel_segmentation.h:65-248
int supervoxel_segmentation(const Metric& metric, const std::vector<std::vector<int>>& neighbors, int n_points,
                            int n_supervoxels, std::vector<int>* labels_out) {
    DisjointSet set(n_points);
    std::vector<int> supervoxels((size_t)n_points);
    for (int i = 0; i < n_points; ++i) supervoxels[(size_t)i] = i;
    std::vector<int> sizes((size_t)n_points, 1), queue((size_t)n_points);
    std::vector<std::vector<int>> adjacents = neighbors;
    int number_of_supervoxels = n_points;
    std::vector<char> visited((size_t)n_points, 0);

    // minimum value of lambda
    std::vector<double> dis((size_t)n_points, DBL_MAX);
    for (int i = 0; i < n_points; ++i)
        for (int j : adjacents[(size_t)i])
            if (i != j) dis[(size_t)i] = std::min(dis[(size_t)i], metric(i, j));
    double lambda;
    {
        std::vector<double> v = dis;
        std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
        lambda = std::max(DBL_EPSILON, v[v.size() / 2]);
    }

    // ---- step 1: fusion with doubling lambda ---------------------------------------------------------------
    for (;; lambda *= 2.0) {
        if (supervoxels.size() <= 1) break;
        for (int i : supervoxels) {
            if (adjacents[(size_t)i].empty()) continue;
            visited[(size_t)i] = 1;
            int front = 0, back = 1;
            queue[(size_t)front++] = i;
            for (int j : adjacents[(size_t)i]) {
                j = set.find(j);
                if (!visited[(size_t)j]) { visited[(size_t)j] = 1; queue[(size_t)back++] = j; }
            }
            std::vector<int> adjacent;
            while (front < back) {
                const int j = queue[(size_t)front++];
                const double loss = sizes[(size_t)j] * metric(i, j);
                const double improvement = lambda - loss;
                if (improvement > 0.0) {
                    set.link(j, i);
                    sizes[(size_t)i] += sizes[(size_t)j];
                    for (int k : adjacents[(size_t)j]) {
                        k = set.find(k);
                        if (!visited[(size_t)k]) { visited[(size_t)k] = 1; queue[(size_t)back++] = k; }
                    }
                    adjacents[(size_t)j].clear();
                    if (--number_of_supervoxels == n_supervoxels) break;
                } else {
                    adjacent.push_back(j);
                }
            }
            adjacents[(size_t)i].swap(adjacent);
            for (int j = 0; j < back; ++j) visited[(size_t)queue[(size_t)j]] = 0;
            if (number_of_supervoxels == n_supervoxels) break;
        }
        number_of_supervoxels = 0;
        for (int i : supervoxels)
            if (set.find(i) == i) supervoxels[(size_t)number_of_supervoxels++] = i;
        supervoxels.resize((size_t)number_of_supervoxels);
        if (number_of_supervoxels == n_supervoxels) break;
    }
    std::vector<int>& labels = *labels_out;
    labels.resize((size_t)n_points);
    for (int i = 0; i < n_points; ++i) labels[(size_t)i] = set.find(i);

    // ---- step 2: boundary refinement -------------------------------------------------------------------------
    for (int i = 0; i < n_points; ++i) dis[(size_t)i] = metric(i, labels[(size_t)i]);
    std::queue<int> q;
    std::vector<char> in_q((size_t)n_points, 0);
    for (int i = 0; i < n_points; ++i)
        for (int j : neighbors[(size_t)i])
            if (labels[(size_t)i] != labels[(size_t)j]) {
                if (!in_q[(size_t)i]) { q.push(i); in_q[(size_t)i] = 1; }
                if (!in_q[(size_t)j]) { q.push(j); in_q[(size_t)j] = 1; }
            }
    while (!q.empty()) {
        const int i = q.front();
        q.pop();
        in_q[(size_t)i] = 0;
        bool change = false;
        for (int j : neighbors[(size_t)i]) {
            const int a = labels[(size_t)i], b = labels[(size_t)j];
            if (a == b) continue;
            const double d = metric(i, b);
            if (d < dis[(size_t)i]) { labels[(size_t)i] = b; dis[(size_t)i] = d; change = true; }
        }
        if (change)
            for (int j : neighbors[(size_t)i])
                if (labels[(size_t)i] != labels[(size_t)j] && !in_q[(size_t)j]) { q.push(j); in_q[(size_t)j] = 1; }
    }

    // ---- step 3: relabel -----------------------------------------------------------------------------------------
    std::vector<int> map((size_t)n_points, 0);
    for (size_t i = 0; i < supervoxels.size(); ++i) map[(size_t)supervoxels[i]] = (int)i;
    for (int i = 0; i < n_points; ++i) labels[(size_t)i] = map[(size_t)labels[(size_t)i]];
    return (int)supervoxels.size();
}

// normals (S.cpp:39-44) + segmentation (S.cpp:51-67) from given neighbour lists
int segment_from_neighbors(const std::vector<double>& pts, int n, const std::vector<std::vector<int>>& neighbors,
                           float sv_resolution, int32_t* labels, int* n_supervoxels) {
    std::vector<V3> normals((size_t)n);
    for (int i = 0; i < n; ++i) normals[(size_t)i] = pca_normal(pts.data(), neighbors[(size_t)i].data(), (int)neighbors[(size_t)i].size());
    const double res = (double)sv_resolution;
    Metric metric{pts.data(), normals.data(), res};
    const int n_sv = count_occupied_cells(pts.data(), n, res);
    std::vector<int> lab;
    const int got = supervoxel_segmentation(metric, neighbors, n, n_sv, &lab);
    for (int i = 0; i < n; ++i) labels[i] = lab[(size_t)i];
    *n_supervoxels = got;
    return PWICP_OK;
}

}  // namespace

extern "C" {

// kNN = 45 in the reference (include/CommonFunc.h:41).  Host-only variant: k-NN with the host KD-tree.
PWICP_API int pwicp_frontend_segment(const float* cloud_xyz4, int n, float sv_resolution, int knn, int32_t* labels,
                                     int* n_supervoxels) {
    if (!cloud_xyz4 || !labels || !n_supervoxels || n <= 0 || knn <= 0 || knn >= n || !(sv_resolution > 0.f))
        return PWICP_E_INVALID;
    std::vector<double> pts((size_t)n * 3);                                  // S.cpp:18-22: float -> double
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) pts[3 * (size_t)i + d] = (double)cloud_xyz4[4 * (size_t)i + d];
    pwhost::KdTree<double> tree;
    tree.build(pts.data(), n, 3);
    std::vector<std::vector<int>> neighbors((size_t)n);
    std::vector<pwhost::KdTree<double>::Hit> hits((size_t)knn);
    for (int i = 0; i < n; ++i) {                                             // S.cpp:37-41
        const int c = tree.knn(pts.data() + 3 * (size_t)i, knn, hits.data());
        std::vector<int>& nb = neighbors[(size_t)i];
        nb.resize((size_t)c);
        for (int k = 0; k < c; ++k) nb[(size_t)k] = hits[(size_t)k].idx;
    }
    return segment_from_neighbors(pts, n, neighbors, sv_resolution, labels, n_supervoxels);
}

// Same result, with the k-NN graph built on the GPU (pwicp_knn): the variant the entry points use.
PWICP_API int pwicp_frontend_segment_dev(pwicp_context* ctx, const float* cloud_xyz4, int n, float sv_resolution, int knn,
                                         float point_spacing, int32_t* labels, int* n_supervoxels) {
    if (!ctx || !cloud_xyz4 || !labels || !n_supervoxels || n <= 0 || knn <= 0 || knn >= n || !(sv_resolution > 0.f))
        return PWICP_E_INVALID;
    std::vector<int32_t> nb((size_t)n * knn);
    const int rc = pwicp_knn(ctx, cloud_xyz4, n, knn, point_spacing > 0.f ? 2.0f * point_spacing : 0.f, nb.data());
    if (rc != PWICP_OK) return rc;
    std::vector<double> pts((size_t)n * 3);
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) pts[3 * (size_t)i + d] = (double)cloud_xyz4[4 * (size_t)i + d];
    std::vector<std::vector<int>> neighbors((size_t)n);
    for (int i = 0; i < n; ++i) neighbors[(size_t)i].assign(nb.begin() + (size_t)i * knn, nb.begin() + (size_t)(i + 1) * knn);
    return segment_from_neighbors(pts, n, neighbors, sv_resolution, labels, n_supervoxels);
}

}  // extern "C"
