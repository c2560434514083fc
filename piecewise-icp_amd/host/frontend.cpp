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
#include <algorithm>
#include <cfloat>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <unordered_set>
#include <vector>

#include "frontend.h"
#include "hostbuf.h"
#include "kdtree.h"
#include "parallel.h"
#include "pwicp.h"

namespace {

// PWICP_TRACE=1: stage timings of the front end on stderr
struct StageTimer {
    const bool on = std::getenv("PWICP_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pwicp front end] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

using Pt = pwhost::FePt;      // one cache line per point: position and PCA normal (double, as in the reference's front end)

// Plane normal of a point from its k nearest neighbours: the direction of least variance of the neighbourhood.
// Specification (must be met to the last bit, the labels depend on it): the reference's front end
// (codelibrary/geometry/point_cloud/pca_estimate_normals.h:42-108) accumulates, in double and in neighbour order, the mean
// and then the six second moments about the mean, each divided by the number of neighbours; takes the smallest root of the
// characteristic polynomial in closed form (trace shift q, scale p = sqrt(|B|^2 / 6), angle acos(det(B/p) / 2) / 3); and
// returns the normalised cross product of the first two rows of (C - lambda I).  Written here as three steps over small
// value types; the arithmetic of every step follows that specification operation by operation.
struct Sym3 {                     // symmetric 3x3: xx xy xz yy yz zz
    double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
};

inline Sym3 neighbourhood_scatter(const Pt* P, const int32_t* nb, int k) {
    double mean[3] = {0, 0, 0}, count = 0;
    for (int e = 0; e < k; ++e) {
        const Pt& v = P[nb[e]];
        mean[0] += 1.0 * v.x; mean[1] += 1.0 * v.y; mean[2] += 1.0 * v.z;       // (unit weights, as the reference passes them)
        count += 1.0;
    }
    const double to_mean = 1.0 / count;
    for (double& m : mean) m *= to_mean;
    Sym3 S;
    double weight = 0;
    for (int e = 0; e < k; ++e) {
        const Pt& v = P[nb[e]];
        const double d0 = v.x - mean[0], d1 = v.y - mean[1], d2 = v.z - mean[2];
        S.xx += 1.0 * d0 * d0; S.xy += 1.0 * d0 * d1; S.xz += 1.0 * d0 * d2;
        S.yy += 1.0 * d1 * d1; S.yz += 1.0 * d1 * d2; S.zz += 1.0 * d2 * d2;
        weight += 1.0;
    }
    const double scale = 1.0 / weight;
    S.xx = S.xx * scale; S.xy = S.xy * scale; S.xz = S.xz * scale; S.yy = S.yy * scale; S.yz = S.yz * scale; S.zz = S.zz * scale;
    return S;
}

inline double smallest_eigenvalue(const Sym3& C) {
    const double shift = (C.xx + C.yy + C.zz) / 3.0;
    const double bx = C.xx - shift, by = C.yy - shift, bz = C.zz - shift;          // diagonal of B = C - shift * I
    const double p = std::sqrt((bx * bx + by * by + bz * bz + 2.0 * (C.xy * C.xy + C.xz * C.xz + C.yz * C.yz)) / 6.0);
    const double inv_p3 = std::pow(1.0 / p, 3.0);
    const double det = inv_p3 * (bx * (by * bz - C.yz * C.yz) - C.xy * (C.xy * bz - C.yz * C.xz) + C.xz * (C.xy * C.yz - by * C.xz));
    const double half = 0.5 * det;
    const double angle = half <= -1.0 ? M_PI / 3.0 : (half >= 1.0 ? 0.0 : std::acos(half) / 3.0);
    return shift + 2.0 * p * std::cos(angle + M_PI * (2.0 / 3.0));
}

// null direction of (C - lambda_min I): cross product of its first two rows, normalised ((0, 0, 1) when it vanishes)
inline void normal_from_scatter(const Sym3& C, double* n3) {
    const double lam = smallest_eigenvalue(C);
    const double n0 = C.xy * C.yz - C.xz * (C.yy - lam);
    const double n1 = C.xy * C.xz - C.yz * (C.xx - lam);
    const double n2 = (C.xx - lam) * (C.yy - lam) - C.xy * C.xy;
    const double len = std::sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    if (len == 0.0) { n3[0] = 0.0; n3[1] = 0.0; n3[2] = 1.0; return; }
    const double unit = 1.0 / len;
    n3[0] = n0 * unit; n3[1] = n1 * unit; n3[2] = n2 * unit;
}

void pca_normal(Pt* P, int self, const int32_t* nb, int k) {
    double n3[3];
    normal_from_scatter(neighbourhood_scatter(P, nb, k), n3);
    P[self].nx = n3[0]; P[self].ny = n3[1]; P[self].nz = n3[2];
}

struct Metric {      // Segmentation.h:362-375
    const Pt* P;
    double resolution;
    double operator()(int a, int b) const {
        const Pt &p = P[a], &q = P[b];
        const double dot = p.nx * q.nx + p.ny * q.ny + p.nz * q.nz;
        const double t1 = p.x - q.x, t2 = p.y - q.y, t3 = p.z - q.z;
        const double dist = std::sqrt(t1 * t1 + t2 * t2 + t3 * t3);
        return 1.0 - std::fabs(dot) + dist / resolution * 0.4;
    }
};

// bit e set <=> labels[row[e]] != mine, for k <= 64 neighbours: the hot loop of the boundary refinement
uint64_t differing_labels_scalar(const int* labels, const int32_t* row, int k, int mine) {
    uint64_t m = 0;
    for (int e = 0; e < k; ++e) m |= (uint64_t)(labels[(size_t)row[e]] != mine) << e;
    return m;
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) uint64_t differing_labels_avx2(const int* labels, const int32_t* row, int k, int mine) {
    const __m256i me = _mm256_set1_epi32(mine);
    uint64_t m = 0;
    int e = 0;
    for (; e + 8 <= k; e += 8) {
        const __m256i idx = _mm256_loadu_si256((const __m256i*)(row + e));
        const __m256i lab = _mm256_i32gather_epi32(labels, idx, 4);
        const unsigned eq = (unsigned)_mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(lab, me)));
        m |= (uint64_t)(~eq & 0xffu) << e;
    }
    for (; e < k; ++e) m |= (uint64_t)(labels[(size_t)row[e]] != mine) << e;
    return m;
}
#endif
typedef uint64_t (*DifferingFn)(const int*, const int32_t*, int, int);
DifferingFn pick_differing() {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) return differing_labels_avx2;
#endif
    return differing_labels_scalar;
}

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
int count_occupied_cells(const Pt* P, int n, double resolution) {
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int i = 0; i < n; ++i) {
        const double c[3] = {P[i].x, P[i].y, P[i].z};
        for (int d = 0; d < 3; ++d) { mn[d] = std::min(mn[d], c[d]); mx[d] = std::max(mx[d], c[d]); }
    }
    const int size1 = (int)((mx[0] - mn[0]) / resolution + 1), size2 = (int)((mx[1] - mn[1]) / resolution + 1),
              size3 = (int)((mx[2] - mn[2]) / resolution + 1);
    std::vector<uint64_t> cells((size_t)n);
    pwhost::parallel_for(n, [&](long long lo, long long hi) {
        for (long long i = lo; i < hi; ++i) {
            int x = (int)((P[i].x - mn[0]) / resolution);
            int y = (int)((P[i].y - mn[1]) / resolution);
            int z = (int)((P[i].z - mn[2]) / resolution);
            x = std::min(std::max(x, 0), size1 - 1);
            y = std::min(std::max(y, 0), size2 - 1);
            z = std::min(std::max(z, 0), size3 - 1);
            cells[(size_t)i] = ((uint64_t)(uint32_t)x << 42) ^ ((uint64_t)(uint32_t)y << 21) ^ (uint64_t)(uint32_t)z;
        }
    });
    std::sort(cells.begin(), cells.end());
    return (int)(std::unique(cells.begin(), cells.end()) - cells.begin());
}

// supervoxel_segmentation.h:172-236: boundary refinement of the labels (root point of every point) in place
void boundary_refinement(const Metric& metric, const int32_t* nb, int k, int n_points, std::vector<int>& labels) {
    StageTimer tm;
    std::vector<double> dis((size_t)n_points);
    pwhost::parallel_for(n_points, [&](long long lo, long long hi) {
        for (long long i = lo; i < hi; ++i) dis[(size_t)i] = metric((int)i, labels[(size_t)i]);
    });
    std::vector<int> q((size_t)n_points);              // ring buffer: every point is queued at most once at a time
    size_t qh = 0, qt = 0, qn = 0;
    const size_t qcap = (size_t)n_points;
    auto push = [&](int v) { q[qt] = v; qt = (qt + 1 == qcap) ? 0 : qt + 1; ++qn; };
    std::vector<char> in_q((size_t)n_points, 0);
    // seed: every point with a differently labelled neighbour, and that neighbour, in scan order.  The label
    // comparisons (n*k random reads) run on all threads into one bit mask per point; the order-dependent pushes
    // then only visit the boundary points.
    if (k <= 64) {
        std::vector<uint64_t> diff((size_t)n_points);
        pwhost::parallel_for(n_points, [&](long long lo, long long hi) {
            for (long long i = lo; i < hi; ++i) {
                const int32_t* row = nb + (size_t)i * (size_t)k;
                const int li = labels[(size_t)i];
                uint64_t m = 0;
                for (int e = 0; e < k; ++e) m |= (uint64_t)(labels[(size_t)row[e]] != li) << e;
                diff[(size_t)i] = m;
            }
        });
        for (int i = 0; i < n_points; ++i) {
            uint64_t m = diff[(size_t)i];
            if (!m) continue;
            const int32_t* row = nb + (size_t)i * (size_t)k;
            if (!in_q[(size_t)i]) { push(i); in_q[(size_t)i] = 1; }
            while (m) {
                const int j = row[__builtin_ctzll(m)];
                m &= m - 1;
                if (!in_q[(size_t)j]) { push(j); in_q[(size_t)j] = 1; }
            }
        }
    } else {
        for (int i = 0; i < n_points; ++i) {
            const int32_t* row = nb + (size_t)i * (size_t)k;
            for (int e = 0; e < k; ++e) {
                const int j = row[e];
                if (labels[(size_t)i] != labels[(size_t)j]) {
                    if (!in_q[(size_t)i]) { push(i); in_q[(size_t)i] = 1; }
                    if (!in_q[(size_t)j]) { push(j); in_q[(size_t)j] = 1; }
                }
            }
        }
    }
    if (tm.on) { char b[64]; std::snprintf(b, sizeof b, "    seeds: %zu", qn); tm.lap(b); }
    size_t pops = 0;
    const DifferingFn differing = pick_differing();
    while (qn) {
        const int i = q[qh];
        qh = (qh + 1 == qcap) ? 0 : qh + 1;
        --qn;
        ++pops;
        in_q[(size_t)i] = 0;
        bool change = false;
        const int32_t* row = nb + (size_t)i * (size_t)k;
        // Only neighbours whose label differs from the point's label at the START of the visit can matter (a label the
        // point switches to during the visit is skipped from then on, its old label can never win again), and a label
        // already tried cannot win later either (dis[i] only decreases): each distinct neighbouring label is evaluated
        // once, in the order of its first occurrence; the outcome is the reference's.
        if (k <= 64) {
            uint64_t m = differing(labels.data(), row, k, labels[(size_t)i]);
            int tried[8], n_tried = 0;
            while (m) {
                const int e = __builtin_ctzll(m);
                m &= m - 1;
                const int a = labels[(size_t)i], b = labels[(size_t)row[e]];
                if (a == b) continue;
                bool seen = false;
                for (int t = 0; t < n_tried; ++t) seen |= (tried[t] == b);
                if (seen) continue;
                if (n_tried < 8) tried[n_tried++] = b;
                const double d = metric(i, b);
                if (d < dis[(size_t)i]) { labels[(size_t)i] = b; dis[(size_t)i] = d; change = true; }
            }
        } else {
            for (int e = 0; e < k; ++e) {
                const int a = labels[(size_t)i], b = labels[(size_t)row[e]];
                if (a == b) continue;
                const double d = metric(i, b);
                if (d < dis[(size_t)i]) { labels[(size_t)i] = b; dis[(size_t)i] = d; change = true; }
            }
        }
        if (change) {
            if (k <= 64) {
                uint64_t m = differing(labels.data(), row, k, labels[(size_t)i]);
                while (m) {
                    const int j = row[__builtin_ctzll(m)];
                    m &= m - 1;
                    if (!in_q[(size_t)j]) { push(j); in_q[(size_t)j] = 1; }
                }
            } else {
                for (int e = 0; e < k; ++e) {
                    const int j = row[e];
                    if (labels[(size_t)i] != labels[(size_t)j] && !in_q[(size_t)j]) { push(j); in_q[(size_t)j] = 1; }
                }
            }
        }
    }
    if (tm.on) { char b[64]; std::snprintf(b, sizeof b, "  boundary refinement (%zu pops)", pops); tm.lap(b); }

}

// supervoxel_segmentation.h:65-248.  nb: the k-NN graph, n_points rows of k indices.
// Adjacency lists: a node that was never a fusion centre still reads its row of the k-NN graph; every list a round
// writes is appended to one arena (list of node i = arena[off[i] .. off[i]+len[i])) which is compacted in place
// between rounds.  The serial, order-dependent fusion pass therefore allocates nothing per supervoxel and touches
// little fresh memory (first-touch page faults dominate this stage on virtualised hosts).
// roots_out != nullptr: stop after the fusion; labels = root point of every point, *roots_out = the roots in ascending order
int supervoxel_segmentation(const Metric& metric, const int32_t* nb, int k, int n_points, int n_supervoxels,
                            std::vector<int>* labels_out, std::vector<int>* roots_out = nullptr) {
    StageTimer tm;
    DisjointSet set(n_points);
    std::vector<int> supervoxels((size_t)n_points);
    for (int i = 0; i < n_points; ++i) supervoxels[(size_t)i] = i;
    std::vector<int> sizes((size_t)n_points, 1), queue((size_t)n_points);
    constexpr size_t kInGraph = ~(size_t)0;                     // off[i]: the list is still row i of nb
    std::vector<size_t> off((size_t)n_points, kInGraph);
    std::vector<int> len((size_t)n_points, k);
    pwhost::HostBuf<int> arena;                 // huge-page backed; grows by doubling (rare: one round appends < n*k entries)
    size_t arena_size = 0;
    if (!arena.reserve((size_t)n_points * (size_t)k)) return -1;
    auto list_of = [&](int i) -> const int* {
        return off[(size_t)i] == kInGraph ? nb + (size_t)i * (size_t)k : arena.data() + off[(size_t)i];
    };
    int number_of_supervoxels = n_points;
    std::vector<char> visited((size_t)n_points, 0);

    // minimum value of lambda
    std::vector<double> dis((size_t)n_points, DBL_MAX);
    pwhost::parallel_for(n_points, [&](long long lo, long long hi) {
        for (long long i = lo; i < hi; ++i) {
            double d = DBL_MAX;
            const int32_t* row = nb + (size_t)i * (size_t)k;
            for (int e = 0; e < k; ++e)
                if (row[e] != (int)i) d = std::min(d, metric((int)i, row[e]));
            dis[(size_t)i] = d;
        }
    });
    double lambda;
    {
        std::vector<double> v = dis;
        std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
        lambda = std::max(DBL_EPSILON, v[v.size() / 2]);
    }
    tm.lap("  adjacency + lambda0");

    // ---- step 1: fusion with doubling lambda ---------------------------------------------------------------
    std::vector<int> adjacent;
    std::vector<double> loss_of((size_t)n_points);
    for (;; lambda *= 2.0) {
        if (supervoxels.size() <= 1) break;
        for (int i : supervoxels) {
            if (len[(size_t)i] == 0) continue;
            visited[(size_t)i] = 1;
            int front = 0, back = 1;
            queue[(size_t)front++] = i;
            {
                const int* a = list_of(i);
                for (int e = 0, m = len[(size_t)i]; e < m; ++e) {
                    const int j = set.find(a[e]);
                    if (!visited[(size_t)j]) { visited[(size_t)j] = 1; queue[(size_t)back++] = j; }
                }
            }
            adjacent.clear();
            bool reached = false;
            while (front < back && !reached) {
                // losses of everything queued so far in one branch-free sweep (a candidate's size cannot change while
                // it waits in the queue: only the centre i absorbs), then the FIFO decisions in the reference's order
                const int stop = back;
                for (int e = front; e < stop; ++e) {
                    const int j = queue[(size_t)e];
                    loss_of[(size_t)e] = sizes[(size_t)j] * metric(i, j);
                }
                while (front < stop) {
                    const int j = queue[(size_t)front];
                    const double loss = loss_of[(size_t)front];
                    ++front;
                    const double improvement = lambda - loss;
                    if (improvement > 0.0) {
                        set.link(j, i);
                        sizes[(size_t)i] += sizes[(size_t)j];
                        const int* a = list_of(j);
                        for (int e = 0, m = len[(size_t)j]; e < m; ++e) {
                            const int kk = set.find(a[e]);
                            if (!visited[(size_t)kk]) { visited[(size_t)kk] = 1; queue[(size_t)back++] = kk; }
                        }
                        len[(size_t)j] = 0;
                        if (--number_of_supervoxels == n_supervoxels) { reached = true; break; }
                    } else {
                        adjacent.push_back(j);
                    }
                }
            }
            off[(size_t)i] = arena_size;
            len[(size_t)i] = (int)adjacent.size();
            if (arena_size + adjacent.size() > arena.n) {
                pwhost::HostBuf<int> bigger;
                if (!bigger.reserve(std::max(2 * arena.n, arena_size + adjacent.size()))) return -1;
                std::memcpy(bigger.p, arena.p, arena_size * sizeof(int));
                arena.swap(bigger);
            }
            if (!adjacent.empty()) std::memcpy(arena.p + arena_size, adjacent.data(), adjacent.size() * sizeof(int));
            arena_size += adjacent.size();
            for (int j = 0; j < back; ++j) visited[(size_t)queue[(size_t)j]] = 0;
            if (number_of_supervoxels == n_supervoxels) break;
        }
        number_of_supervoxels = 0;
        for (int i : supervoxels)
            if (set.find(i) == i) supervoxels[(size_t)number_of_supervoxels++] = i;
        supervoxels.resize((size_t)number_of_supervoxels);
        if (number_of_supervoxels == n_supervoxels) break;
        // compact the arena in place: every surviving non-empty list was appended this round, in supervoxel order
        size_t w = 0;
        for (int i : supervoxels) {
            const int m = len[(size_t)i];
            if (m) std::memmove(arena.data() + w, arena.data() + off[(size_t)i], (size_t)m * sizeof(int));
            off[(size_t)i] = w;
            w += (size_t)m;
        }
        arena_size = w;
        if (tm.on) { char b[64]; std::snprintf(b, sizeof b, "    round -> %d sv, arena %zu", number_of_supervoxels, arena_size); tm.lap(b); }
    }
    std::vector<int>& labels = *labels_out;
    labels.resize((size_t)n_points);
    for (int i = 0; i < n_points; ++i) labels[(size_t)i] = set.find(i);
    tm.lap("  fusion");
    if (roots_out) { *roots_out = supervoxels; return (int)supervoxels.size(); }

    // ---- step 2: boundary refinement -------------------------------------------------------------------------
    boundary_refinement(metric, nb, k, n_points, labels);

    // ---- step 3: relabel -----------------------------------------------------------------------------------------
    std::vector<int> map((size_t)n_points, 0);
    for (size_t i = 0; i < supervoxels.size(); ++i) map[(size_t)supervoxels[i]] = (int)i;
    for (int i = 0; i < n_points; ++i) labels[(size_t)i] = map[(size_t)labels[(size_t)i]];
    return (int)supervoxels.size();
}

// normals (S.cpp:39-44) + segmentation (S.cpp:51-67) from the k-NN graph (n rows of k indices, the point itself first)
int segment_from_neighbors(const float* cloud_xyz4, int n, const int32_t* nb, int k, float sv_resolution, int32_t* labels,
                           int* n_supervoxels) {
    StageTimer tm;
    pwhost::HostBuf<Pt> P;
    if (!P.reserve((size_t)n)) return PWICP_E_NOMEM;
    for (int i = 0; i < n; ++i) {                                            // S.cpp:18-22: float -> double
        P.p[(size_t)i].x = (double)cloud_xyz4[4 * (size_t)i];
        P.p[(size_t)i].y = (double)cloud_xyz4[4 * (size_t)i + 1];
        P.p[(size_t)i].z = (double)cloud_xyz4[4 * (size_t)i + 2];
    }
    pwhost::parallel_for(n, [&](long long lo, long long hi) {
        for (long long i = lo; i < hi; ++i) pca_normal(P.data(), (int)i, nb + (size_t)i * (size_t)k, k);
    });
    tm.lap("pca normals");
    const double res = (double)sv_resolution;
    Metric metric{P.data(), res};
    const int n_sv = count_occupied_cells(P.data(), n, res);
    tm.lap("occupied cells");
    std::vector<int> lab;
    const int got = supervoxel_segmentation(metric, nb, k, n, n_sv, &lab);
    tm.lap("segmentation total");
    if (got < 0) return PWICP_E_NOMEM;
    for (int i = 0; i < n; ++i) labels[i] = lab[(size_t)i];
    *n_supervoxels = got;
    return PWICP_OK;
}

}  // namespace

namespace pwhost {
// stages of the front end for the device pipeline (csrc/frontend.hip), see frontend.h
void fe_points_and_normals(const float* cloud_xyz4, int n, const int32_t* nb, int k, FePt* P) {
    parallel_for(n, [&](long long lo, long long hi) {
        for (long long i = lo; i < hi; ++i) {
            P[(size_t)i].x = (double)cloud_xyz4[4 * (size_t)i];
            P[(size_t)i].y = (double)cloud_xyz4[4 * (size_t)i + 1];
            P[(size_t)i].z = (double)cloud_xyz4[4 * (size_t)i + 2];
        }
    });
    parallel_for(n, [&](long long lo, long long hi) {
        for (long long i = lo; i < hi; ++i) pca_normal(P, (int)i, nb + (size_t)i * (size_t)k, k);
    });
}
int fe_count_occupied_cells(const FePt* P, int n, double resolution) { return count_occupied_cells(P, n, resolution); }
// (min / max of the FLOATS, converted once: float -> double is monotone, so these are the min / max of the doubles; a serial loop over
// doubles cost 4 - 6 ms per 1 M points on the thread the front end's device work waits for)
void fe_bounding_box(const float* xyz4, int n, double mn[3], double mx[3]) {
    for (int d = 0; d < 3; ++d) { mn[d] = DBL_MAX; mx[d] = -DBL_MAX; }
    if (n <= 0) return;
    std::mutex mu;
    parallel_for(n, [&](long long lo, long long hi) {
        float a[4] = {FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX}, b[4] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (long long i = lo; i < hi; ++i) {
            const float* v = xyz4 + 4 * (size_t)i;
            for (int d = 0; d < 4; ++d) { a[d] = v[d] < a[d] ? v[d] : a[d]; b[d] = v[d] > b[d] ? v[d] : b[d]; }
        }
        std::lock_guard<std::mutex> g(mu);
        for (int d = 0; d < 3; ++d) { mn[d] = std::min(mn[d], (double)a[d]); mx[d] = std::max(mx[d], (double)b[d]); }
    }, 65536);
}
void fe_normals_from_scatter(const double* S6, int n, double* normals3) {
    parallel_for(n, [&](long long lo, long long hi) {
        for (long long i = lo; i < hi; ++i) {
            const double* c = S6 + 6 * (size_t)i;
            Sym3 C;
            C.xx = c[0]; C.xy = c[1]; C.xz = c[2]; C.yy = c[3]; C.yz = c[4]; C.zz = c[5];
            normal_from_scatter(C, normals3 + 3 * (size_t)i);
        }
    });
}
void fe_refine_host(const FePt* P, const int32_t* nb, int k, int n, double resolution, std::vector<int>* root_of) {
    Metric metric{P, resolution};
    boundary_refinement(metric, nb, k, n, *root_of);
}
int fe_fusion_host(const FePt* P, const int32_t* nb, int k, int n, double resolution, int n_supervoxels, std::vector<int>* root_of,
                   std::vector<int>* roots) {
    Metric metric{P, resolution};
    return supervoxel_segmentation(metric, nb, k, n, n_supervoxels, root_of, roots);
}

// host part of the front end from a given k-NN graph (thread-safe: no shared state), see io.h
int segment_from_knn(const float* cloud_xyz4, int n, const int32_t* nb, int k, float sv_resolution, int32_t* labels,
                     int* n_supervoxels) {
    return segment_from_neighbors(cloud_xyz4, n, nb, k, sv_resolution, labels, n_supervoxels);
}
}  // namespace pwhost

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
    std::vector<int32_t> nb((size_t)n * (size_t)knn);
    pwhost::parallel_for(n, [&](long long lo, long long hi) {                // S.cpp:37-41
        std::vector<pwhost::KdTree<double>::Hit> hits((size_t)knn);
        for (long long i = lo; i < hi; ++i) {
            const int c = tree.knn(pts.data() + 3 * (size_t)i, knn, hits.data());
            for (int e = 0; e < knn; ++e) nb[(size_t)i * (size_t)knn + (size_t)e] = e < c ? hits[(size_t)e].idx : (int32_t)i;
        }
    });
    return segment_from_neighbors(cloud_xyz4, n, nb.data(), knn, sv_resolution, labels, n_supervoxels);
}

void pw_frontend_count(int which);            // csrc/frontend.hip: the take-over counters of pwicp_frontend_fallback_counts
// Same result, with the k-NN graph built on the GPU (pwicp_knn): the variant the entry points use.
PWICP_API int pwicp_frontend_segment_dev(pwicp_context* ctx, const float* cloud_xyz4, int n, float sv_resolution, int knn,
                                         float point_spacing, int32_t* labels, int* n_supervoxels) {
    if (!ctx || !cloud_xyz4 || !labels || !n_supervoxels || n <= 0 || knn <= 0 || knn >= n || !(sv_resolution > 0.f))
        return PWICP_E_INVALID;
    const float cell_edge = point_spacing > 0.f ? 2.0f * point_spacing : 0.f;
    const char* e = std::getenv("PWICP_FRONTEND");
    if (!(e && std::strcmp(e, "host") == 0))
        return pw_frontend_segment_device(ctx, cloud_xyz4, n, knn, cell_edge, sv_resolution, labels, n_supervoxels);
    // $PWICP_FRONTEND=host: k-NN graph on the device, the serial passes on the host (same labels)
    pw_frontend_count(5);
    pwhost::HostBuf<int32_t> nb;
    if (!nb.reserve((size_t)n * knn)) return PWICP_E_NOMEM;
    const int rc = pwicp_knn(ctx, cloud_xyz4, n, knn, cell_edge, nb.data());
    if (rc != PWICP_OK) return rc;
    return segment_from_neighbors(cloud_xyz4, n, nb.data(), knn, sv_resolution, labels, n_supervoxels);
}

}  // extern "C"
