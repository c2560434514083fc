// Small host-side KD-tree for exact k-nearest-neighbour queries (setup stages only: segmentation front end,
// statistical outlier removal, point-spacing estimate).  Not used inside the registration loop — that runs on
// the GPU grid.  Results are sorted by (distance, index), so they do not depend on the traversal order.
#pragma once

#include <algorithm>
#include <cstdint>
#include <limits>
#include <vector>

namespace pwhost {

// Real = coordinate / metric type: double for the front end (codelibrary works on doubles), float for the
// PCL-style searches (flann::L2_Simple<float>: ((dx*dx)+dy*dy)+dz*dz accumulated in float).
template <typename Real>
class KdTree {
public:
    struct Hit {
        Real d2;
        int idx;
        bool operator<(const Hit& o) const { return d2 < o.d2 || (d2 == o.d2 && idx < o.idx); }
    };

    // pts: n points, `stride` Reals apart
    void build(const Real* pts, int n, int stride) {
        pts_ = pts; n_ = n; stride_ = stride;
        perm_.resize((size_t)n);
        for (int i = 0; i < n; ++i) perm_[(size_t)i] = i;
        nodes_.clear();
        nodes_.reserve((size_t)(n / 4 + 16));
        if (n > 0) build_rec(0, n);
    }

    // k nearest of q (3 Reals), ascending by (d2, idx); out must hold k hits; returns count (= min(k, n))
    int knn(const Real* q, int k, Hit* out) const {
        int cnt = 0;
        if (n_ > 0) search(0, q, k, out, cnt);
        return cnt;
    }

private:
    struct Node {
        int left, right;     // children or -1
        int lo, hi;          // perm range (leaf)
        int dim;
        Real split;
    };
    static constexpr int kLeaf = 12;

    Real coord(int i, int d) const { return pts_[(size_t)i * stride_ + d]; }

    int build_rec(int lo, int hi) {
        const int id = (int)nodes_.size();
        nodes_.push_back(Node{-1, -1, lo, hi, 0, 0});
        if (hi - lo <= kLeaf) return id;
        Real mn[3], mx[3];
        for (int d = 0; d < 3; ++d) { mn[d] = std::numeric_limits<Real>::max(); mx[d] = std::numeric_limits<Real>::lowest(); }
        for (int i = lo; i < hi; ++i)
            for (int d = 0; d < 3; ++d) {
                const Real v = coord(perm_[(size_t)i], d);
                mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v);
            }
        int dim = 0;
        for (int d = 1; d < 3; ++d) if (mx[d] - mn[d] > mx[dim] - mn[dim]) dim = d;
        if (!(mx[dim] > mn[dim])) return id;      // all points identical: keep as a leaf
        const int mid = lo + (hi - lo) / 2;
        std::nth_element(perm_.begin() + lo, perm_.begin() + mid, perm_.begin() + hi,
                         [&](int a, int b) { return coord(a, dim) < coord(b, dim); });
        const Real split = coord(perm_[(size_t)mid], dim);
        const int l = build_rec(lo, mid);
        const int r = build_rec(mid, hi);
        nodes_[(size_t)id].left = l; nodes_[(size_t)id].right = r;
        nodes_[(size_t)id].dim = dim; nodes_[(size_t)id].split = split;
        return id;
    }

    static Real dist2(const Real* a, const Real* b) {
        Real r = 0, d;
        d = a[0] - b[0]; r += d * d;
        d = a[1] - b[1]; r += d * d;
        d = a[2] - b[2]; r += d * d;
        return r;
    }

    static void insert(Hit* out, int& cnt, int k, Hit h) {
        if (cnt == k && !(h < out[k - 1])) return;
        int pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && h < out[pos - 1]) { out[pos] = out[pos - 1]; --pos; }
        out[pos] = h;
        if (cnt < k) ++cnt;
    }

    void search(int id, const Real* q, int k, Hit* out, int& cnt) const {
        const Node& nd = nodes_[(size_t)id];
        if (nd.left < 0) {
            for (int i = nd.lo; i < nd.hi; ++i) {
                const int p = perm_[(size_t)i];
                insert(out, cnt, k, Hit{dist2(q, pts_ + (size_t)p * stride_), p});
            }
            return;
        }
        const double diff = (double)q[nd.dim] - (double)nd.split;
        const int first = diff < 0 ? nd.left : nd.right, second = diff < 0 ? nd.right : nd.left;
        search(first, q, k, out, cnt);
        // visit the far side whenever its slab could still hold a point that ties or beats the current worst
        if (cnt < k || diff * diff * (1.0 - 1e-6) <= (double)out[k - 1].d2) search(second, q, k, out, cnt);
    }

    const Real* pts_ = nullptr;
    int n_ = 0, stride_ = 3;
    std::vector<int> perm_;
    std::vector<Node> nodes_;
};

}  // namespace pwhost
